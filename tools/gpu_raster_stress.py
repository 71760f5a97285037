"""Rasteriser timing on a large procedural scene (run on the GPU box): N finely tessellated spheres + floor."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from transmission_renderer_amd import _lib, meshes, synthetic, wire
if os.environ.get("TR_AB_LIB"):            # an A/B build of the library (tools/build_variant.py)
    _lib.LIB_PATH = os.path.abspath(os.environ["TR_AB_LIB"])
from transmission_renderer_amd.renderer import TransmissionRenderer

n_spheres, seg = int(sys.argv[1]) if len(sys.argv) > 1 else 200, int(sys.argv[2]) if len(sys.argv) > 2 else 72
w, h = 3840, 2160
rng = np.random.default_rng(1)
mb = meshes.ModelBuffers()
S = meshes.Similarity
mb.add_primitive(meshes.plane(40.0, 40.0, cells=8), 0, [(S(np.array([0, 0.6, -6.0], np.float32)), 3)])
sphere = meshes.uv_sphere(1.0, seg, seg // 2)
inst_o = [(S(np.array([rng.uniform(-6, 6), rng.uniform(0.8, 4), rng.uniform(-14, -2)], np.float32), rng.uniform(0.2, 0.6)), int(rng.integers(0, 16)))
          for _ in range(n_spheres)]
inst_t = [(S(np.array([rng.uniform(-5, 5), rng.uniform(0.8, 4), rng.uniform(-10, -1.5)], np.float32), rng.uniform(0.2, 0.5)), 4)
          for _ in range(n_spheres // 4)]
mb.add_primitive(sphere, 0, inst_o)
mb.add_primitive(sphere, 2, inst_t)
geo = mb.finish()
ntri = (len(sphere.index) // 3) * (len(inst_o) + len(inst_t)) + 128
r = TransmissionRenderer(0)
sc = synthetic.make_scene(w, h, num_point_lights=1, with_gbuffer=False)
r.upload_ggx_lut(); r.upload_materials(sc["materials"]); r.upload_geometry(geo)
_, view = wire.default_camera()
culling = wire.CullingPushConstants.new(wire.perspective_matrix_reversed(w, h), view)
o, t = r.new_layer(w, h), r.new_layer(w, h)
for _ in range(3): r.draw_scene(culling, sc["push"], o, t)
torch.cuda.synchronize()
ts = []
for _ in range(10):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); r.draw_scene(culling, sc["push"], o, t); b.record(); b.synchronize()
    ts.append(a.elapsed_time(b))
cov = (o.material_id != -1).float().mean().item(), (t.material_id != -1).float().mean().item()
print(f"{ntri} triangles, {w}x{h}: draw_scene (culling + demultiplex + 2 x (scan, setup, scan, raster, resolve)) p50 {sorted(ts)[5]:.3f} ms, "
      f"coverage opaque {cov[0]:.2f} transmissive {cov[1]:.2f}")
if hasattr(r.lib, "tr_debug_read_raster_timing"):   # a -DTR_RASTER_TIMING build (tools/build_variant.py): the opaque layer's raster waves
    import ctypes
    buf = (ctypes.c_ulonglong * 12)()
    torch.cuda.synchronize(); r.lib.tr_debug_read_raster_timing(buf)
    r.draw_scene(culling, sc["push"], o, t)
    torch.cuda.synchronize(); r.lib.tr_debug_read_raster_timing(buf)
    s, p, bl, tot, items, waves, longest, nblocks, issue, most, wg_longest = [int(x) for x in list(buf)[:11]]
    print(f"  raster waves {waves}: live items {items} (most in a wave {most}), blocks visited {nblocks}; s_memtime ticks per wave, mean: total {tot / waves:.0f} "
          f"(issue {issue / waves:.0f}), preparation {s / waves:.0f}, item hand-over {p / waves:.0f}, block loops {bl / waves:.0f}; longest wave {longest}, "
          f"busiest workgroup's mean wave {wg_longest}")
