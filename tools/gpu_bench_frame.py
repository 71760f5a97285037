"""Whole-frame rate of the glTF-in/frame-out pipeline at steady-state clocks (run on the GPU box): tr_record_frame
(culling, light assignment, demultiplex, rasteriser, opaque, mips, transmissive, tonemap) back to back.
    python tools/gpu_bench_frame.py [scene.glb | meshes | plain] [width height]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from transmission_renderer_amd import _lib, gltf, meshes, synthetic, wire
if os.environ.get("TR_AB_LIB"):            # an A/B build of the library (tools/build_variant.py)
    _lib.LIB_PATH = os.path.abspath(os.environ["TR_AB_LIB"])
from transmission_renderer_amd.renderer import TransmissionRenderer

name = sys.argv[1] if len(sys.argv) > 1 else "meshes"
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3840, 2160)
r = TransmissionRenderer(0)
scene = synthetic.make_scene(w, h, num_point_lights=2, with_gbuffer=False, textured=(name in ("meshes", "room", "room_near_first")))
if name in ("meshes", "plain", "room", "room_near_first"):   # (plain: the same geometry with untextured materials; room: inside a closed room)
    geometry = meshes.make_mesh_scene(extra_instances=True, room={"room": 1, "room_near_first": 2}.get(name, False))
    scene["materials"][2].alpha_clipping_cutoff = 0.75
    scene["materials"][7].alpha_clipping_cutoff = 0.6
else:
    loaded = gltf.load_gltf(name, base_transform=meshes.Similarity(np.array([0.0, 2.0, 0.0], np.float32), 1.0))
    geometry = loaded.geometry()
    scene["materials"] = loaded.materials or [wire.MaterialInfo.default()]
    scene["textures"] = loaded.textures
r.upload_ggx_lut(); r.upload_materials(scene["materials"])
if scene.get("textures"): r.upload_textures(scene["textures"])
r.upload_lights(scene["lights"]); r.upload_geometry(geometry)
_, view = wire.default_camera()
aabbs = r.write_cluster_data(scene["uniforms"], wire.inverse_perspective(w, h), (w, h))
culling = wire.CullingPushConstants.new(wire.perspective_matrix_reversed(w, h), view)
work = r.new_frame_buffers(w, h)
frame = lambda: r.record_frame(scene["uniforms"], scene["push"], culling, view, wire.view_rotation_inverse(view), aabbs, work)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.2:
    for _ in range(4): frame()
    torch.cuda.synchronize()
ts = []
for _ in range(8):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): frame()
    b.record(); b.synchronize(); ts.append(a.elapsed_time(b) / 20)
ts.sort()
t = ts[len(ts) // 2]
if hasattr(r.lib, "tr_debug_read_timing") and os.environ.get("TR_AB_LIB"):   # a -DTR_TIMING=1 build: where the shading waves wait
    import ctypes as C
    buf = (C.c_ulonglong * 12)()
    torch.cuda.synchronize(); r.lib.tr_debug_read_timing(buf)
    for _ in range(10): frame()
    torch.cuda.synchronize(); r.lib.tr_debug_read_timing(buf)
    w0, w1, w2, total, tiles, waves = list(buf)[:6]
    print(f"shading waves of 10 frames (both passes), cycles per tile and wave: inputs wait {w0 / tiles:.0f}, cluster lists {w1 / tiles:.0f}, "
          f"taps + LUT {w2 / tiles:.0f}, texture front end {buf[8] / tiles:.0f}, light loops {buf[9] / tiles:.0f}, whole tile {total / tiles:.0f}; "
          f"{tiles / (waves & 0xFFFFFFFF):.2f} tiles per wave")
if hasattr(r.lib, "tr_debug_read_raster_timing") and os.environ.get("TR_AB_LIB"):   # a -DTR_RASTER_TIMING build: the opaque layer's raster waves
    import ctypes as C
    buf = (C.c_ulonglong * 12)()
    torch.cuda.synchronize(); r.lib.tr_debug_read_raster_timing(buf)
    frame()
    torch.cuda.synchronize(); r.lib.tr_debug_read_raster_timing(buf)
    s_, p_, bl, tot, items, waves, longest, nblocks, issue, most, wg_longest = [int(x) / 1.0 for x in list(buf)[:11]]
    print(f"raster waves {waves:.0f}: live items {items:.0f} (most in a wave {most:.0f}), blocks visited {nblocks:.0f}; s_memtime ticks per wave, mean: "
          f"total {tot / waves:.0f} (issue {issue / waves:.0f}), preparation {s_ / waves:.0f}, item hand-over {p_ / waves:.0f}, block loops {bl / waves:.0f}; "
          f"longest wave {longest:.0f}, busiest workgroup's mean wave {wg_longest:.0f}")
print(f"{name} {w}x{h}, {len(geometry['index']) // 3} triangles: {t * 1e3:.1f} us per frame ({1e3 / t:.0f} frames/s), "
      f"culling -> rasteriser -> opaque -> mips -> transmissive -> tonemap, one tr_record_frame call per frame")
