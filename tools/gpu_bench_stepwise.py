"""The three shading steps through the TGB-v1 planes (tr_shade_opaque -> tr_generate_mips -> tr_shade_transmission) on a
rasterised glTF scene at 4K, back to back: what a host that calls the entry points one by one gets.
    python tools/gpu_bench_stepwise.py scene.glb [libtr_shade.so]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from transmission_renderer_amd import _lib, gltf, meshes, synthetic, wire
if len(sys.argv) > 2:
    _lib.LIB_PATH = os.path.abspath(sys.argv[2])
from transmission_renderer_amd.renderer import OpaquePyramid, TransmissionRenderer

w, h = 3840, 2160
loaded = gltf.load_gltf(sys.argv[1], base_transform=meshes.Similarity(np.array([0.0, 2.0, 0.0], np.float32), 1.0))
r = TransmissionRenderer(0)
r.upload_ggx_lut()
view = wire.default_camera()[1]
sc = synthetic.make_scene(w, h, num_point_lights=2, with_gbuffer=False)
sc["materials"], sc["textures"] = loaded.materials, loaded.textures
q = wire.view_rotation_inverse(view)
culling = wire.CullingPushConstants.new(wire.perspective_matrix_reversed(w, h), view)
r.upload_materials(sc["materials"]); r.upload_textures(sc["textures"]); r.upload_lights(sc["lights"]); r.upload_geometry(loaded.geometry())
aabbs = r.write_cluster_data(sc["uniforms"], wire.inverse_perspective(w, h), (w, h))
r.assign_lights_to_clusters(view, q, aabbs)
o, t = r.new_layer(w, h), r.new_layer(w, h)
r.draw_scene(culling, sc["push"], o, t)
pyr = OpaquePyramid(w, h, r.device)
hdr = torch.zeros((h, w, 4), dtype=torch.float16, device=r.device)
fn = lambda: r.record(o, t, sc["uniforms"], sc["push"], hdr, pyr)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.2:
    for _ in range(4): fn()
    torch.cuda.synchronize()
ts = []
for _ in range(8):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): fn()
    b.record(); b.synchronize(); ts.append(a.elapsed_time(b) / 20)
print(f"stepwise opaque -> mips -> transmissive on {os.path.basename(sys.argv[1])} at {w}x{h}: {sorted(ts)[4] * 1e3:.1f} us")
