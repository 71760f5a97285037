"""4K timing of the TEXTURED shading kernels on a G-buffer where every material carries the typical glTF texture set
(base colour sRGB + metallic-roughness + normal map, 1024^2 each); run on the GPU box.
    python tools/gpu_bench_textured.py [lib.so]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from transmission_renderer_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from transmission_renderer_amd import synthetic, wire
from transmission_renderer_amd.renderer import GBufferPlanes, OpaquePyramid, TransmissionRenderer

w, h = 3840, 2160
r = TransmissionRenderer(0)
scene = synthetic.make_scene(w, h, num_point_lights=1)
rng = np.random.default_rng(4)
size = 1024
x = (np.arange(size) + 0.5) / size
base = np.stack([np.broadcast_to(0.5 + 0.5 * np.sin(x * 60)[None, :], (size, size)), np.broadcast_to(x[:, None], (size, size)),
                 np.broadcast_to(0.3 + 0.5 * x[None, :], (size, size)), np.ones((size, size))], axis=-1)
tex = [((base * 255).astype(np.uint8), True),
       ((np.stack([np.zeros((size, size)), 0.2 + 0.6 * np.broadcast_to(x[None, :], (size, size)), np.ones((size, size)) * 0.5, np.ones((size, size))], -1) * 255).astype(np.uint8), False),
       ((np.stack([0.5 + 0.2 * np.broadcast_to(np.sin(x * 90)[None, :], (size, size)), 0.5 + 0.2 * np.broadcast_to(np.cos(x * 70)[:, None], (size, size)), np.ones((size, size)) * 0.9, np.ones((size, size))], -1) * 255).astype(np.uint8), False)]
for m in scene["materials"]:
    m.textures.diffuse, m.textures.metallic_roughness, m.textures.normal_map = 0, 1, 2
scene["gbuffer"]["uv"] *= np.float32(3.0)
r.upload_ggx_lut(); r.upload_materials(scene["materials"]); r.upload_lights(scene["lights"]); r.upload_textures(tex)
r.set_cluster_tables(torch.from_numpy(scene["cluster_counts"].view(np.int32)).to(r.device), torch.from_numpy(scene["light_indices"].view(np.int32)).to(r.device))
g = GBufferPlanes.from_numpy(scene["gbuffer"], r.device)
pyr = OpaquePyramid(w, h, r.device)
pyr.level(0).copy_(torch.from_numpy(synthetic.make_opaque_mip0(w, h)).to(r.device)); r.generate_mips(pyr)
hdr = torch.zeros((h, w, 4), dtype=torch.float16, device=r.device)
for name, fn in (("transmissive", lambda: r.shade_transmission(g, scene["uniforms"], scene["push"], pyr, hdr)),
                 ("opaque", lambda: r.shade_opaque(g, scene["uniforms"], scene["push"], hdr, None))):
    import time
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.1:   # steady-state clocks (see tools/ab_kernel.py)
        for _ in range(8): fn()
        torch.cuda.synchronize()
    ts = []
    for _ in range(8):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(25): fn()
        b.record(); b.synchronize(); ts.append(a.elapsed_time(b) * 1e3 / 25)
    ts.sort()
    print(f"textured 4K {name}: p50 {ts[len(ts) // 2]:.1f} us per launch (back to back, steady-state clocks)")
