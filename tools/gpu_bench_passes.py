"""4K timings of every kernel on the hot path (untextured synthetic frame); run on the GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from transmission_renderer_amd import _lib
if len(sys.argv) > 2:
    _lib.LIB_PATH = os.path.abspath(sys.argv[2])   # python tools/gpu_bench_passes.py <lights> [lib.so]
from transmission_renderer_amd import synthetic, wire
from transmission_renderer_amd.renderer import GBufferPlanes, OpaquePyramid, TransmissionRenderer

w, h = 3840, 2160
nl = int(sys.argv[1]) if len(sys.argv) > 1 else 1
r = TransmissionRenderer(0)
scene = synthetic.make_scene(w, h, num_point_lights=nl)
r.upload_ggx_lut(); r.upload_materials(scene["materials"]); r.upload_lights(scene["lights"])
r.set_cluster_tables(torch.from_numpy(scene["cluster_counts"].view(np.int32)).to(r.device), torch.from_numpy(scene["light_indices"].view(np.int32)).to(r.device))
g = GBufferPlanes.from_numpy(scene["gbuffer"], r.device)
pyr = OpaquePyramid(w, h, r.device)
hdr = torch.zeros((h, w, 4), dtype=torch.float16, device=r.device)
_, view = wire.default_camera()
aabbs = r.write_cluster_data(scene["uniforms"], wire.inverse_perspective(w, h), (w, h))
px = w * h
passes = [
    ("shade_opaque (+mip0)", lambda: r.shade_opaque(g, scene["uniforms"], scene["push"], hdr, pyr), 52),
    ("generate_mips", lambda: r.generate_mips(pyr), 10.67),
    ("shade_transmission", lambda: r.shade_transmission(g, scene["uniforms"], scene["push"], pyr, hdr), 60),
    ("tonemap", lambda: r.tonemap(hdr), 12),
]
# steady state: the clocks ramp over the first ~10 ms of continuous load, so warm up for 100 ms and time back-to-back
# batches with one pair of events (a launch + sync per step reads 20-30 % slow)
import time
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.1:
    for _, fn, _ in passes: fn()
    torch.cuda.synchronize()
for name, fn, bpp in passes:
    ts = []
    for _ in range(8):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50): fn()
        b.record(); b.synchronize(); ts.append(a.elapsed_time(b) * 1e3 / 50)
    ts.sort()
    t = ts[len(ts) // 2]
    print(f"{name:24s} p50 {t:7.1f} us   {px * bpp / t / 1e6:6.2f} TB/s of {bpp} B/px algorithmic ({px * bpp / 8e6 / t * 100:4.1f} % of 8 TB/s)")
