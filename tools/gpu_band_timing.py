"""Kernel-only expectation for `bench.py --gpus N` on ONE GPU: the 4K transmissive pass over band 0 of N row bands
(tr_band_rows), back to back at steady-state clocks.   python tools/gpu_band_timing.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from transmission_renderer_amd import sharded, synthetic
from transmission_renderer_amd.renderer import GBufferPlanes, OpaquePyramid, TransmissionRenderer, load_ggx_lut
import bench

w, h = 3840, 2160
r = TransmissionRenderer(0)
dev = r.device
scene = synthetic.make_scene(w, h, num_point_lights=1, with_gbuffer=False)
r.upload_materials(scene["materials"]); r.upload_lights(scene["lights"]); r.upload_ggx_lut(load_ggx_lut())
r.set_cluster_tables(torch.from_numpy(scene["cluster_counts"].view(np.int32)).to(dev),
                     torch.from_numpy(scene["light_indices"].view(np.int32)).to(dev))
pyr = OpaquePyramid(w, h, dev)
pyr.level(0).copy_(bench.make_mip0_torch(w, h, dev))
r.generate_mips(pyr)
base = base2 = None
second = torch.cuda.Stream()
for n in (1, 2, 4, 8):
    rows, y0, y1 = sharded.band_rows(h, n, 0)
    g = GBufferPlanes.from_numpy(synthetic.make_gbuffer(w, h, rows=(y0, y1)), dev)
    hdr = torch.zeros((rows * n, w, 4), dtype=torch.float16, device=dev)
    fn = lambda: r.shade_transmission(g, scene["uniforms"], scene["push"], pyr, hdr, (0, y0, w, y1))
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.05:
        for _ in range(32): fn()
        torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(400): fn()
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / 400 * 1e3
    base = base or us
    # the same with consecutive frames' bands on two streams (two targets): a band's fill and drain overlap the next one's
    hdr2 = torch.zeros_like(hdr)
    g2 = GBufferPlanes(g.pos_depth.clone(), g.nrm_scale.clone(), g.uv.clone(), g.material_id.clone(), g.origin_x, g.origin_y)
    pyr2 = OpaquePyramid(w, h, dev)      # (a frame in flight has its own inputs: nothing of the other frame's in the caches)
    pyr2.texels.copy_(pyr.texels)
    cur = torch.cuda.current_stream()

    def in_flight(k):
        for i in range(k):
            with torch.cuda.stream(second if i & 1 else cur):
                r.shade_transmission(g2 if i & 1 else g, scene["uniforms"], scene["push"], pyr2 if i & 1 else pyr, hdr2 if i & 1 else hdr, (0, y0, w, y1))
        cur.wait_stream(second)
    in_flight(64); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); in_flight(400); b.record(); torch.cuda.synchronize()
    us2 = a.elapsed_time(b) / 400 * 1e3
    base2 = base2 or us2
    print(f"band 0 of {n}: {y1 - y0} rows, {us:7.1f} us per launch back to back -> kernel-only speed-up {base / us:4.2f}x of {n};"
          f"  two frames in flight {us2:6.1f} us -> {base2 / us2:4.2f}x")
