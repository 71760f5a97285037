"""Does the placement of the TGB-v1 planes relative to each other matter (HBM channel aliasing)?  The 4K transmissive pass
with the planes where torch puts them, and carved out of one allocation at staggered offsets.
    python tools/gpu_plane_stagger.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from transmission_renderer_amd import synthetic
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
g0 = GBufferPlanes.from_numpy(synthetic.make_gbuffer(w, h), dev)
hdr0 = torch.zeros((h, w, 4), dtype=torch.float16, device=dev)
print("default placement: ", [hex(t.data_ptr() & 0xFFFFFF) for t in (g0.pos_depth, g0.nrm_scale, g0.uv, g0.material_id, hdr0)])

def timeit(g, hdr):
    fn = lambda: r.shade_transmission(g, scene["uniforms"], scene["push"], pyr, hdr)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.1:
        for _ in range(32): fn()
        torch.cuda.synchronize()
    res = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(200): fn()
        b.record(); torch.cuda.synchronize()
        res.append(a.elapsed_time(b) / 200 * 1e3)
    return sorted(res)[2]

print(f"default: {timeit(g0, hdr0):.1f} us")
for step in (256, 1024, 4352, 65536 + 4352, 1 << 20):
    tensors = [g0.pos_depth, g0.nrm_scale, g0.uv, g0.material_id, hdr0]
    sizes = [t.numel() * t.element_size() for t in tensors]
    arena = torch.zeros(sum(sizes) + 8 * (step + (1 << 21)), dtype=torch.uint8, device=dev)
    off, views = 0, []
    base = arena.data_ptr()
    for k, (t, n) in enumerate(zip(tensors, sizes)):
        off = (off + (1 << 21) - 1) // (1 << 21) * (1 << 21) + k * step      # 2 MB boundary + k * step
        pad = (-(base + off)) % 256
        off += pad
        v = arena[off:off + n].view(t.dtype).view(t.shape)
        v.copy_(t)
        views.append(v)
        off += n
    g = GBufferPlanes(views[0], views[1], views[2], views[3])
    print(f"planes at 2 MB boundaries + k * {step} B: {timeit(g, views[4]):.1f} us")
