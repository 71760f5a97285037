"""What bounds a 1/N band of the 4K transmissive pass on ONE GPU (the kernel-only side of `bench.py --gpus N`), with COLD
inputs (every launch another input set): the band's launch, the same grid with an EMPTY body (ablation build, TR_ABLATE =
32 | 64 | 128: no plane loads, no shading, no stores — dispatch + wave start + drain only), and the whole frame.
    python tools/build_variant.py abl -DTR_ABLATION=1 && python tools/gpu_band_floor.py build_ab/libtr_abl.so"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from transmission_renderer_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])          # experiments only (the ablation build)
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
pyrs = []
for _ in range(4):
    p = OpaquePyramid(w, h, dev)
    p.level(0).copy_(bench.make_mip0_torch(w, h, dev))
    r.generate_mips(p)
    pyrs.append(p)
out = {}
for n in (1, 2, 4, 8):
    rows, y0, y1 = sharded.band_rows(h, n, n // 2)      # a band from the middle of the frame
    gs = [bench.make_gbuffer_torch(w, h, dev, rows=(y0, y1)) for _ in range(4)]
    hdrs = [torch.zeros((h, w, 4), dtype=torch.float16, device=dev) for _ in range(4)]
    res = {}
    for name, ablate in (("band", "0"), ("empty_body", "224")):
        os.environ["TR_ABLATE"] = ablate
        k = [0]

        def fn():
            i = k[0] % 4
            k[0] += 1
            r.shade_transmission(gs[i], scene["uniforms"], scene["push"], pyrs[i], hdrs[i], (0, y0, w, y1))
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.05:
            for _ in range(32): fn()
            torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(400): fn()
        b.record(); torch.cuda.synchronize()
        res[name] = round(a.elapsed_time(b) / 400 * 1e3, 2)
        # the same launches replayed from a HIP graph: no host call between them — what the GPU alone needs per launch
        side = torch.cuda.Stream()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            for _ in range(8): fn()
            side.synchronize()
            with torch.cuda.graph(graph, stream=side):
                for _ in range(100): fn()
            graph.replay(); side.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(side)
            for _ in range(4): graph.replay()
            b.record(side); side.synchronize()
        res[name + "_graph"] = round(a.elapsed_time(b) / 400 * 1e3, 2)
    out[f"1/{n}"] = dict(res, rows=y1 - y0)
os.environ["TR_ABLATE"] = "0"
whole = out["1/1"]["band"]
for k_, v in out.items():
    v["speedup_vs_whole_frame"] = round(whole / v["band"], 2)
print(json.dumps(out))
