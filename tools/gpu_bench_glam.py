"""Throughput of the batched glam-pbr API (run on the GPU box): elements/s and achieved HBM GB/s per function."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import glam_cases
from transmission_renderer_amd import synthetic, wire
from transmission_renderer_amd.glam_pbr import GlamPbr
from transmission_renderer_amd.png import read_png_rgba8
from transmission_renderer_amd.renderer import OpaquePyramid, TransmissionRenderer

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
r = TransmissionRenderer(0)
r.upload_ggx_lut(read_png_rgba8(os.path.join(os.path.dirname(wire.__file__), "assets", "ggx_lut.png")))
api = GlamPbr(r)
w, h = 3840, 2160
pyr = OpaquePyramid(w, h, r.device)
pyr.level(0).copy_(torch.from_numpy(synthetic.make_opaque_mip0(w, h)).to(r.device)); r.generate_mips(pyr)
tile = 1 << 16
def dev(records, words):   # a small seeded batch tiled to n elements, resident in HBM
    t = torch.from_numpy(records.view(np.float32).reshape(-1, words).copy()).to(r.device)
    return t.repeat((n + tile - 1) // tile, 1)[:n].contiguous()
cases = [("basic_brdf", lambda a: api.basic_brdf(a), dev(glam_cases.basic_brdf_params(tile), 22), 88 + 24),
         ("transmission_btdf", lambda a: api.transmission_btdf(a), dev(glam_cases.transmission_btdf_params(tile), 19), 76 + 12),
         ("ibl_volume_refraction", lambda a: api.ibl_volume_refraction(a, pyr), dev(glam_cases.ibl_params(tile, w, h), 42), 168 + 12 + 8)]
for name, fn, arr, bytes_per in cases:
    for _ in range(3): fn(arr)
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(arr); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b) * 1e-3)
    t = sorted(ts)[len(ts) // 2]
    print(f"{name:24s} {n} elements: p50 {t * 1e6:8.1f} us  {n / t / 1e9:6.2f} G elements/s  {n * bytes_per / t / 1e9:7.1f} GB/s algorithmic "
          f"({bytes_per} B/element; includes the output allocation)")
