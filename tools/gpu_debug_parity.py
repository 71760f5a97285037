"""Per-pass GPU-vs-oracle error report (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from transmission_renderer_amd import synthetic, wire
from transmission_renderer_amd.renderer import GBufferPlanes, OpaquePyramid, TransmissionRenderer, load_ggx_lut
from oracle import oracle

def stats(name, got, want):
    got = got.astype(np.float64); want = want.astype(np.float64)
    bad = ~(np.isfinite(got) & np.isfinite(want))
    d = np.where(bad, 0, got - want)
    rmse = np.sqrt((d ** 2).mean(axis=(0, 1)))
    rel = np.abs(d) / (np.abs(want) + 1e-3)
    i = np.unravel_index(np.argmax(rel), rel.shape)
    print(f"{name}: rmse/ch {rmse[:3]}, max abs {np.abs(d).max():.3e}, max rel {rel.max():.3e} at {i} got {got[i[0], i[1]]} want {want[i[0], i[1]]}, nonfinite {bad.sum()}")
    return rel

w, h = int(sys.argv[1]) if len(sys.argv) > 1 else 256, int(sys.argv[2]) if len(sys.argv) > 2 else 256
nl = int(sys.argv[3]) if len(sys.argv) > 3 else 2
scene = synthetic.make_scene(w, h, num_point_lights=nl)
lut = load_ggx_lut()
r = TransmissionRenderer(0); dev = r.device
r.upload_materials(scene["materials"]); r.upload_lights(scene["lights"]); r.upload_ggx_lut(lut)
r.set_cluster_tables(torch.from_numpy(scene["cluster_counts"].view(np.int32)).to(dev), torch.from_numpy(scene["light_indices"].view(np.int32)).to(dev))
g = GBufferPlanes.from_numpy(scene["gbuffer"], dev)
b = oracle.SceneBinding(scene, lut)
# opaque, fp32
hdr32 = torch.zeros((h, w, 4), dtype=torch.float32, device=dev)
pyr = OpaquePyramid(w, h, dev)
r.shade_opaque(g, scene["uniforms"], scene["push"], hdr32, pyr)
torch.cuda.synchronize()
o16, o32, omip0 = oracle.shade_opaque(b, scene["gbuffer"], nthreads=8)
stats("opaque fp32", hdr32.cpu().numpy(), o32)
print("opaque mip0 f16 mismatches:", int((pyr.level(0).cpu().numpy().view(np.uint16) != omip0.view(np.uint16)).sum()))
# mips: same input
mip0 = synthetic.make_opaque_mip0(w, h)
tex = oracle.new_pyramid(w, h, mip0)
oracle.generate_mips(w, h, tex)
pyr.level(0).copy_(torch.from_numpy(mip0).to(dev))
r.generate_mips(pyr); torch.cuda.synchronize()
gt = pyr.texels.cpu().numpy()
print("mip chain u16 mismatches:", int((gt.view(np.uint16) != tex.view(np.uint16)).sum()), "of", gt.size)
# transmission fp32 with the oracle pyramid
pyr.texels.copy_(torch.from_numpy(tex).to(dev))
t32 = torch.zeros((h, w, 4), dtype=torch.float32, device=dev)
r.shade_transmission(g, scene["uniforms"], scene["push"], pyr, t32); torch.cuda.synchronize()
w16, w32 = oracle.shade_transmission(b, scene["gbuffer"], tex, nthreads=8)
_, w64 = oracle.shade_transmission(b, scene["gbuffer"], tex, nthreads=8, fp64=True)
rel = stats("transmission fp32 (gpu vs oracle32)", t32.cpu().numpy(), w32)
stats("transmission      (gpu vs oracle64)", t32.cpu().numpy(), w64)
stats("transmission (oracle32 vs oracle64)", w32, w64)
noise = np.abs(w32 - w64).max(axis=2); err = np.abs(t32.cpu().numpy() - w32).max(axis=2)
good = noise <= 1e-5 * (np.abs(w64).max(axis=2) + 1e-3)
print("well-conditioned px:", int(good.sum()), "of", good.size, " rmse on them:", np.sqrt(((t32.cpu().numpy() - w32)[good] ** 2).mean(axis=0)))
mid = scene["gbuffer"]["material_id"]
for m in range(16):
    sel = mid == m
    if sel.any():
        print(f"  material {m:2d}: px {int(sel.sum()):7d} max rel {rel[sel].max():.3e} mean rel {rel[sel].mean():.3e}")
t16 = torch.zeros((h, w, 4), dtype=torch.float16, device=dev)
r.shade_transmission(g, scene["uniforms"], scene["push"], pyr, t16); torch.cuda.synchronize()
stats("transmission f16 target", t16.cpu().numpy().astype(np.float32), w16.astype(np.float32))
