"""Localise transmissive-pass errors (run on the GPU box): error map statistics per region / material / lod."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from transmission_renderer_amd import synthetic, wire
from transmission_renderer_amd.renderer import GBufferPlanes, OpaquePyramid, TransmissionRenderer, load_ggx_lut
from oracle import oracle
w, h, nl = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
ro = float(sys.argv[4]) if len(sys.argv) > 4 and sys.argv[4] != "none" else None
scene = synthetic.make_scene(w, h, num_point_lights=nl, roughness_override=ro)
lut = load_ggx_lut()
r = TransmissionRenderer(0); dev = r.device
r.upload_materials(scene["materials"]); r.upload_lights(scene["lights"]); r.upload_ggx_lut(lut)
r.set_cluster_tables(torch.from_numpy(scene["cluster_counts"].view(np.int32)).to(dev), torch.from_numpy(scene["light_indices"].view(np.int32)).to(dev))
g = GBufferPlanes.from_numpy(scene["gbuffer"], dev)
b = oracle.SceneBinding(scene, lut)
mip0 = synthetic.make_opaque_mip0(w, h)
tex = oracle.new_pyramid(w, h, mip0); oracle.generate_mips(w, h, tex)
pyr = OpaquePyramid(w, h, dev); pyr.texels.copy_(torch.from_numpy(tex).to(dev))
t32 = torch.zeros((h, w, 4), dtype=torch.float32, device=dev)
r.shade_transmission(g, scene["uniforms"], scene["push"], pyr, t32); torch.cuda.synchronize()
_, w64 = oracle.shade_transmission(b, scene["gbuffer"], tex, nthreads=8, fp64=True)
got = t32.cpu().numpy().astype(np.float64)
rel = (np.abs(got - w64) / (np.abs(w64) + 1e-3)).max(axis=2)
print("max rel", rel.max(), "px > 1e-4:", int((rel > 1e-4).sum()), "of", rel.size)
mid = scene["gbuffer"]["material_id"]
for m in range(16):
    sel = mid == m
    if sel.any():
        mi = scene["materials"][m]
        import math
        c = min(max(mi.index_of_refraction * 2 - 2, 0), 1)
        lod = math.log2(w) * mi.roughness_factor * c
        print(f" mat {m:2d} rough {mi.roughness_factor:.3f} ior {mi.index_of_refraction:.3f} lod {lod:.3f} tf {mi.transmission_factor:.1f} px {int(sel.sum()):6d} bad {int((rel[sel] > 1e-4).sum()):6d} max {rel[sel].max():.2e}")
ys, xs = np.nonzero(rel > 1e-4)
if len(ys):
    print("bad y range", ys.min(), ys.max(), "x range", xs.min(), xs.max())
    for k in range(min(5, len(ys))):
        print(" ", ys[k], xs[k], "mat", mid[ys[k], xs[k]], got[ys[k], xs[k], :3], w64[ys[k], xs[k], :3])
