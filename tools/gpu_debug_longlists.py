#!/usr/bin/env python3
"""Debug: borrowed cluster tables with counts past the capacity (tests/test_gpu_parity.py)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle
from transmission_renderer_amd import synthetic, wire
from transmission_renderer_amd.renderer import GBufferPlanes, OpaquePyramid, TransmissionRenderer, load_ggx_lut
lut = load_ggx_lut()
r = TransmissionRenderer(0); r.upload_ggx_lut(lut)
w, h = 250, 130
cap = wire.MAX_LIGHTS_PER_CLUSTER
for variant in ("over", "exact128", "same_lists", "short"):
    scene = synthetic.make_scene(w, h, num_point_lights=3)
    rng = np.random.default_rng(9)
    counts = rng.integers(cap - 3, cap + 120, wire.NUM_CLUSTERS).astype(np.uint32)
    lists = rng.integers(0, 3, (wire.NUM_CLUSTERS, cap)).astype(np.uint32)
    if variant == "exact128": counts[:] = cap
    if variant == "same_lists": lists[:] = lists[0]
    if variant == "short": counts = rng.integers(0, 9, wire.NUM_CLUSTERS).astype(np.uint32)
    for l in scene["lights"]:
        for k in range(3): l.colour_emission_and_falloff_distance_sq[k] *= 1.0 / 64.0
    scene["cluster_counts"], scene["light_indices"] = counts, lists.reshape(-1)
    r.upload_materials(scene["materials"]); r.upload_lights(scene["lights"])
    r.set_cluster_tables(torch.from_numpy(counts.view(np.int32)).to(r.device), torch.from_numpy(lists.reshape(-1).view(np.int32)).to(r.device))
    g = GBufferPlanes.from_numpy(scene["gbuffer"], r.device)
    tex = oracle.new_pyramid(w, h, synthetic.make_opaque_mip0(w, h)); oracle.generate_mips(w, h, tex)
    pyr = OpaquePyramid(w, h, r.device); pyr.texels.copy_(torch.from_numpy(tex).to(r.device))
    t32 = torch.zeros((h, w, 4), dtype=torch.float32, device=r.device)
    o32 = torch.zeros((h, w, 4), dtype=torch.float32, device=r.device)
    r.shade_transmission(g, scene["uniforms"], scene["push"], pyr, t32)
    r.shade_opaque(g, scene["uniforms"], scene["push"], o32, None)
    torch.cuda.synchronize()
    clamped = dict(scene, cluster_counts=np.minimum(counts, cap))
    b = oracle.SceneBinding(clamped, lut)
    _, want_t = oracle.shade_transmission(b, scene["gbuffer"], tex, nthreads=8, fp64=True)
    _, want_o, _ = oracle.shade_opaque(b, scene["gbuffer"], nthreads=8, fp64=True)
    for got, want, what in ((t32.cpu().numpy(), want_t, "transmission"), (o32.cpu().numpy(), want_o, "opaque")):
        e = (got.astype(np.float64) - want) / np.maximum(np.abs(want), 1.0)
        a = np.abs(e[..., :3]).max(axis=2)
        print(variant, what, "rmse", np.sqrt((e[..., :3] ** 2).mean(axis=(0, 1))).max(), "max", a.max(), "bad px", (a > 1e-3).sum())
        ys, xs = np.nonzero(a > 1e-3)
        for y, x in list(zip(ys, xs))[:6]:
            print("   ", y, x, "mat", scene["gbuffer"]["material_id"][y, x], "gpu", got[y, x, :3], "want", want[y, x, :3])
        if len(ys): print("    bad px tile columns (x//16):", np.unique(xs // 16)[:20], "rows//4:", np.unique(ys // 4)[:20])
r.close()
