#!/usr/bin/env python3
"""Prints, for every parity configuration of tests/, the PLAIN all-pixel per-channel RMSE of the HIP passes against
the pinned artefacts (SPIR-V outputs of the golden cases; the fp32 oracle elsewhere) — raw and normalised by
max(|ref|, 1), on the RGBA32F and the RGBA16F target — next to the fp64 twin that explains the outliers.
Run on the GPU box:  python tools/gpu_parity_report.py > gpurun_out/parity_report.txt
"""
import glob
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import oracle  # noqa: E402
from transmission_renderer_amd import synthetic, wire  # noqa: E402
from transmission_renderer_amd.renderer import GBufferPlanes, OpaquePyramid, TransmissionRenderer, load_ggx_lut  # noqa: E402


def rmse(got, ref, norm):
    got, ref = got.astype(np.float64)[..., :3], ref.astype(np.float64)[..., :3]
    e = got - ref
    if norm:
        e = e / np.maximum(np.abs(ref), 1.0)
    e = e.reshape(-1, 3)
    ok = np.isfinite(e).all(axis=1)   # (a value beyond the half range is +inf on both sides of an RGBA16F comparison: counted, not averaged)
    return [float(x) for x in np.sqrt((e[ok] ** 2).mean(axis=0))], float(np.abs(e[ok]).max()), int((~ok).sum())


def line(what, got, ref):
    raw, raw_max, bad = rmse(got, ref, False)
    nrm, nrm_max, _ = rmse(got, ref, True)
    # north_star's literal bound where it is meaningful: plain RMSE over the pixels whose reference is <= 1 in every channel
    g3, r3 = got.astype(np.float64)[..., :3].reshape(-1, 3), ref.astype(np.float64)[..., :3].reshape(-1, 3)
    low = (np.abs(r3) <= 1.0).all(axis=1) & np.isfinite(g3).all(axis=1)
    plain_low = float(np.sqrt(((g3[low] - r3[low]) ** 2).mean(axis=0)).max()) if low.any() else 0.0
    print(f"  {what:<38} raw rmse {max(raw):.3e} (max {raw_max:.3e})   normalised rmse {max(nrm):.3e} (max {nrm_max:.3e})"
          f"   plain rmse where |ref| <= 1: {plain_low:.3e} ({int(low.sum())} px)"
          f"   non-finite {bad}   |ref| max {np.abs(ref[np.isfinite(ref)]).max():.3g}")
    return {"what": what, "raw_rmse": raw, "norm_rmse": nrm, "raw_max": raw_max, "norm_max": nrm_max, "plain_rmse_ref_le_1": plain_low,
            "pixels_ref_le_1": int(low.sum())}


_TM = None


def display(frame16):
    """What the reference presents: fragment_tonemap (Lottes, tr_lottes_defaults) of the RGBA16F attachment, linear [0, 1]."""
    global _TM
    if _TM is None:
        import ctypes as C
        from transmission_renderer_amd import _lib
        lp, _TM = wire.LottesParams(), wire.TonemapParams()
        lib = _lib.load()
        lib.tr_lottes_defaults(C.byref(lp))
        lib.tr_bake_lottes_params(C.byref(lp), C.byref(_TM))
    return oracle.tonemap_frame(frame16, _TM)[1]


def upload(r, scene):
    dev = r.device
    r.upload_materials(scene["materials"])
    r.upload_lights(scene["lights"])
    r.set_cluster_tables(torch.from_numpy(scene["cluster_counts"].view(np.int32)).to(dev),
                         torch.from_numpy(scene["light_indices"].view(np.int32)).to(dev))


def main():
    lut = load_ggx_lut()
    r = TransmissionRenderer(0)
    r.upload_ggx_lut(lut)
    report = []
    from test_oracle_vs_spirv import GOLDEN, GOLDEN_SAMPLED, _scene_from_fixture, sampled_scene
    for path in GOLDEN_SAMPLED:   # sampled pixels of whole 4K frames: d, e the benchmark's own, f textured materials
        z = np.load(path)
        scene, w, h = sampled_scene(z)
        print(os.path.basename(path), f"{w}x{h}, {len(z['pixels'])} sampled pixels")
        upload(r, scene)
        r.upload_textures(scene.get("textures", []))
        planes = GBufferPlanes.from_numpy(synthetic.make_gbuffer(w, h), r.device)
        pyr = OpaquePyramid(w, h, r.device)
        pyr.level(0).copy_(torch.from_numpy(synthetic.make_opaque_mip0(w, h)).to(r.device))
        r.generate_mips(pyr)
        ys, xs = z["pixels"][:, 0], z["pixels"][:, 1]
        iy, ix = torch.from_numpy(ys.astype(np.int64)).to(r.device), torch.from_numpy(xs.astype(np.int64)).to(r.device)
        for dt in (torch.float32, torch.float16):
            t = torch.zeros((h, w, 4), dtype=dt, device=r.device)
            o = torch.zeros((h, w, 4), dtype=dt, device=r.device)
            r.shade_transmission(planes, scene["uniforms"], scene["push"], pyr, t)
            r.shade_opaque(planes, scene["uniforms"], scene["push"], o, None)
            torch.cuda.synchronize()
            for got, key in ((t, "spirv_fragment_transmission"), (o, "spirv_fragment_hdr")):
                want = z[key]
                if dt == torch.float16:
                    with np.errstate(over="ignore"):
                        want = want.astype(np.float16).astype(np.float32)
                report.append(line(f"{key[6:]} {str(dt)[6:]}", got[iy, ix].cpu().numpy().astype(np.float32)[None], want[None]))
            del t, o
        r.upload_textures([])
        del planes, pyr
        torch.cuda.empty_cache()
    for path in GOLDEN:
        z = np.load(path)
        scene, g, w, h = _scene_from_fixture(z)
        print(os.path.basename(path), f"{w}x{h}")
        upload(r, scene)
        r.upload_textures(scene.get("textures", []))
        planes = GBufferPlanes.from_numpy(g, r.device)
        tex = oracle.new_pyramid(w, h, z["opaque_mip0"])
        oracle.generate_mips(w, h, tex)
        pyr = OpaquePyramid(w, h, r.device)
        pyr.texels.copy_(torch.from_numpy(tex).to(r.device))
        ys, xs = z["pixels"][:, 0], z["pixels"][:, 1]
        for dt in (torch.float32, torch.float16):
            t = torch.zeros((h, w, 4), dtype=dt, device=r.device)
            o = torch.zeros((h, w, 4), dtype=dt, device=r.device)
            r.shade_transmission(planes, scene["uniforms"], scene["push"], pyr, t)
            r.shade_opaque(planes, scene["uniforms"], scene["push"], o, None)
            torch.cuda.synchronize()
            for got, key in ((t, "spirv_fragment_transmission"), (o, "spirv_fragment_hdr")):
                want = z[key]
                if dt == torch.float16:
                    with np.errstate(over="ignore"):
                        want = want.astype(np.float16).astype(np.float32)
                report.append(line(f"{key[6:]} {str(dt)[6:]}", got.cpu().numpy().astype(np.float32)[ys, xs][None], want[None]))
        r.upload_textures([])

    cases = [(256, 256, 2, "full", None), (250, 130, 3, "holes", None), (192, 108, 4, "full", 0.25), (64, 64, 0, "full", None),
             (70, 3, 1, "full", None), (1920, 1080, 1, "full", None), (960, 540, 4, "full", 0.25)]
    for (w, h, nl, cov, rough) in cases:
        print(f"synthetic {w}x{h} lights={nl} coverage={cov} roughness_override={rough}")
        scene = synthetic.make_scene(w, h, num_point_lights=nl, coverage=cov, roughness_override=rough)
        upload(r, scene)
        planes = GBufferPlanes.from_numpy(scene["gbuffer"], r.device)
        b = oracle.SceneBinding(scene, lut)
        tex = oracle.new_pyramid(w, h, synthetic.make_opaque_mip0(w, h))
        oracle.generate_mips(w, h, tex)
        pyr = OpaquePyramid(w, h, r.device)
        pyr.texels.copy_(torch.from_numpy(tex).to(r.device))
        t32 = torch.zeros((h, w, 4), dtype=torch.float32, device=r.device)
        t16 = torch.zeros((h, w, 4), dtype=torch.float16, device=r.device)
        o32 = torch.zeros((h, w, 4), dtype=torch.float32, device=r.device)
        r.shade_transmission(planes, scene["uniforms"], scene["push"], pyr, t32)
        r.shade_transmission(planes, scene["uniforms"], scene["push"], pyr, t16)
        r.shade_opaque(planes, scene["uniforms"], scene["push"], o32, None)
        torch.cuda.synchronize()
        nth = os.cpu_count() or 8
        w16, w32 = oracle.shade_transmission(b, scene["gbuffer"], tex, nthreads=nth)
        _, w64 = oracle.shade_transmission(b, scene["gbuffer"], tex, nthreads=nth, fp64=True)
        _, wo32, _ = oracle.shade_opaque(b, scene["gbuffer"], nthreads=nth)
        report.append(line("transmission f32 vs oracle32", t32.cpu().numpy(), w32))
        report.append(line("transmission f16 vs RTNE(oracle32)", t16.cpu().numpy().astype(np.float32), w16.astype(np.float32)))
        report.append(line("DISPLAY-referred f16 frames (raw counts)", display(t16.cpu().numpy()), display(w16)))
        report.append(line("transmission f32 vs oracle64", t32.cpu().numpy(), w64))
        report.append(line("oracle32 vs oracle64", w32, w64))
        report.append(line("opaque f32 vs oracle32", o32.cpu().numpy(), wo32))
    r.close()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "parity_report.json"), "w") as f:
        json.dump(report, f, indent=1)


if __name__ == "__main__":
    main()
