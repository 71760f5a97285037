#!/usr/bin/env python3
"""Localises the error of tests/test_gpu_raster.py::test_rasterize_then_shade_end_to_end: which pixels of the final
RGBA16F frame differ from the oracle's, by pass, material, silhouette and derivative state.  GPU box only."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import oracle  # noqa: E402
from transmission_renderer_amd import meshes, synthetic, wire  # noqa: E402
from transmission_renderer_amd.renderer import OpaquePyramid, TransmissionRenderer, load_ggx_lut  # noqa: E402
from test_gpu_raster import _oracle_layers, _scene  # noqa: E402


def nerr(got, want):
    got, want = got.astype(np.float64), want.astype(np.float64)
    with np.errstate(invalid="ignore"):
        return (got - want) / np.maximum(np.abs(want), 1.0)


def stats(name, e, mask):
    m = mask & np.isfinite(e).all(axis=2)
    if not m.any():
        print(f"  {name:<46} no pixels")
        return
    x = e[m][:, :3]
    print(f"  {name:<46} {m.sum():6d} px  rmse {np.sqrt((x ** 2).mean()):.3e}  max {np.abs(x).max():.3e}  "
          f">1e-3: {(np.abs(x).max(axis=1) > 1e-3).sum()}")


def main():
    lut = load_ggx_lut()
    r = TransmissionRenderer(0)
    r.upload_ggx_lut(lut)
    w, h = 320, 180
    view = wire.default_camera()[1]
    geo = meshes.make_mesh_scene()
    sc = _scene(w, h, view, alpha_cutoffs=(0.0, 0.0))
    sc["lights"] = synthetic.make_lights(2)
    sc["cluster_counts"], sc["light_indices"] = synthetic.all_lights_cluster_tables(2)
    (want_o, want_t), culling = _oracle_layers(geo, sc, w, h, view)
    r.upload_materials(sc["materials"])
    r.upload_textures(sc["textures"])
    r.upload_lights(sc["lights"])
    r.set_cluster_tables(torch.from_numpy(sc["cluster_counts"].view(np.int32)).to(r.device),
                         torch.from_numpy(sc["light_indices"].view(np.int32)).to(r.device))
    r.upload_geometry(geo)
    o, t = r.new_layer(w, h), r.new_layer(w, h)
    r.draw_scene(culling, sc["push"], o, t)
    pyr = OpaquePyramid(w, h, r.device)
    # pass by pass, fp32 targets
    op32 = torch.zeros((h, w, 4), dtype=torch.float32, device=r.device)
    r.shade_opaque(o, sc["uniforms"], sc["push"], op32, None)
    op16 = torch.zeros((h, w, 4), dtype=torch.float16, device=r.device)
    r.shade_opaque(o, sc["uniforms"], sc["push"], op16, pyr)
    r.generate_mips(pyr)
    tr32 = op32.clone()
    r.shade_transmission(t, sc["uniforms"], sc["push"], pyr, tr32)
    torch.cuda.synchronize()
    b = oracle.SceneBinding(sc, lut)
    for layer in (want_o, want_t):
        layer["width"], layer["height"] = w, h
    for fp64 in (False, True):
        print("oracle fp64" if fp64 else "oracle fp32")
        o16, o32, mip0 = oracle.shade_opaque(b, want_o, nthreads=8, fp64=fp64)
        tex = oracle.new_pyramid(w, h, mip0)
        oracle.generate_mips(w, h, tex)
        base = o32.astype(np.float64 if fp64 else np.float32).copy()
        _, t32 = oracle.shade_transmission(b, want_t, tex, hdr_f16=o16.copy(), hdr_f32=base, nthreads=8, fp64=fp64)
        mo, mt = want_o["material_id"], want_t["material_id"]
        cov_o, cov_t = mo != wire.NOT_COVERED, mt != wire.NOT_COVERED
        eo = nerr(op32.cpu().numpy(), o32)
        print("  pyramid bytes equal:", np.array_equal(pyr.texels.cpu().numpy().view(np.uint16), tex.view(np.uint16)),
              " differing texels:", int((pyr.texels.cpu().numpy().view(np.uint16) != tex.view(np.uint16)).any(axis=-1).sum()))
        stats("opaque pass, all covered", eo, cov_o)
        nonfin = ~np.isfinite(o32).all(axis=2)
        print("  oracle non-finite opaque px:", int(nonfin.sum()), " gpu non-finite:", int((~np.isfinite(op32.cpu().numpy()).all(axis=2)).sum()))
        # silhouettes: a quad partner with another material / no fragment
        def sil(m):
            s = np.zeros_like(m, dtype=bool)
            xa = m[:, 0::2] != m[:, 1::2]
            s[:, 0::2] |= xa
            s[:, 1::2] |= xa
            ya = m[0::2, :] != m[1::2, :]
            s[0::2, :] |= ya
            s[1::2, :] |= ya
            return s
        stats("opaque, quad spans materials", eo, cov_o & sil(mo))
        stats("opaque, quad uniform", eo, cov_o & ~sil(mo))
        for m in np.unique(mo[cov_o]):
            stats(f"opaque material {m}", eo, mo == m)
        et = nerr(tr32.cpu().numpy(), t32)
        stats("transmissive pass, covered", et, cov_t)
        stats("transmissive, quad spans materials", et, cov_t & sil(mt))
        stats("transmissive, quad uniform", et, cov_t & ~sil(mt))
        for m in np.unique(mt[cov_t]):
            stats(f"transmissive material {m}", et, mt == m)
        # worst pixels
        bad = np.argwhere((np.abs(et).max(axis=2) > 1e-3) & cov_t)[:12]
        for (y, x) in bad:
            print(f"    t px ({y},{x}) mat {mt[y, x]} gpu {tr32[y, x, :3].cpu().numpy()} oracle {t32[y, x, :3]}")
        bad = np.argwhere((np.abs(eo).max(axis=2) > 1e-3) & cov_o)[:12]
        for (y, x) in bad:
            print(f"    o px ({y},{x}) mat {mo[y, x]} gpu {op32[y, x, :3].cpu().numpy()} oracle {o32[y, x, :3]}")
    r.close()


if __name__ == "__main__":
    main()
