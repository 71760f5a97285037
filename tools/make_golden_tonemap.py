#!/usr/bin/env python3
"""tests/golden/spirv_tonemap.npz: the reference's compiled fragment_tonemap.spv executed by oracle/spirv_ref
on a spread of HDR colours (authoring container only; fixtures hold inputs + outputs)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.spirv_ref import spirv_interp as si
from tools.make_golden_spirv import LibmInterp
import ctypes as C
from transmission_renderer_amd import _lib, wire


def main():
    lib = _lib.load()
    lp = wire.LottesParams(); lib.tr_lottes_defaults(C.byref(lp))
    bp = wire.TonemapParams(); lib.tr_bake_lottes_params(C.byref(lp), C.byref(bp))
    rng = np.random.default_rng(2024)
    cols = np.concatenate([
        (rng.random((600, 3)) * np.float32(10.0) ** rng.integers(-4, 3, (600, 1))).astype(np.float16),
        np.array([[0, 0, 0], [1, 1, 1], [8, 8, 8], [100, 0.5, 0.01], [0.18, 0.18, 0.18], [0, 0.5, 0], [65504, 1, 1]], dtype=np.float16),
    ]).astype(np.float16)
    mod = si.Module("/root/reference/compiled-shaders/normal/fragment_tonemap.spv")
    outs = []
    for c in cols:
        texel = [float(c[0]), float(c[1]), float(c[2]), 1.0]
        it = LibmInterp(mod, "fragment_tonemap", {}, bytes(bp) + bytes(4), {0: [0.5, 0.5]},
                        sample=lambda kind, image, sampler, coord, lod, t=texel: t)
        outs.append(np.array(it.run()[0], dtype=np.float32))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "spirv_tonemap.npz"), hdr=cols,
                        params=np.frombuffer(bytes(bp), dtype=np.float32), lottes=np.frombuffer(bytes(lp), dtype=np.float32),
                        spirv_out=np.stack(outs))
    print("tonemap golden:", len(cols), "colours; black ->", outs[600], " white ->", outs[601])


if __name__ == "__main__":
    main()
