#!/usr/bin/env python3
"""Config 2 (1920x1080): the transmissive pass as 1 / 2 / 3 row bands on as many streams, and the opaque -> mips -> transmissive
frame through `record` with level 1 written by the opaque pass (tr_shade_opaque_pyramid) against the round-5 sequence
(tr_shade_opaque + tr_generate_mips).  Cold inputs (rotating sets), medians of 5 x 200 steps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

def med(wl, fn=None, K=200):
    return sorted(wl.timed(K, fn=fn, first=i * K)[0] * 1e3 for i in range(5))[2]

for w, h in ((1920, 1080), (2560, 1440)):
    for split in (1, 2, 3):
        wl = bench.PassWorkload(0, w, h, split=split)
        wl.ramp(0.15)
        print(f"{w}x{h} transmissive pass, {split} band(s): {med(wl):6.1f} us", flush=True)
        if split == 1:
            def old(k, rotate=True):
                g, pyr, hdr = wl.sets[k % len(wl.sets)]
                wl.r.shade_opaque(g, wl.scene["uniforms"], wl.scene["push"], hdr, pyr)
                wl.r.generate_mips(pyr)
                wl.r.shade_transmission(g, wl.scene["uniforms"], wl.scene["push"], pyr, hdr)
            def new(k, rotate=True):
                g, pyr, hdr = wl.sets[k % len(wl.sets)]
                wl.r.record(g, g, wl.scene["uniforms"], wl.scene["push"], hdr, pyr)
            def opaque_only(k, rotate=True):
                g, pyr, hdr = wl.sets[k % len(wl.sets)]
                wl.r.shade_opaque(g, wl.scene["uniforms"], wl.scene["push"], hdr, pyr)
            def mips_only(k, rotate=True):
                g, pyr, hdr = wl.sets[k % len(wl.sets)]
                wl.r.generate_mips(pyr)
            def mips_from2(k, rotate=True):
                g, pyr, hdr = wl.sets[k % len(wl.sets)]
                wl.r.generate_mips_from(pyr, 2)
            for name, fn in (("opaque + mips + transmissive (round 5)", old), ("record: opaque writes level 1", new), ("opaque pass alone", opaque_only),
                             ("mip chain alone", mips_only), ("mip chain from level 2", mips_from2)):
                wl.timed(20, fn=fn)
                print(f"   {name}: {med(wl, fn):6.1f} us", flush=True)
        wl.close()
        torch.cuda.empty_cache()
