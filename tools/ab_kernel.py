#!/usr/bin/env python3
"""A/B timing of kernel variants on ONE box: python tools/ab_kernel.py [--lights N] [--rounds R] lib_a.so lib_b.so ...
Each variant is timed in its own child process (the library is loaded once per process), variants interleaved over
R rounds; prints the per-variant median of the per-launch HIP-event times of the 4K transmissive pass.
Experiments only: the product always loads transmission_renderer_amd/libtr_shade.so."""
import json, os, statistics, subprocess, sys

CHILD = r'''
import ctypes as C, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.environ["TR_ROOT"])
from transmission_renderer_amd import _lib
_lib.LIB_PATH = os.environ["TR_AB_LIB"]
from transmission_renderer_amd import synthetic, wire
from transmission_renderer_amd.renderer import GBufferPlanes, OpaquePyramid, TransmissionRenderer
w, h, nl = 3840, 2160, int(os.environ.get("TR_AB_LIGHTS", "1"))
r = TransmissionRenderer(0)
scene = synthetic.make_scene(w, h, num_point_lights=nl)
r.upload_ggx_lut(); r.upload_materials(scene["materials"]); r.upload_lights(scene["lights"])
r.set_cluster_tables(torch.from_numpy(scene["cluster_counts"].view(np.int32)).to(r.device), torch.from_numpy(scene["light_indices"].view(np.int32)).to(r.device))
g = GBufferPlanes.from_numpy(scene["gbuffer"], r.device)
pyr = OpaquePyramid(w, h, r.device)
pyr.level(0).copy_(torch.from_numpy(synthetic.make_opaque_mip0(w, h)).to(r.device)); r.generate_mips(pyr)
hdr = torch.zeros((h, w, 4), dtype=torch.float16, device=r.device)
for _ in range(20): r.shade_transmission(g, scene["uniforms"], scene["push"], pyr, hdr)
torch.cuda.synchronize()
ts = []
for _ in range(int(os.environ.get("TR_AB_STEPS", "200"))):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); r.shade_transmission(g, scene["uniforms"], scene["push"], pyr, hdr); b.record(); b.synchronize()
    ts.append(a.elapsed_time(b) * 1e3)
ts.sort()
print(json.dumps({"p50": ts[len(ts) // 2], "p10": ts[len(ts) // 10], "min": ts[0]}))
'''


def main():
    args = sys.argv[1:]
    lights, rounds = "1", 3
    while args and args[0].startswith("--"):
        if args[0] == "--lights": lights = args[1]
        if args[0] == "--rounds": rounds = int(args[1])
        args = args[2:]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {a: [] for a in args}
    for _ in range(rounds):
        for lib in args:
            env = dict(os.environ, TR_ROOT=root, TR_AB_LIB=os.path.abspath(lib), TR_AB_LIGHTS=lights)
            out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
            line = [l for l in out.stdout.splitlines() if l.startswith("{")]
            if not line:
                print(lib, "FAILED", out.stderr[-400:]); continue
            res[lib].append(json.loads(line[-1]))
    for lib, rs in res.items():
        if rs:
            print(f"{os.path.basename(lib):40s} p50 {statistics.median(r['p50'] for r in rs):7.1f} us  p10 {statistics.median(r['p10'] for r in rs):7.1f}  min {min(r['min'] for r in rs):7.1f}   ({[round(r['p50'], 1) for r in rs]})")


if __name__ == "__main__":
    main()
