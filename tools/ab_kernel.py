#!/usr/bin/env python3
"""A/B timing of kernel variants on ONE box: python tools/ab_kernel.py [--lights N] [--rounds R] lib_a.so lib_b.so ...
Each variant is timed in its own child process (the library is loaded once per process), variants interleaved over
R rounds; prints the per-variant median of the average launch time over back-to-back batches of the 4K transmissive
pass at steady-state clocks.
Experiments only: the product always loads transmission_renderer_amd/libtr_shade.so."""
import json, os, statistics, subprocess, sys

CHILD = r'''
import ctypes as C, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.environ["TR_ROOT"])
from transmission_renderer_amd import _lib
_lib.LIB_PATH = os.environ["TR_AB_LIB"]
class _Tolerant(C.CDLL):   # builds of older commits lack the newer entry points: bind what exists (experiments only)
    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            if name.startswith("tr_") and name != "tr_debug_read_timing":
                return type("missing", (), {})()
            raise
C.CDLL = _Tolerant
from transmission_renderer_amd import synthetic, wire
from transmission_renderer_amd.renderer import GBufferPlanes, OpaquePyramid, TransmissionRenderer
w, h, nl = 3840, 2160, int(os.environ.get("TR_AB_LIGHTS", "1"))
r = TransmissionRenderer(0)
scene = synthetic.make_scene(w, h, num_point_lights=nl, roughness_override=float(os.environ["TR_AB_ROUGHNESS"]) if os.environ.get("TR_AB_ROUGHNESS") else None)
if os.environ.get("TR_AB_ALLT"):
    for m in scene["materials"]: m.transmission_factor = 1.0
r.upload_ggx_lut(); r.upload_materials(scene["materials"]); r.upload_lights(scene["lights"])
r.set_cluster_tables(torch.from_numpy(scene["cluster_counts"].view(np.int32)).to(r.device), torch.from_numpy(scene["light_indices"].view(np.int32)).to(r.device))
g = GBufferPlanes.from_numpy(scene["gbuffer"], r.device)
pyr = OpaquePyramid(w, h, r.device)
pyr.level(0).copy_(torch.from_numpy(synthetic.make_opaque_mip0(w, h)).to(r.device)); r.generate_mips(pyr)
hdr = torch.zeros((h, w, 4), dtype=torch.float16, device=r.device)
# steady state: the GPU's clocks ramp over the first ~10 ms of continuous load (a launch + sync per step never gets
# there: 4K launches then take ~130 us instead of ~100), so warm up for 100 ms and time back-to-back batches
import time
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.1:
    for _ in range(16): r.shade_transmission(g, scene["uniforms"], scene["push"], pyr, hdr)
    torch.cuda.synchronize()
ts = []
batch = 50
for _ in range(int(os.environ.get("TR_AB_STEPS", "400")) // batch):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(batch): r.shade_transmission(g, scene["uniforms"], scene["push"], pyr, hdr)
    b.record(); b.synchronize()
    ts.append(a.elapsed_time(b) * 1e3 / batch)
ts.sort()
res = {"p50": ts[len(ts) // 2], "p10": ts[0], "min": ts[0]}
try:   # -DTR_TIMING=1 builds: per-wave wait cycles (100 MHz s_memtime ticks) of one launch
    fn = r.lib.tr_debug_read_timing
    buf = (C.c_ulonglong * 12)()
    fn(buf)
    r.shade_transmission(g, scene["uniforms"], scene["push"], pyr, hdr); torch.cuda.synchronize()
    fn(buf)
    res["timing"] = list(buf)
except AttributeError:
    pass
print(json.dumps(res))
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
            if os.environ.get("TR_TIMING_DUMP"): print(out.stderr[-3000:])
            res[lib].append(json.loads(line[-1]))
            if "timing" in res[lib][-1]:
                w0, w1, w2, total, tiles, waves = res[lib][-1]["timing"][:6]
                longest, waves = waves >> 32, waves & 0xFFFFFFFF
                print(f"  {os.path.basename(lib)}: per tile and wave (s_memtime ticks): planes wait {w0 / tiles:.1f}, cluster lists {w1 / tiles:.1f}, "
                      f"taps+LUT {w2 / tiles:.1f}, whole tile {total / tiles:.1f}; {tiles / waves:.2f} tiles per wave, {waves} waves, longest-lived wave {longest} ticks, mean {total / waves:.0f}; "
                      f"most loaded XCD {res[lib][-1]['timing'][6] / 1000:.3f} of the mean; shader clock while the waves ran {total / max(res[lib][-1]['timing'][7], 1) * 100:.0f} MHz")
    for lib, rs in res.items():
        if rs:
            print(f"{os.path.basename(lib):40s} p50 {statistics.median(r['p50'] for r in rs):7.1f} us  p10 {statistics.median(r['p10'] for r in rs):7.1f}  min {min(r['min'] for r in rs):7.1f}   ({[round(r['p50'], 1) for r in rs]})")


if __name__ == "__main__":
    main()
