#!/usr/bin/env python3
"""A/B of library builds on ONE box with the bench's own step: COLD inputs (rotating input sets), the 4K transmissive pass
as bench.py issues it (two bands on two streams = the metric) and as ONE tr_shade_transmission call per frame, optionally
the all-transmissive scene and config 3.
    python tools/ab_cold.py [--rounds R] [--configs head,single,allt,c3,p1080,p8k] lib_a.so lib_b.so ...
Each (variant, config) is timed in its own child process, variants interleaved over R rounds; prints medians.
Experiments only: the product always loads transmission_renderer_amd/libtr_shade.so."""
import json, os, statistics, subprocess, sys

CHILD = r'''
import json, os, sys
sys.path.insert(0, os.environ["TR_ROOT"])
from transmission_renderer_amd import _lib
_lib.LIB_PATH = os.environ["TR_AB_LIB"]
import bench
cfg = os.environ["TR_AB_CFG"]
kw = dict(head={}, single={}, allt=dict(all_transmissive=True), allt1=dict(all_transmissive=True), c3=dict(lights=4, roughness=0.25),
          p1080={}, p8k={})[cfg]
w, h = dict(p1080=(1920, 1080), p8k=(7680, 4320)).get(cfg, (3840, 2160))
wl = bench.PassWorkload(0, w, h, split=1 if cfg in ("single", "allt1") else 0, **kw)
wl.ramp(0.15)
K = int(os.environ.get("TR_AB_STEPS", "200"))
ts = sorted(wl.timed(K, first=i * K)[1] * 1e3 for i in range(5))
print(json.dumps({"p50": ts[2], "min": ts[0]}))
'''


def main():
    args = sys.argv[1:]
    rounds, configs = 2, ["head", "single"]
    while args and args[0].startswith("--"):
        if args[0] == "--rounds": rounds = int(args[1])
        if args[0] == "--configs": configs = args[1].split(",")
        args = args[2:]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {(a, c): [] for a in args for c in configs}
    for _ in range(rounds):
        for cfg in configs:
            for lib in args:
                env = dict(os.environ, TR_ROOT=root, TR_AB_LIB=os.path.abspath(lib), TR_AB_CFG=cfg)
                out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
                line = [l for l in out.stdout.splitlines() if l.startswith("{")]
                if not line:
                    print(lib, cfg, "FAILED", out.stderr[-600:]); continue
                res[(lib, cfg)].append(json.loads(line[-1]))
    for lib in args:
        cells = []
        for cfg in configs:
            rs = res[(lib, cfg)]
            if rs:
                cells.append(f"{cfg} {statistics.median(r['p50'] for r in rs):6.1f} (min {min(r['min'] for r in rs):6.1f})")
        print(f"{os.path.basename(lib):36s} " + "  ".join(cells), flush=True)


if __name__ == "__main__":
    main()
