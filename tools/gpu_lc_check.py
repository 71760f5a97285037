#!/usr/bin/env python3
"""Loader/consumer variant of the untextured transmissive plane pass against shade_kernel, bit for bit on whole frames.
    python tools/gpu_lc_check.py LIB_A LIB_B      (LIB_B runs with TR_LC=2 in its child's environment)
Each library renders the bench's synthetic frames (several sizes, bands, all-transmissive, config 3) in its own child
process and writes a digest per frame; the parent compares.  Experiments only (build_ab/ libraries)."""
import hashlib, json, os, subprocess, sys

CHILD = r'''
import hashlib, json, os, sys
sys.path.insert(0, os.environ["TR_ROOT"])
from transmission_renderer_amd import _lib
_lib.LIB_PATH = os.environ["TR_AB_LIB"]
import torch, bench
out = {}
for name, (w, h, kw, part) in dict(
        p514=(514, 290, {}, None), p1080=(1920, 1080, {}, None), p4k=(3840, 2160, {}, None),
        band=(3840, 2160, {}, (0, 1080, 3840, 2160)), odd=(1283, 721, {}, (2, 4, 1283, 700)),
        allt=(1920, 1080, dict(all_transmissive=True), None), c3=(1920, 1080, dict(lights=4, roughness=0.25), None)).items():
    wl = bench.PassWorkload(0, w, h, sets=1, split=1, **kw)
    g, pyr, frame = wl.sets[0]
    frame.fill_(0.25)
    for _ in range(2):
        wl.r.shade_transmission(g, wl.scene["uniforms"], wl.scene["push"], pyr, frame, part)
    torch.cuda.synchronize()
    b = frame.cpu().numpy().tobytes()
    out[name] = [hashlib.sha1(b).hexdigest(), float(frame[..., :3].float().abs().mean().item())]
    wl.close()
print(json.dumps(out))
'''

def run(lib, lc):
    env = dict(os.environ, TR_ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), TR_AB_LIB=os.path.abspath(lib))
    if lc: env["TR_LC"] = lc
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=900)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(lib, "FAILED", r.stderr[-1500:]); sys.exit(1)
    return json.loads(line[-1])

a, b = run(sys.argv[1], None), run(sys.argv[2], os.environ.get("TR_LC_CHECK", "2"))
bad = 0
for k in a:
    same = a[k][0] == b[k][0]
    bad += not same
    print(f"{k:6s} {'identical' if same else 'DIFFERENT'}  mean |rgb| {a[k][1]:.5f} / {b[k][1]:.5f}")
sys.exit(1 if bad else 0)
