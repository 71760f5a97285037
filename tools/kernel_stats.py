#!/usr/bin/env python3
"""Register / LDS / spill figures of every kernel of the library (device-only compile to assembly, the metadata notes
parsed): python tools/kernel_stats.py [pattern] [-D...]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

args = sys.argv[1:]
pat = args[0] if args and not args[0].startswith("-") else ""
extra = [a for a in args if a.startswith("-")]
os.makedirs(os.path.join(ROOT, "build_ab"), exist_ok=True)
asm = os.path.join(ROOT, "build_ab", "tr_shade_stats.s")
flags = [f for f in g.HIPCC_FLAGS if f not in ("-fPIC", "-shared")]
cmd = ["/opt/rocm/bin/hipcc"] + flags + extra + ["-S", "--cuda-device-only", "-o", asm, os.path.join(g.CSRC, "tr_shade.hip")]
subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
s = open(asm).read()
for b in s.split("- .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", b).group(1)
    if pat not in name:
        continue
    f = lambda k: re.search(r"\.%s:\s+(\d+)" % k, b).group(1)
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    print(f"{dem[:80]:<80} vgpr {f('vgpr_count'):>3} sgpr {f('sgpr_count'):>3} spill v{f('vgpr_spill_count')}/s{f('sgpr_spill_count')} "
          f"lds {f('group_segment_fixed_size'):>6} scratch {f('private_segment_fixed_size')}")
