#!/usr/bin/env python3
"""Cuts one kernel out of build_ab/tr_shade_stats.s (tools/kernel_stats.py writes it) and prints its instruction
histogram: python tools/extract_kernel_asm.py MANGLED_SUBSTRING [out.s]"""
import collections
import re
import sys

s = open("build_ab/tr_shade_stats.s").read()
pat = sys.argv[1]
m = re.search(r"^(\S*%s\S*):" % re.escape(pat), s, re.M)
i = m.start()
j = s.index("s_endpgm", i)
body = s[i:j]
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(body)
ins = [l.strip().split()[0] for l in body.split("\n") if l.startswith("\t") and l.strip() and not l.strip().startswith((".", ";"))]
c = collections.Counter(ins)
g = collections.Counter()
for k, v in c.items():
    grp = ("valu_trans" if k.split("_e")[0] in ("v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32") else
           "valu" if k.startswith("v_") else "smem" if k.startswith("s_load") else "salu" if k.startswith("s_") else
           "vmem" if k.startswith(("global_", "buffer_", "flat_", "scratch_")) else "lds" if k.startswith("ds_") else "other")
    g[grp] += v
print(m.group(1)[:80], "total", len(ins), dict(g))
