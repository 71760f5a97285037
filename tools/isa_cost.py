#!/usr/bin/env python3
"""Issue-cycle estimate per basic block of one kernel's assembly (tools/extract_kernel_asm.py output), from the
per-class wave64 issue costs measured on MI355X (DESIGN.md 3.1 / tools/ubench/valu_rate.hip):
pairable fp32 2.35, any VALU with an SGPR source 4.0, half-rate class 4.1, integer 3.5, transcendental 8.1.
python tools/isa_cost.py build_ab/headline2.s"""
import re
import sys

HALF = ("v_fma_mix", "v_cvt", "v_floor", "v_cmp", "v_max3", "v_fract", "v_med3", "v_perm", "v_readlane", "v_readfirstlane",
        "v_cndmask", "v_mad_mix", "v_min3")
INT = ("v_add_u32", "v_sub_u32", "v_subrev_u32", "v_lshl", "v_lshr", "v_and", "v_or", "v_mul_u32", "v_mad_u32", "v_add3",
       "v_add_lshl", "v_lshl_or", "v_and_or", "v_min_u32", "v_max_u32", "v_addc", "v_subb", "v_xor", "v_bfe", "v_mul_hi",
       "v_mul_lo", "v_ashr", "v_mov_b64")
TRANS = ("v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32")


def cost(line):
    op = line.split()[0]
    if not op.startswith("v_"):
        return 0.0, "other"
    if op.startswith(TRANS):
        return 8.1, "trans"
    operands = line.split(None, 1)[1] if len(line.split(None, 1)) > 1 else ""
    src = operands.split(",", 1)[1] if "," in operands else ""
    sgpr = bool(re.search(r"(?<![a-z_])-?\|?s\d+|s\[\d+:\d+\]|vcc|exec", src)) and not op.startswith(("v_cndmask", "v_addc", "v_subb"))
    if op.startswith(HALF):
        return 4.1, "half"
    if op.startswith(INT):
        return max(3.5, 4.0 if sgpr else 0), "int"
    if sgpr:
        return 4.0, "sgpr"
    return 2.35, "pair"


blocks, cur = [], None
for raw in open(sys.argv[1]):
    l = raw.rstrip("\n")
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m or l.startswith("; %bb."):
        cur = {"name": m.group(1) if m else l.strip("; ").split()[0], "cyc": 0.0, "n": 0, "cls": {}, "first": None}
        blocks.append(cur)
        continue
    if cur is None:
        cur = {"name": "entry", "cyc": 0.0, "n": 0, "cls": {}, "first": None}
        blocks.append(cur)
    if l.startswith("\t") and l.strip() and not l.strip().startswith((".", ";")):
        c, k = cost(l.strip())
        if c:
            cur["cyc"] += c
            cur["n"] += 1
            cur["cls"][k] = cur["cls"].get(k, 0) + 1
tot = 0
for b in blocks:
    if b["n"]:
        print(f'{b["name"]:12s} valu {b["n"]:4d}  cycles {b["cyc"]:7.1f}  {b["cls"]}')
        tot += b["cyc"]
print("static total", round(tot, 1))
