#!/usr/bin/env python3
"""The transmissive kernel's punctual-light loop as the compiler emitted it for gfx950, every instruction with the issue
cost class tools/isa_regions.py charges it (wave64 issue cycles measured on MI355X, tools/ubench/valu_rate.hip):
    python tools/dump_light_loop_isa.py > profiles/rNN/config3_light_loop_isa.txt
The loop is found as the innermost backward branch whose body holds instructions of eval_punctual (line tables)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as g  # noqa: E402
import isa_regions as ir  # noqa: E402

KERNEL = "_ZN2tr12shade_kernelILb1E15HIP_vector_typeIjLj2EELi0ELb0EEEvNS_9tr_launchE"


def main():
    asm = os.path.join(ROOT, "build_ab", "tr_shade_lines.s")
    os.makedirs(os.path.dirname(asm), exist_ok=True)
    flags = [f for f in g.HIPCC_FLAGS if f not in ("-fPIC", "-shared")]
    subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["-gline-tables-only", "-S", "--cuda-device-only", "-o", asm,
                                                      os.path.join(g.CSRC, "tr_shade.hip")], check=True, stderr=subprocess.DEVNULL)
    s = open(asm).read()
    files = {int(m.group(1)): os.path.basename(m.group(3)) for m in re.finditer(r'^\s*\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"', s, re.M)}
    start = s.index(KERNEL + ":")
    body = s[start:s.index("s_endpgm", start)].split("\n")
    table = ir.regions_of(os.path.join(g.CSRC, "tr_kernels.h"))
    # basic blocks: a label or the instruction after a branch starts one
    blocks, cur, region = [], None, "entry"

    def new_block(name):
        nonlocal cur
        cur = {"name": name, "rows": [], "succ": []}
        blocks.append(cur)

    new_block("entry")
    for l in body:
        t = l.strip()
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
        if m:
            if files.get(int(m.group(1))) == "tr_kernels.h":
                region = "?"
                for first, name in table:
                    if int(m.group(2)) >= first:
                        region = name
            continue
        m = re.match(r"^(\.LBB\w+):", t)
        if m:
            prev = cur
            new_block(m.group(1))
            if not prev["rows"] or not prev["rows"][-1][0].startswith("s_branch"):
                prev["succ"].append(m.group(1))
            continue
        if not t or t.startswith((".", ";", "//")) or t.endswith(":"):
            continue
        ins = t.split(";")[0].strip()
        cur["rows"].append((ins, region))
        m = re.match(r"s_(c?branch)\w*\s+(\.LBB\w+)", ins)
        if m:
            cur["succ"].append(m.group(2))
            prev = cur
            new_block(f"{prev['name']}+")
            if m.group(1) == "cbranch":
                prev["succ"].append(cur["name"])
    index = {b["name"]: i for i, b in enumerate(blocks)}
    succ = [[index[x] for x in b["succ"] if x in index] for b in blocks]
    # natural loops: for every edge u -> h, the blocks that reach u without passing h; the light loop is the smallest one
    # (that does not swallow the kernel's entry) holding instructions of eval_punctual
    pred = [[] for _ in blocks]
    for u, ss in enumerate(succ):
        for v in ss:
            pred[v].append(u)
    loops = []
    for u, ss in enumerate(succ):
        for h in ss:
            body_set, work = {h}, [u]
            while work:
                x = work.pop()
                if x in body_set:
                    continue
                body_set.add(x)
                work.extend(pred[x])
            if 0 not in body_set and any(r == "eval_punctual" for i in body_set for _, r in blocks[i]["rows"]):
                loops.append(sorted(body_set))
    loops.sort(key=len)
    comp = loops[0]
    print(f"# {KERNEL}")
    print(f"# the punctual-light loop (one trip per light of the cluster's list) as emitted for gfx950: the {len(comp)} basic blocks of its cycle,")
    print("# in layout order (a trip takes one side of each material / lobe condition: the counts below are of the whole body).")
    print("# cost = wave64 issue cycles charged per class: pair 2.35 (dual-issued VGPR-operand fp32), sgpr 4.0 (an SGPR / literal source),")
    print("#        half 4.1 (conversions, compares, v_fma_mix), int 3.5, trans 8.1; scalar / memory instructions issue beside the vector port")
    tot, by = 0.0, {}
    for i in comp:
        print(f"# block {blocks[i]['name']} -> {', '.join(blocks[i]['succ'])}")
        for t, r in blocks[i]["rows"]:
            c, k = ir.cost(t)
            tot += c
            by[k] = by.get(k, 0) + 1
            print(f"{c:5.2f} {k:6s} {r:14s} {t}")
    nv = sum(v for k, v in by.items() if k in ("pair", "sgpr", "half", "int", "trans"))
    print(f"# whole body: {nv} vector instructions, {tot:.0f} vector issue cycles; classes {by}")


if __name__ == "__main__":
    main()
