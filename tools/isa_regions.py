#!/usr/bin/env python3
"""Static issue-cycle estimate of one kernel split by SOURCE REGION: compiles the library's device code with line tables
(-gline-tables-only), walks the kernel's assembly and charges every instruction (per-class wave64 issue costs measured on
MI355X, tools/isa_cost.py) to the region its `.loc` falls in.  Instructions of the small helpers (tr_common.h, HIP math
headers) are charged to the last region seen.  Static counts: loops count once, both sides of a branch count.

    python tools/isa_regions.py MANGLED_SUBSTRING [-DFOO ...]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as g  # noqa: E402

HALF = ("v_fma_mix", "v_cvt", "v_floor", "v_cmp", "v_max3", "v_fract", "v_med3", "v_perm", "v_readlane", "v_readfirstlane",
        "v_cndmask", "v_mad_mix", "v_min3")
INT = ("v_add_u32", "v_sub_u32", "v_subrev_u32", "v_lshl", "v_lshr", "v_and", "v_or", "v_mul_u32", "v_mad_u32", "v_add3",
       "v_add_lshl", "v_lshl_or", "v_and_or", "v_min_u32", "v_max_u32", "v_addc", "v_subb", "v_xor", "v_bfe", "v_mul_hi",
       "v_mul_lo", "v_ashr", "v_mov_b64", "v_mad_u64", "v_add_co", "v_sub_co", "v_min_i32", "v_max_i32", "v_mad_i32", "v_add_i32", "v_sub_i32")
TRANS = ("v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32", "v_rcp_iflag")


def cost(line):
    op = line.split()[0]
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return 0.0, "vmem"
    if op.startswith("ds_"):
        return 0.0, "lds"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return 0.0, "smem"
    if op.startswith("s_"):
        return 0.0, "salu"
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


def regions_of(path):
    """Named line ranges of a source file, from `// @region name` ... markers or the function table below."""
    src = open(path).read().split("\n")
    table = []
    name = os.path.basename(path)
    if name == "tr_kernels.h":
        marks = [("eval_light", r"__device__ __forceinline__ void eval_light\("), ("eval_punctual", r"void eval_punctual\("),
                 ("pyramid+lut", r"^struct tap_pair"), ("digest", r"^__device__ __forceinline__ void lut_rows"),
                 ("cluster_lookup", r"^constexpr uint32_t kNoCluster"), ("pixel:frame", r"^__device__ __forceinline__ f3 shade_pixel\("),
                 ("pixel:taps", r"auto issue_taps = "), ("pixel:sun", r"auto lights_phase = "), ("pixel:punctual", r"auto punctual = "),
                 ("pixel:tail", r"auto tail = "), ("textured_front_end", r"^struct quad_derivs"),
                 ("lite_front_end", r"f3 shade_pixel_lite\("), ("kernel:prologue", r"^constexpr uint32_t planes_nt_mask"),
                 ("kernel:fetch", r"auto fetch = "), ("kernel:vis_fetch", r"if constexpr \(VIS\) \{$"), ("kernel:plane_fetch", r"t.mat = ld_plane<uint32_t"),
                 ("kernel:loop", r"const uint32_t wave_tiles = "), ("kernel:quad_derivs", r"quad_derivs qd;"),
                 ("kernel:material_loop", r"^            while \(todo\) \{"), ("kernel:store", r"const uint32_t out_px = cur.px"),
                 ("kernel:vis_zero", r"the reader of a visibility word leaves it zeroed"),
                 ("kernel:mip1", r"Level 1 of the opaque pyramid straight from"), ("kernel:write", r"if \(write && "),
                 ("other_kernels", r"^__global__ __launch_bounds__\(256\) void depth_slice_kernel")]
        for nm, pat in marks:
            for i, l in enumerate(src):
                if re.search(pat, l):
                    table.append((i + 1, nm))
                    break
            else:   # (a marker that is not found would silently charge its instructions to the region in front of it)
                raise SystemExit(f"tools/isa_regions.py: source anchor of region {nm!r} not found in {name}: {pat!r}")
        table.sort()
    return table


def main():
    pat = sys.argv[1]
    extra = [a for a in sys.argv[2:] if a.startswith("-")]
    os.makedirs(os.path.join(ROOT, "build_ab"), exist_ok=True)
    asm = os.path.join(ROOT, "build_ab", "tr_shade_lines.s")
    flags = [f for f in g.HIPCC_FLAGS if f not in ("-fPIC", "-shared")]
    subprocess.run(["/opt/rocm/bin/hipcc"] + flags + extra + ["-gline-tables-only", "-S", "--cuda-device-only", "-o", asm,
                                                              os.path.join(g.CSRC, "tr_shade.hip")], check=True, stderr=subprocess.DEVNULL)
    s = open(asm).read()
    files = {}
    for m in re.finditer(r'^\s*\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"', s, re.M):
        files[int(m.group(1))] = os.path.basename(m.group(3))
    m = re.search(r"^(\S*%s\S*):" % re.escape(pat), s, re.M)
    body = s[m.start():s.index("s_endpgm", m.start())]
    tables = {f: regions_of(os.path.join(g.CSRC, f)) for f in ("tr_kernels.h",)}
    region = "entry"
    agg = {}
    for l in body.split("\n"):
        t = l.strip()
        lm = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
        if lm:
            f, line = files.get(int(lm.group(1)), "?"), int(lm.group(2))
            if f == "tr_kernels.h":
                r = "?"
                for start, nm in tables[f]:
                    if line >= start:
                        r = nm
                region = r
            elif f in ("tr_texture_kernels.h",):
                region = "texture_sampler"
            elif f in ("tr_visibility.h",):
                region = "vis_interpolate"
            continue
        if not l.startswith("\t") or not t or t.startswith((".", ";")):
            continue
        c, k = cost(t)
        a = agg.setdefault(region, {"cyc": 0.0, "n": {}})
        a["cyc"] += c
        a["n"][k] = a["n"].get(k, 0) + 1
    tot = sum(a["cyc"] for a in agg.values())
    print(m.group(1)[:100])
    for r, a in sorted(agg.items(), key=lambda kv: -kv[1]["cyc"]):
        valu = sum(v for k, v in a["n"].items() if k in ("pair", "sgpr", "half", "int", "trans"))
        print(f"  {r:22s} valu {valu:4d}  issue cycles {a['cyc']:7.1f} ({100 * a['cyc'] / tot:4.1f} %)  {a['n']}")
    print(f"  static total {tot:.0f} VALU issue cycles")


if __name__ == "__main__":
    main()
