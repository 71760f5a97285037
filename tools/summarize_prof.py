"""Summarise rocprofv3 CSV output (kernel trace + PMC passes) into one text/JSON report."""
import csv, glob, json, os, sys, collections
out = sys.argv[1]
rep = {}
def find(pat):
    return sorted(glob.glob(os.path.join(out, pat), recursive=True))
# kernel trace: per-kernel durations
for f in find("trace/**/*kernel_trace.csv"):
    d = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        d[row["Kernel_Name"]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    rep["kernel_trace_us"] = {k[:100]: {"calls": len(v), "avg": sum(v) / len(v), "min": min(v), "max": max(v),
                                        "p50": sorted(v)[len(v) // 2]} for k, v in d.items()}
for f in find("trace/**/*kernel_stats.csv"):
    rep["kernel_stats_csv"] = open(f).read().splitlines()[:12]
# PMC passes
pmc = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for f in find("pmc*/**/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"][:100]
        pmc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        meta[k] = {x: row.get(x) for x in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Grid_Size", "Workgroup_Size")}
rep["pmc_avg_per_dispatch"] = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in pmc.items()}
rep["dispatch_meta"] = meta
json.dump(rep, open(os.path.join(out, "summary.json"), "w"), indent=1)
for k, v in rep.get("kernel_trace_us", {}).items():
    print(f"{v['avg']:10.2f} us avg  {v['p50']:10.2f} p50  x{v['calls']:4d}  {k}")
for k, cs in rep["pmc_avg_per_dispatch"].items():
    if "shade_kernel" in k or "downsample" in k:
        print(k, meta.get(k))
        for c, v in sorted(cs.items()):
            print(f"    {c:40s} {v:18.1f}")
