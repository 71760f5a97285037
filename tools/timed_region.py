"""Cuts the rocprofv3 kernel trace of a `bench.py` run to the K launches the bench times.
    python tools/timed_region.py gpurun_out/<round>/bench_default/t_kernel_trace.csv <bench JSON line file> > out.json
bench.py lists the sequence of its tr::shade_kernel launches in `launch_log` ([[phase, count], ...]): clock ramp,
W warm-up launches, K timed launches back to back, K more with per-launch events, ..."""
import csv, json, statistics as st, sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "shade_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
t0s = [int(r["Start_Timestamp"]) / 1e3 for r in rows]
t1s = [int(r["End_Timestamp"]) / 1e3 for r in rows]
b = json.load(open(sys.argv[2]))
r = lambda x: round(x, 2)
out = {"source": "rocprofv3 --kernel-trace --stats -- python3 bench.py (tools/prof_round.sh), kernel tr::shade_kernel<true, uint2, 0, false>; durations in us",
       "launches_total": len(d), "launches_logged": sum(c for _, c in b["launch_log"]),
       "all_launches": {"avg": r(st.mean(d)), "min": r(min(d)), "max": r(max(d))}, "phases": {}}
pos = 0
for phase, count in b["launch_log"]:
    seg = d[pos:pos + count]
    if seg:
        # span_per_launch: (last end - first start) / launches of the phase — with the frame's bands (or S frames) on several streams the launches of
        # different streams overlap, a launch's own start-to-end duration ("avg") is then ~S x the time per frame
        span = (max(t1s[pos:pos + count]) - min(t0s[pos:pos + count])) / len(seg)
        out["phases"][phase] = {"launches": len(seg), "avg": r(st.mean(seg)), "p50": r(st.median(seg)), "min": r(min(seg)), "max": r(max(seg)),
                                "span_per_launch": r(span)}
    pos += count
out["timed_region"] = dict(out["phases"].get("timed", {}), note="the K back-to-back launches bench.py times (dispatch latency of a dependent launch is inside these durations)")
out["bench_line"] = {"avg_kernel_ms": b["roofline"]["avg_kernel_ms"], "frac": b["roofline"]["frac"], "value": b["value"],
                     "streams": b["roofline"].get("streams", 1), "kernel_ms_on_its_stream": b["roofline"].get("kernel_ms_on_its_stream"),
                     "single_stream_avg_kernel_ms": b.get("single_stream", {}).get("avg_kernel_ms")}
print(json.dumps(out, indent=1))
