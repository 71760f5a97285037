"""Cuts the rocprofv3 kernel trace of a `bench.py` run to the K launches the bench times.
    python tools/timed_region.py gpurun_out/<round>/bench_default/t_kernel_trace.csv <bench JSON line file> > out.json
bench.py issues: clock ramp + W warm-up launches, K timed launches back to back, K more with per-launch events."""
import csv, json, statistics as st, sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "shade_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
b = json.load(open(sys.argv[2]))
n, K = len(d), b["steps"]
timed, second = d[n - 2 * K:n - K], d[n - K:]
r = lambda x: round(x, 2)
print(json.dumps({
    "source": "rocprofv3 --kernel-trace --stats -- python3 bench.py (tools/prof_round.sh), kernel tr::shade_kernel<true, uint2, false>; durations in us",
    "launches_total": n, "all_launches": {"avg": r(st.mean(d)), "min": r(min(d)), "max": r(max(d))},
    "clock_ramp_and_warmup_launches": n - 2 * K,
    "timed_region": {"launches": K, "avg": r(st.mean(timed)), "p50": r(st.median(timed)), "min": r(min(timed)), "max": r(max(timed)),
                     "note": "the K back-to-back launches bench.py times (dispatch latency of a dependent launch is inside these durations)"},
    "percentile_pass": {"launches": K, "avg": r(st.mean(second)), "p50": r(st.median(second)),
                        "note": "per-launch events between the launches: ~10 us gaps"},
    "bench_line": {"avg_kernel_ms": b["roofline"]["avg_kernel_ms"], "frac": b["roofline"]["frac"], "value": b["value"]}}, indent=1))
