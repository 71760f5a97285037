#!/bin/bash
# Round-6 profile set (run on the GPU box via gpurun): the default bench line on a fresh box, the same command under
# rocprofv3 --kernel-trace --stats with its timed region cut out, and the whole-frame loops per kernel.  Every rocprofv3
# call is bounded by `timeout`; counters are collected by bench.py itself (separate --pmc child passes).
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/${1:-r06}
mkdir -p $O
cd $R
python bench.py > $O/bench_first_on_fresh_box_line.json 2> $O/bench_first.err
python -m pytest tests -m gpu -q -s > $O/gputest_s.log 2>&1; tail -1 $O/gputest_s.log
grep -h "^\[parity\]\|^\[golden" $O/gputest_s.log > $O/parity_lines_of_the_gpu_tests.txt
python tools/gpu_parity_report.py > $O/parity_report.txt 2> $O/parity_report.err; tail -2 $O/parity_report.txt | cut -c1-160
python tools/make_demo_gltf.py $O/demo.glb > /dev/null
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_default -o t -- python3 $R/bench.py --no-cpu-baseline > $O/bench_default_line.json 2> $O/bench_default.err
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fl_meshes -o t -- python3 $R/tools/gpu_bench_frame.py meshes > $O/fl_meshes.log 2>&1
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fl_gltf -o t -- python3 $R/tools/gpu_bench_frame.py $O/demo.glb > $O/fl_gltf.log 2>&1
cd $R
python3 tools/timed_region.py $(find $O/bench_default -name '*kernel_trace.csv' | head -1) $O/bench_default_line.json > $O/bench_default_timed_region.json 2>> $O/bench_default.err
for d in bench_default fl_meshes fl_gltf; do
  f=$(find $O/$d -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/${d}_kernel_stats.csv
done
rm -rf $O/bench_default $O/fl_meshes $O/fl_gltf $O/demo.glb
tail -1 $O/fl_meshes.log | cut -c1-90; tail -1 $O/fl_gltf.log | cut -c40-130
python3 - $O <<'PY'
import json, sys
o = sys.argv[1]
for name in ("bench_first_on_fresh_box_line.json", "bench_default_line.json"):
    d = json.loads(open(f"{o}/{name}").read().strip().splitlines()[-1])
    print(name, "value", d["value"], "ms_per_step", d["ms_per_step"], "frac", d["roofline"]["frac"], "same_input", d["same_input"]["us"],
          "single", d["single_stream"]["us"], "frame", (d.get("frame_pipeline") or {}).get("us_per_frame"))
print(open(f"{o}/bench_default_timed_region.json").read()[:600])
PY
