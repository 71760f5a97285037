R=$PWD
python -m pytest tests -q -m gpu > gpurun_out/r02q_tests.txt 2>&1; grep -n "passed\|failed" gpurun_out/r02q_tests.txt | tail -2
bash tools/prof_round.sh r02b > gpurun_out/r02b_round.log 2>&1
OUT=$R/gpurun_out/r02b
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/frame_loop -o t -- python3 $R/tools/gpu_bench_frame.py $OUT/demo.glb > $OUT/frame_loop.log 2>&1
grep "us per frame" $OUT/frame_loop.log
cd $R
bash tools/prof_pmc.sh r02b_pmc > gpurun_out/r02b_pmc.log 2>&1; grep -A3 "FETCH_SIZE\|WRITE_SIZE" gpurun_out/r02b_pmc.log | head -8
grep "^{" gpurun_out/r02b/bench_default.log | cut -c1-300
