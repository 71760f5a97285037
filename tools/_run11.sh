R=$PWD
OUT=$R/gpurun_out/r02m
mkdir -p $OUT
python3 tools/make_demo_gltf.py $OUT/demo.glb > /dev/null
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/frame_loop -o t -- python3 $R/tools/gpu_bench_frame.py $OUT/demo.glb > $OUT/frame_loop.log 2>&1
grep "us per frame" $OUT/frame_loop.log
f=$(find $OUT/frame_loop -name '*kernel_stats.csv' | head -1); head -8 "$f" | cut -c1-130
