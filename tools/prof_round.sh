#!/bin/bash
# Round-end profiles (run on the GPU box via gpurun): rocprofv3 --kernel-trace --stats of the DEFAULT bench command and
# of the whole glTF-in/frame-out pipeline at 4K.  Summaries are copied to profiles/ by hand afterwards.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-round}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_default -o t -- python3 $R/bench.py > $OUT/bench_default.log 2>&1
python3 $R/tools/make_demo_gltf.py $OUT/demo.glb > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pipeline_gltf -o t -- python3 -m transmission_renderer_amd.cli $OUT/demo.glb --width 3840 --height 2160 --out $OUT/demo_4k.png > $OUT/pipeline_gltf.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pipeline_meshes -o t -- python3 -m transmission_renderer_amd.cli meshes --width 3840 --height 2160 --out $OUT/meshes_4k.png > $OUT/pipeline_meshes.log 2>&1
rm -f $OUT/demo_4k.png $OUT/meshes_4k.png
for d in bench_default pipeline_gltf pipeline_meshes; do
  f=$(find $OUT/$d -name '*kernel_stats.csv' | head -1)
  echo "== $d"; [ -n "$f" ] && head -14 "$f"
done
tail -1 $OUT/bench_default.log
