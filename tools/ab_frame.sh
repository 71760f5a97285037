#!/bin/bash
# A/B of whole-frame rates between library builds ON ONE BOX (boxes of the pool differ by up to 5 %): ab_frame.sh ROUNDS lib...
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$1; shift
python tools/make_demo_gltf.py /tmp/demo.glb > /dev/null
for k in $(seq $R); do
  for lib in "$@"; do
    a=$(TR_AB_LIB=$lib timeout 200 python tools/gpu_bench_frame.py meshes 2>&1 | tail -1 | sed 's/.*triangles: \([0-9.]*\) us.*/\1/')
    b=$(TR_AB_LIB=$lib timeout 200 python tools/gpu_bench_frame.py /tmp/demo.glb 2>&1 | tail -1 | sed 's/.*triangles: \([0-9.]*\) us.*/\1/')
    echo "$(basename $lib): meshes $a  gltf $b"
  done
done
