#!/bin/bash
# Shader clock and power while a variant of the 4K transmissive kernel runs continuously (run on the GPU box):
#   bash tools/clock_watch.sh build_ab/a.so build_ab/b.so ...
cd ${GRAFT_REPO_ROOT:-.}
for lib in "$@"; do
  TR_AB_STEPS=40000 timeout 60 python3 tools/ab_kernel.py --lights 1 --rounds 1 $lib > /tmp/ab_$$.log 2>&1 &
  AB=$!
  sleep 4
  for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Power (W)\|Socket Power" | tr '\n' ' '; echo; sleep 0.3; done
  wait $AB
  tail -1 /tmp/ab_$$.log
done
