#!/bin/bash
# Grid-size sweep of the 4K transmissive kernel at steady-state clocks (run on the GPU box):
#   bash tools/grid_sweep.sh build_ab/a.so "256 512 1024 2048"     (blocks per XCD; 256 = one resident round of 256-thread blocks)
cd ${GRAFT_REPO_ROOT:-.}
for b in ${2:-256 512 1024 2048 4096}; do echo "TR_BLOCKS_PER_XCD=$b"; TR_BLOCKS_PER_XCD=$b timeout 200 python3 tools/ab_kernel.py --lights 1 --rounds 2 $1 | tail -1; done
