#!/bin/bash
# Steady-state ablation sweep of the 4K transmissive kernel (run on the GPU box): profiling builds made with
#   hipcc ... -DTR_ABLATION=1 [-DTR_TIMING=1] transmission_renderer_amd/csrc/tr_shade.hip -o build_ab/<lib>.so
# TR_ABLATE bits: 1 no pyramid taps, 2 no LUT, 4 no sun, 8 no punctual lights, 32 streaming skeleton only,
# 64 no G-buffer traffic (synthetic inputs), 128 no stores.
cd ${GRAFT_REPO_ROOT:-.}
LIB=${1:-build_ab/ablate.so}
for a in ${2:-0 64 32 3 12 15 79 128}; do echo "TR_ABLATE=$a"; TR_ABLATE=$a timeout 200 python3 tools/ab_kernel.py --lights 1 --rounds 1 $LIB; done
(while true; do rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -2; sleep 0.5; done) > /tmp/clk.log 2>&1 &
CLK=$!
timeout 100 python3 tools/ab_kernel.py --lights 1 --rounds 1 build_ab/longlists.so
kill $CLK
sort /tmp/clk.log | uniq -c | sort -rn | head -8
