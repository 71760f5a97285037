#!/bin/bash
# Issue-side PMC passes for the transmissive kernel, with the VALU microbenchmark as calibration (same counters on
# kernels whose VALU pipe is known to be saturated).  Run on the GPU box via gpurun; kernel-trace + pmc only.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-pmc2}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
A="SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU"
B="SQ_INSTS_BRANCH SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_INSTS_VALU_TRANS_F32"
C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LEVEL_WAVES SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_SALU"
i=0
for P in "$A" "$B" "$C"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/bench$i -o p -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-variants --no-traffic > $OUT/bench$i.log 2>&1 || echo "bench pass $i failed"
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/ubench$i -o p -- $R/build_ab/valu_rate > $OUT/ubench$i.log 2>&1 || echo "ubench pass $i failed"
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for tag in ("bench", "ubench"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(f"{out}/{tag}*/**/*counter_collection.csv", recursive=True)):
        for row in csv.DictReader(open(f)):
            agg[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in agg.items():
        if "shade_kernel<true" in k or "void k<" in k:
            print(tag, k)
            print("   ", {c: round(sum(v) / len(v)) for c, v in sorted(cs.items())})
PY
