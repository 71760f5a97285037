#!/bin/bash
# PMC counter passes over the whole-frame loop (tools/gpu_bench_frame.py), per kernel.  Run on the GPU box via gpurun:
#   bash tools/prof_frame_pmc.sh [outdir-name] [scene.glb | meshes]
# Counters in their own runs, kernel-trace only.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-frame_pmc}
SCENE=${2:-$R/gpurun_out/demo.glb}
mkdir -p $OUT
export PYTHONPATH=$R
[ -f $R/gpurun_out/demo.glb ] || python3 $R/tools/make_demo_gltf.py $R/gpurun_out/demo.glb > /dev/null
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
         "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT TCC_HIT_sum TCC_MISS_sum" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_GDS SQ_INSTS_FLAT"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc$i -o p -- python3 $R/tools/gpu_bench_frame.py $SCENE > $OUT/pmc$i.log 2>&1 || echo "pass $i failed"
done
python3 - $OUT <<'PY'
import csv, glob, os, sys, collections, json
out = sys.argv[1]
pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(out, "pmc*/**/*counter_collection.csv"), recursive=True)):
    for row in csv.DictReader(open(f)):
        pmc[row["Kernel_Name"][:110]][row["Counter_Name"]].append(float(row["Counter_Value"]))
rep = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in pmc.items() if k.startswith(("tr::", "void tr::"))}
json.dump(rep, open(os.path.join(out, "summary.json"), "w"), indent=1)
for k, cs in rep.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"    {c:28s} {v:16.1f}")
PY
