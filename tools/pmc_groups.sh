#!/bin/bash
# rocprofv3 --pmc passes (one counter group per pass, kernel trace only) over tools/pmc_child.py; prints per counter the
# median over the shading kernel's launches.  usage: pmc_groups.sh OUTDIR "GROUP1" "GROUP2" ...   (env: TR_AB_LIB, TR_ABLATE, TR_AB_CFG)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$(realpath -m $1); shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "$@"; do
  i=$((i+1))
  timeout ${TR_PMC_TIMEOUT:-150} rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/p$i -o p -- python3 $R/tools/pmc_child.py > $OUT/p$i.log 2>&1 || echo "pass $i ($C) failed: $(tail -2 $OUT/p$i.log)"
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections
vals = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "shade_" in row["Kernel_Name"]:
            vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(vals):
    v = sorted(vals[k])
    print(f"{k:44s} {v[len(v)//2]:16.0f}   (n={len(v)})")
PY
