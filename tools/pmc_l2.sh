#!/bin/bash
# Runs ON THE GPU BOX: L2 hit/miss of the association kernel for the current env (one PMC pass).
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_l2_$1
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/prof_assoc.py c2 > /dev/null 2> $OUT.err
python3 - $OUT <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "assoc_search" in r["Kernel_Name"]: per[(r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
    for (name, _), v in per.items(): agg[name].append(v)
print(sys.argv[1].split("_")[-1], {k: round(sum(v)/len(v)) for k, v in sorted(agg.items())})
PY
