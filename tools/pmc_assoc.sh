#!/bin/bash
# Runs ON THE GPU BOX: PMC passes over tools/prof_assoc.py (association kernel only), one counter group per pass.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_assoc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS" "TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/tools/prof_assoc.py c2 > /dev/null 2> $OUT/p$i.err
done
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_assoc"
agg = collections.defaultdict(list)
for f in glob.glob(out + "/p*/*/*counter_collection.csv"):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "assoc_search" in r["Kernel_Name"]:
            per[(r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
    for (name, _), v in per.items(): agg[name].append(v)
for name in sorted(agg): print(f"{name:36s} mean/launch {sum(agg[name])/len(agg[name]):14.0f}  n {len(agg[name])}")
PY
