#!/bin/bash
# Runs ON THE GPU BOX: VALU / SALU / LDS instruction counts of the tube kernel's diagnostic instantiation with sections skipped
# (VELO_DEBUG_SKIP: 64 = nothing skipped, +1 sweep, +2 staging+sweep, +4 rows and everything below) -> instruction mix by section.
set -u
cd /tmp && export TMPDIR=/tmp
for skip in 64 65 66 68; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_valu_$skip
  rm -rf $OUT; mkdir -p $OUT
  export VELO_DEBUG_SKIP=$skip VELO_ASSOC_VARIANT=5 VELO_LIB_PATH=$GRAFT_REPO_ROOT/vision-enhanced-lidar-odometry_amd/csrc/libvelo_hip_diag.so
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/prof_assoc.py c2 > /dev/null 2> $OUT.err
  python3 - $OUT $skip <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "assoc_search" in r["Kernel_Name"]: per[(r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
    for (name, _), v in per.items(): agg[name].append(v)
print("skip", sys.argv[2], {k: round(sum(v)/len(v)) for k, v in sorted(agg.items())})
PY
done
