#!/bin/bash
# Runs ON THE GPU BOX: what the tube kernel waits for -- SQ issue / wait split by instruction class, LDS conflicts, texture-addresser and L1 stalls,
# (a TA_* + GRBM group aborted rocprofv3 and hung the box until the time limit: the TA / TCP / TCC groups are gone) -- one rocprofv3 --pmc pass per counter group on bench.py's single-pair shape (ARGS overrides the bench arguments).
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_deep
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export VELO_DRIVE_CACHE=/tmp/velo_drive_cache
ARGS=${ARGS:---steps 3 --warmup 1 --batch 1 --no-cpu-baseline --no-legs}
python3 $GRAFT_REPO_ROOT/bench.py $ARGS > /dev/null 2>&1
i=0
while read -r G; do
  [ -z "$G" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $G --output-format csv -d $OUT/g$i -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > /dev/null 2> $OUT/g$i.err
done <<'GROUPS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL
GROUPS
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_deep"
for d in sorted(glob.glob(out + "/g*/")):
    f = glob.glob(d + "/*/*counter_collection.csv")
    if not f: print(os.path.basename(d.rstrip("/")), "no counters:", open(d.rstrip("/") + ".err").read()[-300:]); continue
    per = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        key = "assoc" if "assoc_search_v5" in k else ("lm" if ("lm_iter" in k or "eval_step" in k) else None)
        if key: per[key][int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    for key in per:
        ids = sorted(per[key])
        names = sorted({n for i in ids for n in per[key][i]})
        print(os.path.basename(d.rstrip("/")), key, "n", len(ids), " ".join(f"{n}={sum(per[key][i][n] for i in ids) / len(ids):.4g}" for n in names), flush=True)
PY
