#!/bin/bash
# Runs ON THE GPU BOX: dynamic instruction counts of the tube kernel per section.  The diagnostic instantiation can switch sections
# off (VELO_DEBUG_SKIP bits: 1 sweep, 2 stage+sweep, 4 all row work, 128 finish gathers, 1024 phase 2); the difference of the SQ
# counters between two settings is what the section costs.  Summaries: gpurun_out/assoc_pmc/skip_<bits>/...
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/assoc_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export VELO_LIB_PATH=$GRAFT_REPO_ROOT/vision-enhanced-lidar-odometry_amd/csrc/libvelo_hip_diag.so VELO_ASSOC_VARIANT=5
for S in 64 65 66 68 192 1088; do      # 64 = a bit without meaning: selects the diagnostic instantiation with nothing switched off
  export VELO_DEBUG_SKIP=$S
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/skip_$S -- python3 $GRAFT_REPO_ROOT/tools/assoc_diag.py child > /dev/null 2> $OUT/skip_$S.err
done
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/assoc_pmc"
for d in sorted(glob.glob(out + "/skip_*/")):
    f = glob.glob(d + "/*/*counter_collection.csv")
    if not f: print(d, "no counters"); continue
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f[0])):
        if "assoc_search_v5" not in r["Kernel_Name"]: continue
        per[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    ids = sorted(per)[6:]
    names = sorted({k for i in ids for k in per[i]})
    print(os.path.basename(d.rstrip("/")), " ".join(f"{n}={sum(per[i][n] for i in ids) / max(len(ids), 1):.3g}" for n in names), flush=True)
    for i in ids: print("    round", " ".join(f"{n}={per[i][n]:.3g}" for n in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "GRBM_GUI_ACTIVE")))
PY
