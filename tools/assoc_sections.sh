#!/bin/bash
# Runs ON THE GPU BOX: duration of the tube kernel with sections switched off (diagnostic instantiation, VELO_DEBUG_SKIP bits:
# 1 sweep, 2 stage+sweep, 4 all row work, 128 finish gathers), rounds 1-6 of the C2 pose sequence, from a rocprofv3 kernel trace.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/assoc_sections
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export VELO_LIB_PATH=$GRAFT_REPO_ROOT/vision-enhanced-lidar-odometry_amd/csrc/libvelo_hip_diag.so VELO_ASSOC_VARIANT=5
for S in ${SKIPS:-64 65 66 68 196}; do
  export VELO_DEBUG_SKIP=$S
  rocprofv3 --kernel-trace --output-format csv -d $OUT/skip_$S -- python3 $GRAFT_REPO_ROOT/tools/assoc_diag.py child > /dev/null 2> $OUT/skip_$S.err
done
python3 - <<'PY'
import csv, glob, os
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/assoc_sections"
for d in sorted(glob.glob(out + "/skip_*/")):
    f = glob.glob(d + "/*/*kernel_trace.csv")
    if not f: print(d, "no trace"); continue
    rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(f[0])) if "assoc_search_v5" in r["Kernel_Name"])
    print(os.path.basename(d.rstrip("/")), " ".join(f"{(e - s) / 1000:.1f}" for s, e in rows[6:]), "us", flush=True)
PY
