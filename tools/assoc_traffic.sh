#!/bin/bash
# Runs ON THE GPU BOX: HBM-side bytes of the tube kernel BY SECTION (where do the 6 x algorithmic bytes come from?).  The diagnostic
# instantiation switches sections off (VELO_DEBUG_SKIP bits: 1 sweep, 2 stage+sweep, 4 all row work, 128 finish gathers, 1024 phase 2;
# 64 = nothing off); FETCH_SIZE and WRITE_SIZE in separate passes, per launch, one pair alone on the chip; the difference between two
# settings is what the section moves.  Traffic = 2 x FETCH_SIZE + WRITE_SIZE (KiB), as everywhere in profiles/*_traffic.json.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/assoc_traffic
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export VELO_LIB_PATH=$GRAFT_REPO_ROOT/vision-enhanced-lidar-odometry_amd/csrc/libvelo_hip_diag.so VELO_ASSOC_VARIANT=5
for S in 64 65 66 68 192 1088; do
  export VELO_DEBUG_SKIP=$S
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --output-format csv -d $OUT/skip_${S}_$C -- python3 $GRAFT_REPO_ROOT/tools/assoc_diag.py child > /dev/null 2> $OUT/skip_${S}_$C.err
  done
done
python3 - <<'PY' | tee $OUT/summary.txt
import csv, glob, os, collections
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/assoc_traffic"
names = {64: "everything on", 65: "sweep off", 66: "staging + sweep off", 68: "all row work off (intervals, run list, staging, sweep)", 192: "finish gathers off", 1088: "phase 2 (askers) off"}
res = {}
for S in (64, 65, 66, 68, 192, 1088):
    v = {}
    for C in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob(f"{out}/skip_{S}_{C}/*/*counter_collection.csv")
        if not f: continue
        per = collections.defaultdict(float)
        for r in csv.DictReader(open(f[0])):
            if "assoc_search_v5" in r["Kernel_Name"] and r["Counter_Name"] == C: per[int(r["Dispatch_Id"])] += float(r["Counter_Value"])
        ids = sorted(per)[6:]                      # (the first call's rounds are cold)
        v[C] = [per[i] for i in ids]
    if len(v) == 2 and v["FETCH_SIZE"] and v["WRITE_SIZE"]:
        n = min(len(v["FETCH_SIZE"]), len(v["WRITE_SIZE"]))
        f = sum(v["FETCH_SIZE"][:n]) / n; w = sum(v["WRITE_SIZE"][:n]) / n
        res[S] = (f, w, (2 * f + w) * 1024 / 1e6)
        print(f"skip {S:5d} ({names[S]}): FETCH_SIZE {f:9.0f} KiB  WRITE_SIZE {w:9.0f} KiB  traffic {res[S][2]:7.2f} MB per launch (algorithmic 6.24 MB)")
if 64 in res:
    b = res[64][2]
    for S, what in ((65, "the sweep (candidate tiles read back from LDS: should be ~0)"), (66, "staging + sweep (candidate gathers from the cell-sorted copy)"), (68, "all row work"), (192, "the finish (winner + ring-neighbour gathers, correspondence stores)"), (1088, "phase 2")):
        if S in res: print(f"  {what}: {b - res[S][2]:6.2f} MB")
PY
