#!/bin/bash
# Runs ON THE GPU BOX: per-section budget of one live LM launch (tools/lm_trace.py: stage stamps of the diagnostics build, one pair) ALONE and
# UNDER LOAD -- a second process keeps 8 fixed pairs in flight (bench.py --same-pairs) while the traced pair runs.  -> gpurun_out/r05_lm_sections.txt
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_lm_sections.txt
export VELO_DRIVE_CACHE=/tmp/velo_drive_cache
{
echo "== one pair alone (tools/lm_trace.py; lm_iter_kernel: every workgroup consumes the previous sweep's rows, runs the transition, sweeps) =="
python3 $R/tools/lm_trace.py 2>&1 | tail -25
echo
echo "== the same pair while another process keeps 8 pairs in flight on the chip =="
python3 $R/bench.py --no-legs --no-cpu-baseline --same-pairs --steps 6000 --warmup 5 > /dev/null 2>&1 &
BG=$!
sleep 12
python3 $R/tools/lm_trace.py 2>&1 | tail -25
kill $BG 2>/dev/null; wait $BG 2>/dev/null
} > $OUT
cat $OUT
