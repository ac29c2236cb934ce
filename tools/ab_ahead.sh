#!/bin/bash
# A/B on the GPU box: drive steps with the next frame announced (velo_hint_next_frame, default) against steps that load their own frame
# (VELO_BENCH_NO_AHEAD=1), alternating, main leg only.  usage: tools/ab_ahead.sh [runs] [extra bench flags]
runs=${1:-3}; shift
mkdir -p gpurun_out
for r in $(seq 1 $runs); do
  for mode in ahead plain; do
    if [ $mode = plain ]; then export VELO_BENCH_NO_AHEAD=1; else unset VELO_BENCH_NO_AHEAD; fi
    v=$(timeout 300 python bench.py --no-legs --no-cpu-baseline --steps 20 --warmup 5 "$@" 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.readlines()[-1])['value'])")
    echo "$mode run $r: $v"
  done
done | tee gpurun_out/ab_ahead.txt
