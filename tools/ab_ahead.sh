#!/bin/bash
# A/B on the GPU box, main leg only, alternating: drive steps as ONE call with the next frame announced (default), as two calls with the
# announcement (VELO_BENCH_TWO_CALLS=1: velo_hint_next_frame + velo_register_batch + velo_pose_handoff), and steps that load their own
# frame (VELO_BENCH_NO_AHEAD=1, rounds 4 and before).  usage: tools/ab_ahead.sh [runs] [extra bench flags]
runs=${1:-3}; shift
mkdir -p gpurun_out
for r in $(seq 1 $runs); do
  for mode in one_call two_calls plain; do
    unset VELO_BENCH_NO_AHEAD VELO_BENCH_TWO_CALLS
    [ $mode = plain ] && export VELO_BENCH_NO_AHEAD=1
    [ $mode = two_calls ] && export VELO_BENCH_TWO_CALLS=1
    v=$(timeout 300 python bench.py --no-legs --no-cpu-baseline --steps 20 --warmup 5 "$@" 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.readlines()[-1])['value'])")
    echo "$mode run $r: $v"
  done
done | tee -a gpurun_out/ab_ahead.txt
