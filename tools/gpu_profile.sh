#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats of the default bench command plus the PMC passes the
# roofline object needs (separate passes, no trace domains mixed with --pmc).  Output under gpurun_out/$1/.
set -u
TAG=${1:-r01}
WL=${WORKLOAD:-c2}          # WORKLOAD=c4 tools/gpu_profile.sh r02_c4: the same passes on the 2M-point map
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
STEPS=${STEPS:-20}          # the driver's command: python bench.py --steps 20 --warmup 5
# the drives' frames are synthesised ONCE, before any profiler is loaded (worker processes); every pass below reads them back
export VELO_DRIVE_CACHE=/tmp/velo_drive_cache
python3 $GRAFT_REPO_ROOT/bench.py --steps $STEPS --warmup 5 --no-cpu-baseline --no-legs --workload $WL > $OUT/bench_plain.json 2> $OUT/bench_plain.err
python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --batch 1 --no-cpu-baseline --no-legs --workload $WL > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --batch 2 --no-cpu-baseline --no-legs --workload $WL > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps $STEPS --warmup 5 --no-cpu-baseline --no-legs --workload $WL > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --batch 1 --no-cpu-baseline --no-legs --workload $WL > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --batch 1 --no-cpu-baseline --no-legs --workload $WL > /dev/null 2>&1
# the kernels of a lock-step group alone on the chip (one group of two contexts): HBM-side bytes per launch of the batched kernels
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_b2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --batch 2 --no-cpu-baseline --no-legs --workload $WL > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_b2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --batch 2 --no-cpu-baseline --no-legs --workload $WL > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --batch 1 --no-cpu-baseline --no-legs --workload $WL > /dev/null 2>&1
# round 4: the same counters on THE BENCH'S OWN COMMAND SHAPE -- 8 pairs in flight, the instantiations the measured mode launches
# (eval_step_batch_lean_v_kernel, assoc_search_v5_batch_kernel ...).  rocprofv3 serialises the dispatches while it counts, so these are
# per-launch figures of the kernels that ran, not of their overlap; GRBM_GUI_ACTIVE / SQ_BUSY_CYCLES say how much of a launch the
# shader engines were busy.
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_b8 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-legs --workload $WL > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_b8 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-legs --workload $WL > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq_b8 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-legs --workload $WL > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES GRBM_COUNT --output-format csv -d $OUT/pmc_busy_b8 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-legs --workload $WL > $OUT/pmc_busy_b8.out 2> $OUT/pmc_busy_b8.err
ls $OUT
