cd /tmp && export TMPDIR=/tmp
for L in H N; do
  export VELO_LIB_PATH=$GRAFT_REPO_ROOT/build_ab/lib$L.so
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02_prof_$L -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-legs > $GRAFT_REPO_ROOT/gpurun_out/r02_prof_$L.json 2> $GRAFT_REPO_ROOT/gpurun_out/r02_prof_$L.err
done
