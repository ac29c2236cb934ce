cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu -x > gpurun_out/gpu_tests.log 2>&1; grep -n "passed\|failed" gpurun_out/gpu_tests.log | tail -2
for f in 1 0 1 0; do
VELO_LM_FUSED=$f python3 bench.py --workload c3 --steps 16 --warmup 4 --no-cpu-baseline --no-legs 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3 batch8 fused $f', round(d['value'],1), round(d['ms_per_step'],3))"
done
