cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu -x > gpurun_out/gpu_tests.log 2>&1; tail -3 gpurun_out/gpu_tests.log
for f in 1 2 3; do
python3 bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-legs 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2 batch8', round(d['value'],1), round(d['ms_per_step'],3))"
done
python3 bench.py --batch 1 --steps 20 --warmup 4 --no-cpu-baseline --no-legs 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2 single', round(d['value'],1), round(d['ms_per_step'],3))"
