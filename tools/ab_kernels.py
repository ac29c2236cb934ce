#!/usr/bin/env python3
"""Dev tool (GPU box): bench.py on the diagnostics build under a few environment settings; per setting: pairs/s, single pair ms, LM evaluations per pair and
the top kernels of the timed region (share, launches, average launch).   python tools/ab_kernels.py [--bench-args "..."] "SETTING" ..."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
bench_args = "--steps 20 --warmup 5"
if args and args[0] == "--bench-args":
    bench_args = args[1]; args = args[2:]
lib = os.path.join(ROOT, "vision-enhanced-lidar-odometry_amd", "csrc", "libvelo_hip_diag.so")
for s in args or [""]:
    env = dict(os.environ, VELO_LIB_PATH=lib)
    extra = []
    for kv in s.split():
        if kv.startswith("--") or "=" not in kv: extra.append(kv); continue
        k, v = kv.split("=", 1); env[k] = v
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-legs", "--no-cpu-baseline", *bench_args.split(), *extra], env=env, capture_output=True, text=True)
    try:
        line = json.loads(out.stdout.strip().splitlines()[-1])
    except Exception:
        print("FAILED", repr(s), out.stderr[-1500:], flush=True); continue
    sp = line.get("single_pair") or {}
    print(f"{s or '(defaults)'}: {line['value']:.0f} pairs/s, {line['ms_per_step']:.3f} ms/step, single {sp.get('ms_per_pair', 0):.3f} ms, evals/pair {line['config']['lm_evaluations_per_pair']:.1f}, chain {line['chain']['calls']}/{line['chain']['misses']}")
    for k in line["kernels"]:
        print(f"    {k['kernel']:40s} share {k['share']:.2f} launches {k['launches']:5d} avg {k['avg_launch_us']:7.1f} us")
    for k in sp.get("kernels", []):
        print(f"    single: {k['kernel']:32s} share {k['share']:.2f} avg {k['avg_launch_us']:7.1f} us")
