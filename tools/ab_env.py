#!/usr/bin/env python3
"""Dev tool (GPU box): same-box A/B of environment settings on the DIAGNOSTICS build (the only build that honours the A/B switches):
alternating bench.py runs, medians.   python tools/ab_env.py [--rounds 3] [--bench-args "--workload c2 --steps 40"] "" "VELO_LM_LEAN=0" "VELO_LM_LEAN=0 VELO_ASSOC_LDS_PAD=0" ...
An empty setting = the build's defaults.  Prints one line per setting: median / all values of `value`, single-pair ms."""
import json, os, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
rounds, bench_args, lib = 3, "--steps 40 --warmup 5", os.path.join(ROOT, "vision-enhanced-lidar-odometry_amd", "csrc", "libvelo_hip_diag.so")
while args and args[0].startswith("--"):
    if args[0] == "--rounds": rounds = int(args[1]); args = args[2:]
    elif args[0] == "--bench-args": bench_args = args[1]; args = args[2:]
    elif args[0] == "--product": lib = None; args = args[1:]
    else: raise SystemExit("unknown option " + args[0])
settings = args or [""]
res = {s: [] for s in settings}
for r in range(rounds):
    for s in settings:
        env = dict(os.environ)
        if lib: env["VELO_LIB_PATH"] = lib
        extra = []
        for kv in s.split():
            if kv.startswith("--") or not "=" in kv: extra.append(kv); continue      # bench.py arguments of this setting
            k, v = kv.split("=", 1); env[k] = v                                    # (VELO_LIB_PATH=... selects another build)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-legs", "--no-cpu-baseline", *bench_args.split(), *extra], env=env, capture_output=True, text=True)
        try:
            line = json.loads(out.stdout.strip().splitlines()[-1])
            res[s].append((line["value"], (line.get("single_pair") or {}).get("ms_per_pair", 0.0), line["roofline"]["avg_launch_us"]))
        except Exception:       # noqa: BLE001
            print("FAILED", repr(s), out.stderr[-800:], flush=True)
for s in settings:
    v = [a for a, _, _ in res[s]]
    if v:
        print(f"{s or '(defaults)':60s} median {statistics.median(v):8.1f} pairs/s  all {[round(a) for a in v]}  single {[round(b, 3) for _, b, _ in res[s]]} ms  assoc {[round(c, 1) for _, _, c in res[s]]} us", flush=True)
