#!/usr/bin/env python3
"""Dev tool (GPU box): same-box A/B of BUILDS of the library (product settings, no environment switches): alternating bench.py runs with
VELO_LIB_PATH pointing at each build, medians.   python tools/ab_libs.py [--rounds 3] [--bench-args "..."] name=path ...   ("product" = the in-tree library)"""
import json, os, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
rounds, bench_args = 3, "--steps 20 --warmup 5"
while args and args[0].startswith("--"):
    if args[0] == "--rounds": rounds = int(args[1]); args = args[2:]
    elif args[0] == "--bench-args": bench_args = args[1]; args = args[2:]
    else: raise SystemExit("unknown option " + args[0])
libs = [a.split("=", 1) if "=" in a else (a, None) for a in args]
res = {n: [] for n, _ in libs}
for r in range(rounds):
    for n, pth in libs:
        env = dict(os.environ)
        if pth: env["VELO_LIB_PATH"] = pth if os.path.isabs(pth) else os.path.join(ROOT, pth)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-legs", "--no-cpu-baseline", *bench_args.split()], env=env, capture_output=True, text=True)
        try:
            line = json.loads(out.stdout.strip().splitlines()[-1])
            ks = {k["kernel"]: k["avg_launch_us"] for k in line["kernels"]}
            sp = line.get("single_pair") or {}
            res[n].append((line["value"], sp.get("ms_per_pair", 0.0), ks, sp.get("assoc_avg_launch_us", 0.0)))
        except Exception:       # noqa: BLE001
            print("FAILED", n, out.stderr[-800:], flush=True)
for n, _ in libs:
    v = [a[0] for a in res[n]]
    if not v: continue
    names = sorted({k for a in res[n] for k in a[2]})
    print(f"{n:12s} median {statistics.median(v):8.1f} pairs/s  all {[round(a) for a in v]}  single {[round(a[1], 3) for a in res[n]]} ms  single-assoc {[round(a[3], 1) for a in res[n]]} us", flush=True)
    for k in names:
        print(f"      {k:40s} {[round(a[2].get(k, 0), 1) for a in res[n]]} us")
