#!/usr/bin/env python3
"""Dev tool: same-box A/B of two builds of the library (boxes differ by +-10 %, single runs by +-5 %): alternates
`bench.py` subprocesses with VELO_LIB_PATH=A / B, N rounds, prints medians.
Usage: python tools/ab_bench.py libA.so libB.so [rounds] [extra bench args...]"""
import json, os, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
a, b = sys.argv[1], sys.argv[2]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 5
extra = sys.argv[4:]
res = {(lib, mode): [] for lib in (a, b) for mode in ("1", "8")}
for r in range(rounds):
    for lib in (a, b) if r % 2 == 0 else (b, a):
        for mode in ("1", "8"):
            env = dict(os.environ, VELO_LIB_PATH=lib)
            out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", mode, "--no-cpu-baseline", "--no-legs", *extra],
                                 env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
            res[(lib, mode)].append(json.loads(out)["value"])
for mode in ("1", "8"):
    ma, mb = statistics.median(res[(a, mode)]), statistics.median(res[(b, mode)])
    print(f"pairs in flight {mode}: A {ma:8.1f}  B {mb:8.1f}  B/A {mb / ma:.3f}   A runs {[round(v) for v in res[(a, mode)]]}  B runs {[round(v) for v in res[(b, mode)]]}")
