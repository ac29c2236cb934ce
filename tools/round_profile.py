#!/usr/bin/env python3
"""Dev tool (round 6): the six association rounds of a DRIVE pair replayed host-driven at the poses the LM solves really produce
(associate -> solve -> associate ...), each association timed alone (call incl. sync) and -- on the diagnostics build with
VELO_DEBUG_SKIP=16 VELO_DEBUG_EACH=1 -- with its counters (clusters, row chunks, staged candidates, of them in phase 2, rows).
    python tools/round_profile.py [pair index k >= 1 of drive 0] [reps]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("VELO_DEBUG_SKIP"):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _diag  # noqa: E702,F401
import numpy as np
import velo_amd  # noqa: F401
from velo_amd import api, synth

c4 = len(sys.argv) > 1 and sys.argv[1] == "c4"
k = 3 if c4 else (int(sys.argv[1]) if len(sys.argv) > 1 else 3)
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
if c4:                                                                # the scan-to-map pair of BASELINE configs[3]
    m = synth.scan_to_map(2_000_000)
    tgt, src = (m["tgt_xyz"], m["tgt_off"]), (m["src_xyz"], m["src_off"])
    x_guess = np.asarray(m["x0"], dtype=np.float64)
    plan = {"x_true": {k: np.asarray(m["x_true"])}}
else:
    plan = synth.drive(k + 2, seed=0)
    tgt, src = plan["frames"][k], plan["frames"][k + 1]
    x_guess = np.asarray(plan["x_true"][k - 1], dtype=np.float64)    # constant velocity: the previous pair's motion
c = api.Context(0, icp_skip=1)
c.set_target(*tgt)
acc = np.zeros(6); moves = []; nv = []
for rep in range(reps + 2):
    c.set_source(*src)
    x = x_guess.copy()
    for r in range(6):
        it = 1 if r < 3 else 2
        c.synchronize(); t0 = time.perf_counter(); n = c.associate(x, it); dt = time.perf_counter() - t0
        if rep >= 2: acc[r] += dt
        x_new = np.asarray(c.solve(x)[0])
        if rep == 2:
            moves.append((float(np.linalg.norm(x_new[3:] - x[3:])), float(np.linalg.norm(x_new[:3] - x[:3]))))
            nv.append(n)
        x = x_new
print("pair", k, "of drive 0: us per association round (call + sync):", " ".join("%.0f" % (1e6 * v / reps) for v in acc), flush=True)
print("pose change by the solve behind each round (m, rad):", " ".join("(%.4f, %.5f)" % m for m in moves))
print("valid rows:", nv, "| error of the guess (m, rad): %.4f %.5f" % (np.linalg.norm(x_guess[3:] - plan["x_true"][k][3:]), np.linalg.norm(x_guess[:3] - plan["x_true"][k][:3])))
c.close()
