#!/usr/bin/env python3
"""Dev tool (GPU box): how many LM launches a lock-step pairing of the 8 drives costs per step -- sum over the six solves of the larger of the two
contexts' evaluation counts -- for the fixed pairing (0,1)(2,3)(4,5)(6,7) against a pairing chosen per step from the PREVIOUS step's counts
(contexts sorted by their predicted total, neighbours paired)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py"]
import numpy as np
import bench
import velo_amd  # noqa: F401
from velo_amd import api
B, K = 8, 25
drives = bench.make_drives(B, K + 1, 0)
ctxs = [api.Context(0, icp_skip=1) for _ in range(B)]
w = bench.DriveWalker(api, ctxs, [p["frames"] for p in drives], 0)
E = []
for _ in range(K):
    xs, Ts, Ss = w.step()
    E.append(np.array([[s.solves[r].evaluations for r in range(6)] for s in Ss]))
E = np.array(E)          # [step][ctx][solve]
def cost(pairs, e):
    return sum(int(np.maximum(e[i], e[j]).sum()) for i, j in pairs)
fixed = [(0, 1), (2, 3), (4, 5), (6, 7)]
tot_fixed = tot_sorted = tot_oracle = live = 0
for k in range(1, K):
    e = E[k]
    tot_fixed += cost(fixed, e)
    order = np.argsort(E[k - 1].sum(axis=1))
    tot_sorted += cost([(order[2 * g], order[2 * g + 1]) for g in range(4)], e)
    order = np.argsort(e.sum(axis=1))
    tot_oracle += cost([(order[2 * g], order[2 * g + 1]) for g in range(4)], e)
    live += int(e.sum())
n = (K - 1)
print(f"per step: evaluations of the 8 pairs {live / n:.1f}; live launches of 4 groups: fixed pairing {tot_fixed / n:.1f}, paired by the previous step's totals {tot_sorted / n:.1f}, "
      f"paired by this step's own totals (not knowable) {tot_oracle / n:.1f}; lower bound (no pairing loss) {live / n / 2:.1f}")
print("evaluations per solve, mean over steps and drives:", np.round(E[1:].mean(axis=(0, 1)), 2), " std over drives of the per-step totals:", np.round(E[1:].sum(axis=2).std(axis=1).mean(), 2))
for c in ctxs: c.close()
