#!/usr/bin/env python3
"""Dev tool: wall clock per association round (call incl. sync) along frame_to_frame's pose sequence on C2: cold round, two warm
rounds at gate 0.5 m^2, three at 0.031 m^2.  Use with the library's environment switches / VELO_LIB_PATH for A/B runs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import velo_amd
from velo_amd import api, synth
d = synth.scan_to_map(2_000_000) if (len(sys.argv) > 1 and sys.argv[1] == "c4") else synth.scan_pair()
x0, x1 = d["x0"], d["x_true"]
seq = [(1, x0), (1, x0 + 0.7 * (x1 - x0)), (1, x0 + 0.97 * (x1 - x0)), (2, x1 + 2e-3), (2, x1 + 2e-4), (2, x1)]
c = api.Context(0, icp_skip=1)
c.set_target(d["tgt_xyz"], d["tgt_off"])
acc = np.zeros(len(seq)); reps = 10 if (len(sys.argv) > 1 and sys.argv[1] == "c4") else 30; nv = []
for rep in range(reps + 3):
    c.set_source(d["src_xyz"], d["src_off"])
    for k, (it, x) in enumerate(seq):
        c.synchronize(); t0 = time.perf_counter(); n = c.associate(x, it); dt = time.perf_counter() - t0
        if rep >= 3: acc[k] += dt
        if rep == 3: nv.append(n)
print("us per round:", " ".join("%.0f" % (1e6 * v / reps) for v in acc), "| n_valid", nv, flush=True)
c.close()
