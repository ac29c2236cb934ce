#!/usr/bin/env python3
"""Dev tool: what phase 2 (queries whose bound reaches beyond their own cell +- 1) costs in a warm association round.
Runs the diagnostic instantiation twice -- with a harmless debug bit, and with the bit that skips phase 2 (wrong results) --
and prints the mean launch time per round of frame_to_frame's pose sequence."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _diag  # noqa: E702  the A/B switches exist in the diagnostics build only
    import numpy as np, time
    import velo_amd
    from velo_amd import api, synth
    d = synth.scan_pair()
    x0, x1 = d["x0"], d["x_true"]
    seq = [(1, x0), (1, x0 + 0.7 * (x1 - x0)), (1, x0 + 0.97 * (x1 - x0)), (2, x1 + 2e-3), (2, x1 + 2e-4), (2, x1)]
    c = api.Context(0, icp_skip=1)
    c.set_target(d["tgt_xyz"], d["tgt_off"])
    acc = np.zeros(len(seq))
    reps = 30
    for rep in range(reps + 3):
        c.set_source(d["src_xyz"], d["src_off"])
        for k, (it, x) in enumerate(seq):
            c.synchronize(); t0 = time.perf_counter(); c.associate(x, it); dt = time.perf_counter() - t0
            if rep >= 3: acc[k] += dt
    print(" ".join("%.0f" % (1e6 * v / reps) for v in acc), flush=True)
    c.close(); sys.exit(0)
for name, bit in (("all phases", "64"), ("no phase 2", "1024")):
    out = subprocess.run([sys.executable, __file__, "--child"], env=dict(os.environ, VELO_DEBUG_SKIP=bit), capture_output=True, text=True)
    print("%-12s us per round (call incl. sync):" % name, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-400:])
