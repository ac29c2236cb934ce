#!/usr/bin/env python3
"""Dev tool: warm-started association rounds (VELO_WARM_START=1, default) must give the tables of cold rounds bit for bit.
Walks a pose sequence like frame_to_frame does (iter 1 x3, iter 2 x3) on one context and compares every round with a
context that never has seeds, then compares whole frame_to_frame runs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _diag  # noqa: E702  the A/B switches exist in the diagnostics build only
import velo_amd
from velo_amd import api, synth

d = synth.scan_pair()
os.environ["VELO_WARM_START"] = "1"; warm = api.Context(0, icp_skip=1)
os.environ["VELO_WARM_START"] = "0"; cold = api.Context(0, icp_skip=1)
for c in (warm, cold):
    c.set_target(d["tgt_xyz"], d["tgt_off"]); c.set_source(d["src_xyz"], d["src_off"])
x0, x1 = d["x0"], d["x_true"]
seq = [(1, x0), (1, x0 + 0.3 * (x1 - x0)), (1, x0 + 0.8 * (x1 - x0)), (2, x1), (2, x1 + 1e-4), (2, x1),
       (1, np.array([0.02, -0.01, 0.03, 0.4, -0.3, 1.5])), (2, x0), (1, x1)]
ok = True
for it, x in seq:
    nw, nc = warm.associate(x, it), cold.associate(x, it)
    tw, tc = warm.correspondences(), cold.correspondences()
    same = nw == nc and tw.tobytes() == tc.tobytes()
    ok &= same
    print("iter", it, "n_valid", nw, nc, "identical" if same else "MISMATCH")
a = warm.frame_to_frame(x0); b = cold.frame_to_frame(x0)
print("f2f identical:", np.array_equal(a[0], b[0]), a[0])
print("ALL IDENTICAL" if ok and np.array_equal(a[0], b[0]) else "MISMATCH")
