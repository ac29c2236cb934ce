#!/usr/bin/env python3
"""Dev tool: correspondence tables of association-kernel variants must be identical (index- and bit-exact).
Usage: python tools/variant_check.py base_variant:cluster_w variant:cluster_w ...   (C2 scan pair, iters 1 and 2, 3 poses)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _diag  # noqa: E702  the A/B switches exist in the diagnostics build only
import velo_amd
from velo_amd import api, synth

d = synth.scan_pair()
variants = sys.argv[1:] or ["4:6", "5:6"]
ctxs = []
for v in variants:
    av, cw = v.split(":")
    os.environ["VELO_ASSOC_VARIANT"], os.environ["VELO_CLUSTER_W"] = av, cw
    c = api.Context(0, icp_skip=1)
    c.set_target(d["tgt_xyz"], d["tgt_off"]); c.set_source(d["src_xyz"], d["src_off"])
    ctxs.append((v, c))
poses = [d["x0"], d["x_true"], np.array([0.02, -0.01, 0.03, 0.4, -0.3, 1.5])]
ok = True
for it in (1, 2):
    for k, x in enumerate(poses):
        ref = None
        for v, c in ctxs:
            n = c.associate(x, it)
            t = c.correspondences()
            if ref is None:
                ref = (n, t); continue
            same = n == ref[0] and t.tobytes() == ref[1].tobytes()
            if not same:
                ok = False
                bad = [f for f in t.dtype.names if not np.array_equal(t[f], ref[1][f])]
                print(f"iter {it} pose {k}: variant {v} differs from {variants[0]} in {bad}; n_valid {n} vs {ref[0]}")
print("identical" if ok else "MISMATCH")
