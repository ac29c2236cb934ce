#!/usr/bin/env python3
"""Dev tool (experiment): what would the tube kernel gain if a group's 64 queries were a compact PATCH (8 rings x 8 consecutive
points) instead of 64 consecutive points of one ring?  The source cloud is re-ordered on the host into patch order and handed over
with arbitrary ring offsets (with icp_skip = 1 the source's ring structure only fixes the query order), so the tables are the same
set of correspondences, permuted.  Prints wall clock per association round for both orders."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _diag  # noqa: E702  the A/B switches exist in the diagnostics build only
import velo_amd
from velo_amd import api, synth
d = synth.scan_pair()
x0, x1 = d["x0"], d["x_true"]
seq = [(1, x0), (1, x0 + 0.7 * (x1 - x0)), (1, x0 + 0.97 * (x1 - x0)), (2, x1 + 2e-3), (2, x1 + 2e-4), (2, x1)]
off = d["src_off"].astype(np.int64)
R = len(off) - 1


def patch_order(bh, seg_len):
    ring = np.repeat(np.arange(R), np.diff(off))
    k = np.arange(off[-1]) - off[ring]
    n = np.diff(off)[ring]
    band = ring // bh
    M = np.zeros(R // bh + 1, dtype=np.int64)
    for b in range(len(M)):
        rs = np.arange(b * bh, min((b + 1) * bh, R))
        if len(rs): M[b] = max(1, -(-int(np.diff(off)[rs].max()) // seg_len))
    seg = (k * M[band]) // n
    return np.lexsort((k, ring, seg, band))


def run(order, label):
    xyz = d["src_xyz"][order]
    fake_off = np.linspace(0, len(xyz), R + 1).astype(np.int32)
    c = api.Context(0, icp_skip=1)
    c.set_target(d["tgt_xyz"], d["tgt_off"])
    acc = np.zeros(len(seq)); reps = 20; nv = []
    for rep in range(reps + 3):
        c.set_source(xyz, fake_off)
        for j, (it, x) in enumerate(seq):
            c.synchronize(); t0 = time.perf_counter(); n = c.associate(x, it); dt = time.perf_counter() - t0
            if rep >= 3: acc[j] += dt
            if rep == 3: nv.append(n)
    print(f"{label:28s} us per round:", " ".join("%.0f" % (1e6 * v / reps) for v in acc), "| n_valid", nv, flush=True)
    c.close()


os.environ["VELO_PATCH_ORDER"] = "0"          # the library must not re-order the probe's lists


def morton_order(cell):
    p = d["src_xyz"].astype(np.float64)
    q = np.floor((p - p.min(0)) / cell).astype(np.int64)
    key = np.zeros(len(p), dtype=np.int64)
    for b in range(12):
        for a in range(3):
            key |= ((q[:, a] >> b) & 1) << (3 * b + a)
    return np.argsort(key, kind="stable")


run(np.arange(off[-1]), "ring order (as is)")
for cell in (0.18, 0.36, 0.72, 1.5):
    run(morton_order(cell), f"morton, cell {cell} m")
for bh, sl in ((8, 8), (4, 16), (16, 4), (2, 32), (8, 4), (4, 8)):
    run(patch_order(bh, sl), f"patches {bh} rings x {sl} pts")
