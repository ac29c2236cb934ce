#!/usr/bin/env python3
"""Dev tool: times one association round (wall clock around velo_associate, which syncs) for kernel variants.
Usage: python tools/assoc_bench.py [c2|c4] [variants...]   variants like 0:3 16:3 16:6 (assoc_variant:cluster_w)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import velo_amd
from velo_amd import api, synth

wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
variants = sys.argv[2:] or ["0:3", "1:3", "2:3", "4:1", "4:3", "4:6", "8:3"]
d = synth.scan_to_map(2_000_000) if wl == "c4" else synth.scan_pair()
ref = None
for v in variants:
    av, cw, *rest = v.split(":")
    os.environ["VELO_ASSOC_VARIANT"], os.environ["VELO_CLUSTER_W"] = av, cw
    os.environ["VELO_DEBUG_SKIP"] = rest[0] if rest else "0"
    c = api.Context(0, icp_skip=1)
    t0 = time.perf_counter(); c.set_target(d["tgt_xyz"], d["tgt_off"]); c.synchronize(); t_tgt = time.perf_counter() - t0
    c.set_source(d["src_xyz"], d["src_off"])
    out = []
    for it, x in ((1, d["x0"]), (1, d["x_true"]), (2, d["x_true"])):
        c.associate(x, it)
        reps = 3 if av == "0" else 10
        t0 = time.perf_counter()
        for _ in range(reps):
            nv = c.associate(x, it)
        dt = (time.perf_counter() - t0) / reps
        out.append((it, nv, dt * 1e6))
    cor = c.correspondences()
    key = (cor["valid"].sum(), cor["idx_i"].sum(), cor["idx_j"].sum(), cor["idx_k"].sum())
    if ref is None: ref = key
    print(f"variant {v:>5}: set_target {t_tgt*1e3:.2f} ms | " + " | ".join(f"iter{it} valid={nv} {us:.0f} us" for it, nv, us in out) + f" | same={key == ref}", flush=True)
    c.close()
