#!/usr/bin/env python3
"""Dev tool: association-kernel time per variant, measured with the library's HIP events inside frame_to_frame
(6 rounds: 3 at gate 0.5, 3 at gate 0.03125), median over repetitions.
Usage: python tools/assoc_bench.py [c2|c4] [variants...]   variant = assoc_variant:cluster_w[:debug_skip[:map]]  (map -> VELO_XCD_MAP for the box kernels, VELO_TUBE_MAP for the tube kernel)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _diag  # noqa: E702  the A/B switches exist in the diagnostics build only
import velo_amd
from velo_amd import api, synth

wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
variants = sys.argv[2:] or ["4:6"]
d = synth.scan_to_map(2_000_000) if wl == "c4" else synth.scan_pair()
ref = None
ctxs = []
for v in variants:
    av, cw, *rest = v.split(":")
    os.environ["VELO_ASSOC_VARIANT"], os.environ["VELO_CLUSTER_W"] = av, cw
    os.environ["VELO_DEBUG_SKIP"] = rest[0] if rest else "0"
    os.environ["VELO_XCD_MAP"] = rest[1] if len(rest) > 1 else "0"
    os.environ["VELO_TUBE_MAP"] = rest[1] if len(rest) > 1 else "1"
    c = api.Context(0, icp_skip=1)
    c.set_timing(True)
    c.set_target(d["tgt_xyz"], d["tgt_off"]); c.set_source(d["src_xyz"], d["src_off"])
    ctxs.append((v, c))
reps = 3 if wl == "c4" else 15
samples = {v: [] for v, _ in ctxs}
walls = {v: [] for v, _ in ctxs}
for r in range(reps):                     # interleaved rounds in one process (guide rule 24)
    for v, c in ctxs:
        t0 = time.perf_counter()
        x, T, s = c.frame_to_frame(d["x0"])
        walls[v].append(time.perf_counter() - t0)
        samples[v].append(s.assoc_kernel_ms / max(s.assoc_kernel_launches, 1) * 1e3)
        key = tuple(np.round(x, 12))
        if ref is None: ref = key
        if key != ref: print("!! variant", v, "differs", x)
for v, c in ctxs:
    a = np.array(samples[v][1:]); w = np.array(walls[v][1:])
    print(f"variant {v:>8}: assoc kernel median {np.median(a):7.1f} us  min {a.min():7.1f} | f2f wall median {np.median(w)*1e3:6.2f} ms min {w.min()*1e3:6.2f}", flush=True)
    c.close()
