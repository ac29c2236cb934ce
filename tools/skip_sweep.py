#!/usr/bin/env python3
"""Dev tool: frame_to_frame latency on the C2 pair for icp_skip = 1 ... 200 (120k ... 640 queries per round) -- where the
one-wave-per-query kernel (VELO_ASSOC_DIRECT_MAX) takes over from the tube kernel.  Run it once per setting of the variable."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _diag  # noqa: E702  the A/B switches exist in the diagnostics build only
import velo_amd
from velo_amd import api, synth
d = synth.scan_pair()
for skip in (1, 4, 8, 12, 16, 24, 32, 64, 200):
    c = api.Context(0, icp_skip=skip)
    c.set_target(d["tgt_xyz"], d["tgt_off"]); c.set_source(d["src_xyz"], d["src_off"])
    c.set_timing(True)
    for _ in range(5): x, T, s = c.frame_to_frame(d["x0"])
    t0 = time.perf_counter()
    for _ in range(30): x, T, s = c.frame_to_frame(d["x0"])
    dt = (time.perf_counter() - t0) / 30
    print(f"icp_skip {skip:3d}: {s.n_queries:6d} queries, {dt * 1e3:.3f} ms per call, association {s.assoc_kernel_ms / max(s.assoc_kernel_launches, 1) * 1e3:.1f} us per round", flush=True)
    c.close()
