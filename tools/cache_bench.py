#!/usr/bin/env python3
"""Dev tool: what the device-resident scan cache saves per look-up on the 120k-point scan: a fresh target
(host upload + index build / device-resident cloud + index build) against velo_cache_load (hit, index reused) and the
store; same for the source side."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import velo_amd
from velo_amd import api, synth

d = synth.scan_pair()
dev_t = torch.from_numpy(d["tgt_xyz"]).to("cuda:0"); dev_s = torch.from_numpy(d["src_xyz"]).to("cuda:0")
c = api.Context(0, icp_skip=1)
cache = api.ScanCache(0, 50)
c.set_target(d["tgt_xyz"], d["tgt_off"]); c.set_source(d["src_xyz"], d["src_off"])
cache.store(0, c, True); cache.store(1, c, False)

def timed(fn, reps=50):
    for _ in range(5): fn()
    c.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    c.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6

rows = [
    ("target: host rings -> upload + index", lambda: c.set_target(d["tgt_xyz"], d["tgt_off"])),
    ("target: device-resident cloud -> index", lambda: c.set_target(dev_t, d["tgt_off"])),
    ("target: cache hit (index reused)", lambda: cache.load(0, c, True)),
    ("target: cache entry without index", lambda: cache.load(1, c, True)),
    ("target: cache hit again (restores frame 0)", lambda: cache.load(0, c, True)),
    ("store target (cloud + index), node recycled", lambda: cache.store(0, c, True)),
    ("source: host rings -> upload + query list", lambda: c.set_source(d["src_xyz"], d["src_off"])),
    ("source: cache hit", lambda: cache.load(1, c, False)),
]
c.set_target(d["tgt_xyz"], d["tgt_off"])
for name, fn in rows:
    print("%-44s %8.1f us" % (name, timed(fn)))
x0 = d["x0"]
cache.load(0, c, True); cache.load(1, c, False)
x, _, _ = c.frame_to_frame(x0)
print("registration from cached scans: x =", np.round(x, 5))
