#!/usr/bin/env python3
"""Dev tool (GPU box): what the index build of a target costs.  The 2M-point map of BASELINE config 4 (or the C2 scan with `c2`) is made
resident on the device, velo_set_target runs 20 times with every launch bracketed (timing level 3), and the kernels of the ingest +
build are listed with their average launch time; wall clock per set_target (incl. its final sync) beside it.
Use with VELO_LIB_PATH / the diagnostics build's switches for A/B runs (VELO_GRID_BLOCK=0/1)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import velo_amd
from velo_amd import api, synth
d = synth.scan_pair() if (len(sys.argv) > 1 and sys.argv[1] == "c2") else synth.scan_to_map(2_000_000)
lib = api.load_library(os.environ["VELO_LIB_PATH"]) if os.environ.get("VELO_LIB_PATH") else None
t = torch.from_numpy(np.ascontiguousarray(d["tgt_xyz"])).to("cuda:0")
c = api.Context(0, lib=lib, icp_skip=1)
c.set_timing(3)
reps = 20
for k in range(reps + 3):
    if k == 3:
        c.kernel_times(reset=True); c.synchronize(); t0 = time.perf_counter()
    c.set_target(t, d["tgt_off"]); c.synchronize()
c.synchronize(); wall = (time.perf_counter() - t0) / reps
tot = 0.0
for name, (ms, n, b) in sorted(c.kernel_times(reset=True).items(), key=lambda kv: -kv[1][0]):
    if n > 0:
        print("  %-32s avg %8.1f us  x %.1f per build  algorithmic %6.1f MB" % (name, 1e3 * ms / n, n / reps, b / n / 1e6)); tot += ms
print("kernels per build %.1f us; wall per velo_set_target (bracketed launches, one sync) %.1f us" % (1e3 * tot / reps, 1e6 * wall), flush=True)
c.close()
