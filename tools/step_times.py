#!/usr/bin/env python3
"""Dev tool (GPU box): wall time of EVERY drive step (bench.py's DriveWalker, 8 drives, optionally with stereo matches) -- is a low run one
slow step or all of them?   python tools/step_times.py [c2|c3] [runs]"""
import os, sys, time, gc
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import velo_amd  # noqa: F401
from velo_amd import api, synth
import bench

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
B, NF = 8, 26
drives = bench.make_drives(B, NF)
host = os.environ.get("STEP_TIMES_HOST")              # frames stay in pageable host memory (bench.py --host-inputs)
frames = [[(np.ascontiguousarray(f[0]) if host else torch.from_numpy(np.ascontiguousarray(f[0])).cuda(), f[1]) for f in d["frames"]] for d in drives]
vis = [[synth.stereo_matches(1000, seed=3 + 1000 * i + k, x_true=drives[i]["x_true"][k]) for k in range(NF - 1)] for i in range(B)] if wl == "c3" else None
ctxs = [api.Context(0, icp_skip=1) for _ in range(B)]
w = bench.DriveWalker(api, ctxs, frames, 0, vis)
for r in range(runs):
    w.restart()
    gc.collect(); gc.disable()
    ts = []
    for _ in range(NF - 1):
        t = time.perf_counter(); w.step(); ts.append(1e3 * (time.perf_counter() - t))
        if os.environ.get("VELO_ALLOC_TRACE") or os.environ.get("VELO_BATCH_TRACE"):
            print(f"[step {len(ts)}] {ts[-1]:.2f} ms", file=sys.stderr, flush=True)
    gc.enable()
    t20 = ts[5:]
    print(f"run {r}: steps 6..25: {B * len(t20) / (1e-3 * sum(t20)):.0f} pairs/s; ms per step: " + " ".join(f"{x:.2f}" for x in ts), flush=True)
