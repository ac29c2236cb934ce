#!/usr/bin/env python3
"""Dev tool (GPU box): do the B drives of bench.py's step have to advance in ONE library call?  H host threads, each driving B / H
drives through its own velo_register_batch call per frame (the reference runs its sequences as parallel processes, run.fish:2), against
the one-call step.   python tools/halves_bench.py [steps] [warmup]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 5
B = 8
drives = bench.make_drives(B, warm + steps + 1)
import torch
import velo_amd
from velo_amd import api
dev = torch.device("cuda", 0)
frames = [[(torch.from_numpy(np.ascontiguousarray(f[0])).to(dev), f[1]) for f in p["frames"]] for p in drives]
torch.cuda.synchronize()
for H in (1, 2, 4, 1, 2):
    ctxs = [api.Context(0, icp_skip=1) for _ in range(B)]
    per = B // H
    walkers = [bench.DriveWalker(api, ctxs[h * per:(h + 1) * per], frames[h * per:(h + 1) * per], 0) for h in range(H)]
    bar = threading.Barrier(H + 1)
    def run(w):
        for _ in range(warm): w.step()
        bar.wait()
        for _ in range(steps): w.step()
        bar.wait()
    th = [threading.Thread(target=run, args=(w,)) for w in walkers]
    for t in th: t.start()
    bar.wait(); t0 = time.perf_counter()
    bar.wait(); dt = time.perf_counter() - t0
    for t in th: t.join()
    print(f"H={H}: {steps * B / dt:8.1f} pairs/s  ({1e3 * dt / steps:.3f} ms per frame of all {B} drives)", flush=True)
    for c in ctxs: c.close()
