import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.argv = ["bench.py"]
import numpy as np
import bench
import velo_amd
from velo_amd import api
B, W, K = 8, 5, 20
drives = bench.make_drives(B, W + K + 1, 0)
import torch
dev = torch.device("cuda", 0)
frames = [[(torch.from_numpy(np.ascontiguousarray(f[0])).to(dev), f[1]) for f in p["frames"]] for p in drives]
torch.cuda.synchronize()
for mode in ("seq", "step", "seq", "step"):
    ctxs = [api.Context(0, icp_skip=1) for _ in range(B)]
    for c in ctxs: c.set_timing(int(os.environ.get("TIMING", "2")))
    w = bench.DriveWalker(api, ctxs, frames, 0)
    if mode == "seq":
        w.walk(w.prepare(W))
        prep = w.prepare(K)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        xs, Ts, Ss = w.walk(prep)
        t1 = time.perf_counter()
    else:
        for _ in range(W): w.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K): w.step()
        t1 = time.perf_counter()
    print(mode, "total ms", round(1e3 * (t1 - t0), 2), "per step", round(1e3 * (t1 - t0) / K, 3), flush=True)
    for c in ctxs: c.close()
