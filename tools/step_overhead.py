#!/usr/bin/env python3
"""Dev tool (GPU box): where the HOST spends a drive step outside the library -- bench.py's DriveWalker.step with timers around the three
API calls (hint_next_frames, register_batch, pose_handoff) and around the raw C entry point inside register_batch.
   python tools/step_overhead.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import velo_amd  # noqa: F401
from velo_amd import api, synth
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 24
B = 8
drives = [synth.drive(steps + 2, seed=300 + s) for s in range(B)]
import torch
frames = [[(torch.from_numpy(np.ascontiguousarray(f[0])).cuda(), f[1]) for f in d["frames"]] for d in drives]
ctxs = [api.Context(0, icp_skip=1) for _ in range(B)]
w = bench.DriveWalker(api, ctxs, frames, 0)
lib = ctxs[0]._lib
acc = {"hint": 0.0, "register_batch": 0.0, "C call": 0.0, "handoff": 0.0}
raw = lib.velo_register_batch


def timed_raw(*a):
    t = time.perf_counter(); r = raw(*a); acc["C call"] += time.perf_counter() - t; return r


lib.velo_register_batch = timed_raw
for name, key in (("hint_next_frames", "hint"), ("register_batch", "register_batch"), ("pose_handoff", "handoff")):
    f = getattr(api, name)

    def wrap(*a, _f=f, _k=key, **kw):
        t = time.perf_counter(); r = _f(*a, **kw); acc[_k] += time.perf_counter() - t; return r
    setattr(api, name, wrap)
for _ in range(4):
    w.step()
for k in acc:
    acc[k] = 0.0
n = steps - 5
t0 = time.perf_counter()
for _ in range(n):
    w.step()
tot = time.perf_counter() - t0
print(f"{n} steps of {B} pairs: {1e6 * tot / n:.0f} us per step ({B * n / tot:.0f} pairs/s)")
print(f"  inside velo_register_batch (C)        {1e6 * acc['C call'] / n:7.0f} us")
print(f"  api.register_batch around it          {1e6 * (acc['register_batch'] - acc['C call']) / n:7.0f} us")
print(f"  api.hint_next_frames                  {1e6 * acc['hint'] / n:7.0f} us")
print(f"  api.pose_handoff                      {1e6 * acc['handoff'] / n:7.0f} us")
print(f"  the rest of DriveWalker.step          {1e6 * (tot - acc['register_batch'] - acc['hint'] - acc['handoff']) / n:7.0f} us")
