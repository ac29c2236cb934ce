import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import velo_amd
from velo_amd import api, synth
d = synth.scan_pair()
c = api.Context(0)          # reference constants: icp_skip = 200
c.set_target(d["tgt_xyz"], d["tgt_off"]); c.set_source(d["src_xyz"], d["src_off"])
for _ in range(5): x, T, s = c.frame_to_frame(d["x0"])
t0 = time.perf_counter()
for _ in range(50): x, T, s = c.frame_to_frame(d["x0"])
dt = (time.perf_counter() - t0) / 50
print("C1 (icp_skip=200, %d queries): %.3f ms per frame_to_frame, %d LM evaluations, x=%s" % (s.n_queries, dt * 1e3, sum(s.solves[k].evaluations for k in range(s.n_solves)), np.round(x, 4)))
t0 = time.perf_counter()
for _ in range(20):
    c.set_target(d["tgt_xyz"], d["tgt_off"]); c.set_source(d["src_xyz"], d["src_off"]); x, T, s = c.frame_to_frame(d["x0"])
print("incl. host upload + index build: %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
