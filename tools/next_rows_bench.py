#!/usr/bin/env python3
"""Dev tool: wall-clock of the SURVEY 8(f) row-3 entry points (camera projection of a 120k-pt scan, depth for 5,000 keypoints)
and of row 4 (triangulation of 3,000 and 50,000 landmarks) on the GPU next to the CPU oracle on the same inputs.  Kernel times come from `rocprofv3 --kernel-trace --stats` of this script."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import velo_amd
from velo_amd import api, synth

d = synth.scan_pair()
w = synth.cam_window()
kps = synth.keypoints_in_window(5000, seed=31)
c = api.Context(0)
c.set_target(d["tgt_xyz"], d["tgt_off"])
reps = 30
for _ in range(3):
    c.project_lidar(True, synth.CAM_TRANS[0], w); c.depth_association(kps)
t0 = time.perf_counter()
for _ in range(reps):
    n = c.project_lidar(True, synth.CAM_TRANS[0], w)
t1 = time.perf_counter()
for _ in range(reps):
    kd, has = c.depth_association(kps)
t2 = time.perf_counter()
out = {"project_lidar_ms": (t1 - t0) / reps * 1e3, "depth_association_ms": (t2 - t1) / reps * 1e3, "kept_points": n,
       "keypoints": len(kps), "with_depth": int((has >= 0).sum())}
tri = {n: synth.triangulation_problem(n, n_frames=12, seed=13) for n in (3000, 50000)}
tri_args = {n: (p["camera_poses"], p["cam_trans"], p["obs"], p["obs_offsets"], p["points0"], p["initial_guess"]) for n, p in tri.items()}
for n, a in tri_args.items():
    c.triangulate_points(*a)
    t0 = time.perf_counter()
    for _ in range(10):
        gp, gr = c.triangulate_points(*a)
    out[f"triangulate_{n}_ms"] = (time.perf_counter() - t0) / 10 * 1e3
    out[f"triangulate_{n}_obs"] = int(len(a[2]))
    out[f"triangulate_{n}_evaluations"] = int(gr["evaluations"].sum())
if "--no-cpu" not in sys.argv:
    import oracle_lib as O
    t0 = time.perf_counter()
    for _ in range(reps):
        proj, pts, off = O.project_lidar(d["tgt_xyz"], d["tgt_off"], synth.CAM_TRANS[0], w)
    t1 = time.perf_counter()
    for _ in range(reps):
        O.depth_association(proj, pts, off, kps)
    t2 = time.perf_counter()
    out["cpu_project_lidar_ms"] = (t1 - t0) / reps * 1e3
    out["cpu_depth_association_ms"] = (t2 - t1) / reps * 1e3
    for n, a in tri_args.items():
        t0 = time.perf_counter()
        wp, wr = O.triangulate_points(*a)
        out[f"cpu_triangulate_{n}_ms"] = (time.perf_counter() - t0) * 1e3
        gp, gr = c.triangulate_points(*a)
        out[f"triangulate_{n}_bit_identical"] = float(np.all(gp.view(np.uint32) == wp.view(np.uint32), axis=1).mean())
print(json.dumps(out))
