#!/usr/bin/env python3
"""Dev tool: the whole LiDAR-odometry loop (SURVEY 8(f) rows 1 + 2 around the path) on a synthetic drive: per frame one upload of
the raw records, ring segmentation on the device, promotion of the previous scan to target, frame_to_frame.  Prints frames/s and
the drift against the simulated trajectory."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import velo_amd
from velo_amd import odometry, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
frames, truth = synth.velodyne_sequence(n)
odo = odometry.LidarOdometer(0, icp_skip=1)
odo.push(frames[0]); odo.push(frames[1])                       # warm-up (allocations)
odo.close()
odo = odometry.LidarOdometer(0, icp_skip=1)
t0 = time.perf_counter()
for rec in frames:
    odo.push(rec)
dt = time.perf_counter() - t0
calls, misses = odo.ctx.chain_stats()
err = np.linalg.norm(odo.poses[-1][:3, 3] - truth[-1][:3, 3])
print(f"{n} frames of {len(frames[0])} points, icp_skip=1: {dt / (n - 1) * 1e3:.2f} ms per frame ({(n - 1) / dt:.0f} frames/s); "
      f"end-point drift {err * 100:.1f} cm over {np.linalg.norm(truth[-1][:3, 3]):.1f} m")
print(f"chain mode: {calls} calls enqueued as one chain, {misses} of them outran their predicted launch counts and were repeated host-driven")
