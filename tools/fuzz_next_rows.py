#!/usr/bin/env python3
"""Dev tool: randomized sweep of the SURVEY 8(f) rows against the oracle: camera projection + keypoint depth on scans of random
size for both cameras, and batched triangulation problems of random size / frame count / noise.  The device ring segmenter on sweeps with random drop-outs.  Bit-exact comparisons."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import velo_amd
from velo_amd import api, synth
import oracle_lib as O


def run(n_seeds, first_seed=0):
    checked = 0
    for seed in range(first_seed, first_seed + n_seeds):
        rng = np.random.default_rng(7000 + seed)
        d = synth.scan_pair(n_beams=int(rng.choice([8, 16, 32, 64])), n_azimuth=int(rng.integers(150, 900)))
        w = synth.cam_window()
        c = api.Context(0)
        c.set_target(d["tgt_xyz"], d["tgt_off"])
        for cam in (0, 1):
            t = synth.CAM_TRANS[cam]
            n = c.project_lidar(True, t, w)
            got = c.projection()
            want = O.project_lidar(d["tgt_xyz"], d["tgt_off"], t, w)
            assert n == len(want[0]) and np.array_equal(got[2], want[2]), ("projection offsets", seed, cam)
            assert np.array_equal(got[0].view(np.uint32), want[0].view(np.uint32)) and np.array_equal(got[1].view(np.uint32), want[1].view(np.uint32)), ("projection", seed, cam)
            kps = synth.keypoints_in_window(int(rng.integers(1, 3000)), seed=int(rng.integers(1, 10 ** 6)))
            thr = float(rng.choice([synth.DEPTH_ASSOC_THRESH, 0.05, 0.2]))
            kd, has = c.depth_association(kps, thr)
            wkd, whas = O.depth_association(*want, kps, thr)
            assert np.array_equal(has, whas) and np.array_equal(kd.view(np.uint32), wkd.view(np.uint32)), ("depth", seed, cam)
            checked += 2
        # row 1: the device ring segmenter on a sweep with random drop-outs (ragged rings), bit for bit vs the numpy restatement
        scene = synth.Scene(int(rng.integers(0, 5)))
        pts = synth.hdl64_scan(scene, synth.pose_matrix(float(rng.normal(0, 0.05)), 0, 0, (float(rng.normal(0, 1)), float(rng.normal(0, 0.3)), 0)),
                               noise_seed=int(rng.integers(1, 10 ** 6)), n_beams=int(rng.choice([8, 16, 32, 64])), n_azimuth=int(rng.integers(100, 700)))
        pts = pts[synth.uniform01(int(rng.integers(1, 10 ** 6)), len(pts)) > float(rng.uniform(0.0, 0.4))]
        rec = np.zeros((len(pts), 4), dtype=np.float32); rec[:, :3] = pts
        want_xyz, want_off = synth.segment_points(pts)
        as_target = bool(rng.integers(0, 2))
        c.set_scan_velodyne(as_target, rec, synth.VELO_TO_CAM)
        assert np.array_equal(c.ring_offsets(as_target), want_off), ("segmenter offsets", seed)
        assert np.array_equal(c.cloud(as_target).view(np.uint32), want_xyz.view(np.uint32)), ("segmenter cloud", seed)
        checked += 1
        pr = synth.triangulation_problem(int(rng.integers(1, 1500)), n_frames=int(rng.integers(3, 50)), seed=int(rng.integers(1, 10 ** 6)))
        args = (pr["camera_poses"], pr["cam_trans"], pr["obs"], pr["obs_offsets"], pr["points0"], pr["initial_guess"])
        gp, gr = c.triangulate_points(*args)
        wp, wr = O.triangulate_points(*args)
        assert np.array_equal(gp.view(np.uint32), wp.view(np.uint32)), ("triangulated points", seed, int(np.argmax(np.any(gp != wp, axis=1))))
        for f in ("n_solves", "termination", "lm_iterations", "evaluations"):
            assert np.array_equal(gr[f], wr[f]), ("triangulation summary " + f, seed)
        checked += 1
        c.close()
    return checked


if __name__ == "__main__":
    n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    print("fuzz next rows: %d seeds, %d comparisons, all equal" % (n_seeds, run(n_seeds)))
