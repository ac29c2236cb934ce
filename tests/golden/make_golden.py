#!/usr/bin/env python3
"""Generates tests/golden/*.npz.

The reference has no tests, fixtures or golden vectors and cannot be built or imported here, so these
vectors are REGRESSION fixtures produced by this repository's own CPU restatement (oracle/velo_oracle.cpp) -- they pin
the restatement and the HIP path against silent drift; they do not pin either to the reference ("parity unpinned").
Inputs are stored too, so the fixtures stay valid if the synthetic generator changes.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import oracle_lib as ol  # noqa: E402
import velo_amd  # noqa: E402,F401
from velo_amd import api, synth  # noqa: E402


def mini_pair():
    d = synth.scan_pair(n_beams=16, n_azimuth=64)
    vis = api.matches_from_dict(synth.stereo_matches(n_per_cam=24, mix="all"))
    o = ol.Oracle(icp_skip=1)
    o.set_target(d["tgt_xyz"], d["tgt_off"])
    o.set_source(d["src_xyz"], d["src_off"])
    o.set_visual(vis)
    out = dict(src_xyz=d["src_xyz"], src_off=d["src_off"], tgt_xyz=d["tgt_xyz"], tgt_off=d["tgt_off"],
               x0=d["x0"], x_true=d["x_true"], matches=vis)
    poses = np.stack([d["x0"], d["x_true"], np.array([0.02, -0.01, 0.03, 0.1, -0.05, 1.2])])
    out["poses"] = poses
    for it in (1, 2):
        o.build_visual(poses[1], it)
        out[f"good_iter{it}"] = o.good_matches()
        for k, x in enumerate(poses):
            o.associate(x, it)
            out[f"corr_iter{it}_pose{k}"] = o.correspondences()
            o.build_visual(x, it)
            c, Hm, g = o.evaluate(x)
            out[f"eval_iter{it}_pose{k}"] = np.concatenate([[c], Hm.ravel(), g])
    x, T, s = o.frame_to_frame(d["x0"])
    out["f2f_x"] = x
    out["f2f_T"] = T
    out["f2f_solves"] = np.array([[s.solves[k].termination, s.solves[k].lm_iterations, s.solves[k].evaluations,
                                   s.solves[k].n_icp_valid, s.solves[k].n_visual_blocks] for k in range(s.n_solves)])
    out["f2f_costs"] = np.array([[s.solves[k].initial_cost, s.solves[k].final_cost] for k in range(s.n_solves)])
    out["f2f_bytes"] = np.array([s.algorithmic_bytes, s.assoc_bytes], dtype=np.uint64)
    return out


def functor_vectors():
    rng = np.random.default_rng(2024)
    kinds, consts, xs, rs, Js = [], [], [], [], []
    ncon = {0: 6, 1: 8, 2: 8, 3: 7, 4: 9}
    for kind in (0, 1, 2, 3, 4):
        for trial in range(6):
            c = np.zeros(9)
            c[:ncon[kind]] = rng.normal(size=ncon[kind]) * (0.3 if kind == 3 else 3.0)
            x = np.concatenate([rng.normal(size=3) * (0.0 if trial == 0 else 1e-9 if trial == 1 else 0.3), rng.normal(size=3)])
            r, J = ol.functor(kind, c, x)
            rr, JJ = np.zeros(3), np.zeros((3, 6))
            rr[:len(r)] = r
            JJ[:len(r)] = J
            kinds.append(kind); consts.append(c); xs.append(x); rs.append(rr); Js.append(JJ)
    return dict(kind=np.array(kinds), c=np.array(consts), x=np.array(xs), r=np.array(rs), J=np.array(Js))


def depth_mini():
    """a 16-ring x 700-azimuth scan, camera 1 (non-zero cam_t), 200 keypoints."""
    d = synth.scan_pair(n_beams=16, n_azimuth=700)
    w = synth.cam_window()
    t = synth.CAM_TRANS[1]
    proj, pts, off = ol.project_lidar(d["tgt_xyz"], d["tgt_off"], t, w)
    kps = synth.keypoints_in_window(200, seed=5)
    kd, has = ol.depth_association(proj, pts, off, kps, synth.DEPTH_ASSOC_THRESH)
    return dict(xyz=d["tgt_xyz"], off=d["tgt_off"], cam_t=t, window=w, proj_xy=proj, proj_pts=pts, proj_off=off,
                keypoints=kps, thresh=np.float64(synth.DEPTH_ASSOC_THRESH), kp_with_depth=kd, has_depth=has)


def triangulation_mini():
    """64 landmarks over 8 frames with the oracle's points and per-landmark solver summaries."""
    pr = synth.triangulation_problem(64, n_frames=8, seed=21)
    pts, res = ol.triangulate_points(pr["camera_poses"], pr["cam_trans"], pr["obs"], pr["obs_offsets"], pr["points0"], pr["initial_guess"])
    out = {k: pr[k] for k in ("camera_poses", "cam_trans", "obs", "obs_offsets", "points0", "initial_guess")}
    out["points"] = pts
    out["results"] = res
    return out


FIXTURES = {"mini_pair": mini_pair, "functors": functor_vectors, "depth_mini": depth_mini, "triangulation_mini": triangulation_mini}

if __name__ == "__main__":
    for name in (sys.argv[1:] or list(FIXTURES)):            # python make_golden.py [fixture ...]
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **FIXTURES[name]())
        print(name + ".npz", os.path.getsize(path), "bytes")
