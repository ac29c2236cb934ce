"""CPU tests of the synthetic HDL-64E / KITTI-layout inputs and the reference's ring segmenter."""
import os
import tempfile

import numpy as np

import velo_amd  # noqa: F401
from velo_amd import synth


def test_full_scan_shape_and_ring_rule():
    d = synth.scan_pair()
    assert d["src_xyz"].shape == (120000, 3) and d["src_xyz"].dtype == np.float32
    assert len(d["src_off"]) == 65 and np.all(np.diff(d["src_off"]) == 1875)      # 64 rings x 1875 (kitti.h:166 rule)
    assert len(d["tgt_off"]) == 65 and d["tgt_off"][-1] == 120000
    # camera-0 frame: y down, ground ~ +1.65 m below the camera, z forward
    assert 1.5 < np.percentile(d["tgt_xyz"][:, 1], 99) < 1.8
    assert np.all(np.isfinite(d["src_xyz"]))


def test_generator_is_deterministic_and_counter_based():
    a = synth.scan_pair(n_beams=8, n_azimuth=32)
    b = synth.scan_pair(n_beams=8, n_azimuth=32)
    assert np.array_equal(a["src_xyz"], b["src_xyz"]) and np.array_equal(a["tgt_xyz"], b["tgt_xyz"])
    u = synth.uniform01(7, 1000)
    assert np.array_equal(u[:10], synth.uniform01(7, 10)) and 0 < u.min() and u.max() < 1
    g = synth.normal01(3, 200000)
    assert abs(g.mean()) < 0.01 and abs(g.std() - 1) < 0.01


def test_segmenter_reorder_preserves_cyclic_adjacency():
    n = 10
    az = (np.arange(n) + 0.5) * 2 * np.pi / n
    ring = np.stack([np.cos(az), np.sin(az), np.zeros(n)], 1).astype(np.float32)
    pts = np.concatenate([ring, ring * 2])                     # two rings: x>0 & y sign flip between them
    xyz, off = synth.segment_points(pts, np.eye(4, dtype=np.float32))
    assert list(off) == [0, 10, 20]
    src = n - 1 - ((np.arange(n) + n // 2) % n)                # kitti.h:180
    assert np.array_equal(xyz[:n], ring[src])
    idx = src
    assert all(((idx[i] - idx[(i + 1) % n]) % n) in (1, n - 1) for i in range(n))


def test_kitti_bin_roundtrip():
    p = (np.random.default_rng(0).normal(size=(100, 3)) * 10).astype(np.float32)
    with tempfile.TemporaryDirectory() as td:
        f = os.path.join(td, "000000.bin")
        synth.write_kitti_bin(f, p)
        assert os.path.getsize(f) == 100 * 16                 # float32 x,y,z,reflectance (kitti.h:130-148)
        assert np.array_equal(synth.read_kitti_bin(f), p)


def test_true_motion_maps_current_into_previous_frame():
    """x_true applied to the (noise-free) current scan puts its points back on the scene surfaces of the previous frame."""
    d = synth.scan_pair(n_beams=16, n_azimuth=256, sigma=0.0)
    R = synth.rotvec_to_matrix(d["x_true"][:3])
    C = synth.VELO_TO_CAM.astype(np.float64)

    def surface_dist(p_cam):
        pv = (p_cam - C[:3, 3]) @ C[:3, :3]                       # camera-0 -> velodyne/world of the previous pose
        return np.min(np.abs(np.stack([pv[:, 2] + synth.SENSOR_HEIGHT, pv[:, 1] - 8, pv[:, 1] + 8, pv[:, 0] - 45, pv[:, 0] + 45])), axis=0)
    moved = d["src_xyz"].astype(np.float64) @ R.T + d["x_true"][3:]
    on_surface = surface_dist(moved) < 2e-3                       # float32 storage of 45 m coordinates
    tgt_frac = (surface_dist(d["tgt_xyz"].astype(np.float64)) < 2e-3).mean()
    assert tgt_frac > 0.5                                          # the rest of the returns are on the seeded boxes
    assert abs(on_surface.mean() - tgt_frac) < 0.02                # moved current scan: as much on the planes as the previous scan
    assert (surface_dist(d["src_xyz"].astype(np.float64)) < 2e-3).mean() < 0.2   # untransformed: mostly off


def test_map_and_stereo_generators():
    m = synth.scan_to_map(n_target=50000, n_beams=16, n_azimuth=256)
    assert m["tgt_xyz"].shape == (50000, 3) and m["tgt_off"][-1] == 50000 and np.all(np.diff(m["tgt_off"]) > 0)
    v = synth.stereo_matches(50, mix="all")
    assert len(v["cam"]) == 100 and set(v["cam"]) == {0, 1}
    inl = np.ones(100, bool)
    # d1-only matches reproject: (R p3_1 + t + t_cam) projects near p2_2
    R = synth.rotvec_to_matrix(synth.velo_pose_to_cam_x(synth.pose_matrix(**synth.TRUE_MOTION))[:3])
    t = synth.velo_pose_to_cam_x(synth.pose_matrix(**synth.TRUE_MOTION))[3:]
    M = v["p3_1"].astype(np.float64) @ R.T + t + v["t_cam"]
    err = np.linalg.norm(M[:, :2] / M[:, 2:3] - v["p2_2"], axis=1)
    assert np.median(err[inl]) < 5e-3


def test_distinct_pairs_are_distinct_and_keep_the_ring_layout():
    """bench.py's step registers B DIFFERENT pairs: own scene, noise, place and motion each; pair 0 is the canonical pair."""
    ps = synth.distinct_pairs(3, n_beams=16, n_azimuth=200)
    canon = synth.scan_pair(n_beams=16, n_azimuth=200)
    assert np.array_equal(ps[0]["src_xyz"], canon["src_xyz"]) and np.array_equal(ps[0]["x0"], canon["x0"])
    for d in ps:
        assert len(d["src_off"]) - 1 == 16 and len(d["tgt_off"]) - 1 == 16          # the segmenter finds every ring again
        assert d["src_xyz"].shape == (3200, 3) and d["src_xyz"].dtype == np.float32
    assert not np.array_equal(ps[1]["tgt_xyz"], ps[2]["tgt_xyz"]) and not np.allclose(ps[1]["x_true"], ps[2]["x_true"])
    for d in ps[1:]:                                                                    # the guess is the previous frame's motion: near the truth, not equal
        e = np.abs(np.asarray(d["x0"]) - np.asarray(d["x_true"]))
        assert 0 < e[3:].max() <= 0.12 and e[:3].max() <= 0.012
    again = synth.distinct_pairs(3, n_beams=16, n_azimuth=200)
    assert all(np.array_equal(a["src_xyz"], b["src_xyz"]) and np.array_equal(a["x0"], b["x0"]) for a, b in zip(ps, again))   # counter-based RNG: reproducible


def test_scan_to_map_with_several_query_scans():
    m = synth.scan_to_map(3 * 16 * 100, n_beams=16, n_azimuth=100, n_queries=3)
    qs = m["queries"]
    assert len(qs) == 3 and np.array_equal(qs[0]["src_xyz"], m["src_xyz"])
    assert all(len(q["src_off"]) - 1 == 16 for q in qs) and not np.array_equal(qs[1]["src_xyz"], qs[2]["src_xyz"])
    assert "queries" not in synth.scan_to_map(3 * 16 * 100, n_beams=16, n_azimuth=100)


def test_drive_plans_are_car_like_and_prefix_consistent():
    """bench.py's default workload (synth.drive_plan): speed 0.6-1.4 m per frame changing by at most 0.1, the car inside the free lane
    (boxes start 1.7 m from the centre line), smooth yaw; drives of up to 42 frames share their first poses with shorter ones (what the
    profiling passes' frame cache relies on); every drive has its own scene and trajectory."""
    for seed in range(8):
        p = synth.drive_plan(40, seed)
        X = np.array(p["x_true"])
        v = np.linalg.norm(X[:, 3:], axis=1)
        assert v.min() > 0.55 and v.max() < 1.45 and np.abs(np.diff(X[:, 5])).max() <= 0.11
        assert np.abs(X[:, 1]).max() < 0.03 and np.abs(np.diff(X[:, 1])).max() < 0.01          # yaw about the camera's y axis
        ys = np.array([T[1, 3] for T in p["poses_velo"]])
        assert np.abs(ys).max() < 1.3
        q = synth.drive_plan(26, seed)
        assert q["scene_kw"] == p["scene_kw"]
        assert all(np.array_equal(a, b) for a, b in zip(q["poses_velo"], p["poses_velo"][:26]))
    a, b = synth.drive_plan(10, 0), synth.drive_plan(10, 1)
    assert not np.allclose(a["x_true"][3], b["x_true"][3])
    long = synth.drive_plan(211, 3)                                        # --steps 200: the road grows with the drive
    assert long["scene_kw"]["half_length"] > 150 and long["scene_kw"]["box_horizon"] == 80.0
    assert max(abs(T[0, 3]) for T in long["poses_velo"]) < long["scene_kw"]["half_length"] - 10


def test_drive_frames_are_ring_scans_and_the_bench_cache_round_trips(tmp_path, monkeypatch):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    plan = synth.drive_plan(3, 5)
    xyz, off = synth.drive_frame(plan, 1, n_beams=16, n_azimuth=128)
    assert xyz.dtype == np.float32 and xyz.shape == (16 * 128, 3) and off[0] == 0 and off[-1] == 16 * 128 and len(off) in (17, 18)
    again = synth.drive_frame(plan, 1, n_beams=16, n_azimuth=128)
    assert np.array_equal(xyz, again[0])                                   # frames are functions of (plan, k): a pool may make them in any order
    monkeypatch.setenv("VELO_DRIVE_CACHE", str(tmp_path))
    small = synth.drive_frame
    monkeypatch.setattr(synth, "drive_frame", lambda pl, k: small(pl, k, 8, 64))
    bench._drives.clear()
    a = bench.make_drives(2, 5, procs=1)
    bench._drives.clear()
    b = bench.make_drives(2, 4, procs=1)                                   # read back: a prefix of the cached drives
    assert (tmp_path / "drives_b2.npz").exists()
    assert all(np.array_equal(a[i]["frames"][k][0], b[i]["frames"][k][0]) and np.array_equal(a[i]["frames"][k][1], b[i]["frames"][k][1])
               for i in range(2) for k in range(4))
    bench._drives.clear()
