"""-m gpu: the rows SURVEY.md 8(f) marks "next", each to the same bar as the path:
  row 1  device-side scan ingestion  (kitti.h:121-185)  -- bit-exact against the restatements of the ring segmenter in tests/segmenter_ref.py;
  row 2  pose hand-off / odometry loop (main.cpp:305-331,407-437, kitti.h:202-216) -- same chain as the oracle driven on the CPU;
  row 3  projectLidarToCamera + featureDepthAssociation (velo.h:329-497) -- bit-exact (float) against the oracle's restatement;
  row 4  batched triangulatePoint (velo.h:1027-1130) -- float32 points within 1e-4 (relative to max(1, |p|)) of the oracle's, solver
         summaries equal; in practice bit-identical because the device follows the same operation order in double."""
import numpy as np
import pytest

import helpers as H
import oracle_lib as O
import segmenter_ref                     # the CHECKER of the device segmenter lives under tests/ (pinned by tests/test_segmenter_ref.py)
from velo_amd import api, odometry, synth

pytestmark = pytest.mark.gpu


def records_of(pts):
    rec = np.zeros((len(pts), 4), dtype=np.float32)
    rec[:, :3] = pts
    rec[:, 3] = 0.5
    return rec


@pytest.mark.parametrize("shape", [(64, 1875), (16, 200)])
def test_device_segmenter_matches_numpy_bit_exact(hip_lib, shape):
    scene = synth.Scene(0)
    pts = synth.hdl64_scan(scene, synth.pose_matrix(0.01, 0, 0, (1.0, 0.2, 0)), noise_seed=5, n_beams=shape[0], n_azimuth=shape[1])
    if shape[0] == 16:                          # ragged rings: drop 10 % of the returns (file order preserved)
        keep = synth.uniform01(9, len(pts)) > 0.1
        pts = pts[keep]
    want_xyz, want_off = segmenter_ref.segment_points(pts, synth.VELO_TO_CAM)
    if shape[0] == 16:                          # small enough for the literal scalar transcription of kitti.h:158-183 as well
        sc_xyz, sc_off = segmenter_ref.segment_points_scalar(pts, synth.VELO_TO_CAM)
        assert np.array_equal(sc_off, want_off) and np.array_equal(sc_xyz.view(np.uint32), want_xyz.view(np.uint32))
    c = api.Context(0)
    for as_target in (True, False):
        c.set_scan_velodyne(as_target, records_of(pts), synth.VELO_TO_CAM)
        assert np.array_equal(c.ring_offsets(as_target), want_off)
        assert np.array_equal(c.cloud(as_target).view(np.uint32), want_xyz.view(np.uint32))
    c.close()


def test_device_segmenter_feeds_the_path_like_host_rings(hip_lib):
    scene = synth.Scene(0)
    a = synth.hdl64_scan(scene, synth.pose_matrix(0, 0, 0, (0, 0, 0)), noise_seed=1, n_beams=32, n_azimuth=400)
    b = synth.hdl64_scan(scene, synth.pose_matrix(**synth.TRUE_MOTION), noise_seed=2, n_beams=32, n_azimuth=400)
    c1, c2 = api.Context(0, icp_skip=1), api.Context(0, icp_skip=1)
    c1.set_scan_velodyne(True, records_of(a), synth.VELO_TO_CAM)
    c1.set_scan_velodyne(False, records_of(b), synth.VELO_TO_CAM)
    ta, oa = segmenter_ref.segment_points(a, synth.VELO_TO_CAM)
    tb, ob = segmenter_ref.segment_points(b, synth.VELO_TO_CAM)
    c2.set_target(ta, oa)
    c2.set_source(tb, ob)
    x1, _, _ = c1.frame_to_frame(synth.INITIAL_GUESS)
    x2, _, _ = c2.frame_to_frame(synth.INITIAL_GUESS)
    assert np.array_equal(x1, x2)
    c1.close()
    c2.close()


def test_source_promoted_to_target_equals_a_fresh_upload(hip_lib):
    """The scan cache of the loop: frame k, segmented on the device as source, becomes the target of the next registration
    without another upload (velo_source_to_target) -- same cloud, same ring table, same index, same result bit for bit."""
    scene = synth.Scene(0)
    f0 = records_of(synth.hdl64_scan(scene, synth.pose_matrix(0, 0, 0, (0, 0, 0)), noise_seed=1, n_beams=32, n_azimuth=400))
    f1 = records_of(synth.hdl64_scan(scene, synth.pose_matrix(**synth.TRUE_MOTION), noise_seed=2, n_beams=32, n_azimuth=400))
    a, b = api.Context(0, icp_skip=1), api.Context(0, icp_skip=1)
    with pytest.raises(api.VeloError):
        a.source_to_target()                                  # nothing to promote yet
    a.set_scan_velodyne(False, f0, synth.VELO_TO_CAM)
    a.source_to_target()
    with pytest.raises(api.VeloError):
        a.frame_to_frame(synth.INITIAL_GUESS)                 # the source slot is empty now
    a.set_scan_velodyne(False, f1, synth.VELO_TO_CAM)
    b.set_scan_velodyne(True, f0, synth.VELO_TO_CAM)
    b.set_scan_velodyne(False, f1, synth.VELO_TO_CAM)
    assert np.array_equal(a.ring_offsets(True), b.ring_offsets(True))
    assert np.array_equal(a.cloud(True).view(np.uint32), b.cloud(True).view(np.uint32))
    xa, Ta, sa = a.frame_to_frame(synth.INITIAL_GUESS)
    xb, Tb, sb = b.frame_to_frame(synth.INITIAL_GUESS)
    assert np.array_equal(xa, xb) and np.array_equal(Ta, Tb)
    a.close()
    b.close()


def test_odometry_loop_matches_cpu_chain_and_truth(hip_lib, oracle, tmp_path):
    frames, truth = synth.velodyne_sequence(5, n_beams=32, n_azimuth=400)
    odo = odometry.LidarOdometer(0, icp_skip=1)
    for rec in frames:
        odo.push(rec)
    # the same hand-off logic with the CPU oracle as frameToFrame
    poses = [np.eye(4)]
    segs = [synth.segment_points(r[:, :3]) for r in frames]
    for k in range(1, len(frames)):
        if k > 1:
            dT = np.linalg.inv(poses[k - 2]) @ poses[k - 1]
            x0 = oracle.pose_mat_to_vec(dT)
        else:
            x0 = odometry.FIRST_GUESS
        orc = oracle.Oracle(threads=8, icp_skip=1)
        orc.set_target(*segs[k - 1])
        orc.set_source(*segs[k])
        x, T, _ = orc.frame_to_frame(x0)
        poses.append(poses[k - 1] @ T)
    for k in range(len(frames)):
        assert H.pose_close(api.pose_mat_to_vec(odo.poses[k]), oracle.pose_mat_to_vec(poses[k]), 1e-4 * max(k, 1), 1e-5 * max(k, 1))
        # drift against the simulated trajectory stays at noise level over the short drive
        assert np.linalg.norm(odo.poses[k][:3, 3] - truth[k][:3, 3]) < 0.02 * max(k, 1)
    assert all(np.linalg.norm(a[3:]) < odometry.AGREEMENT_T_THRESH and np.linalg.norm(a[:3]) < odometry.AGREEMENT_R_THRESH
               for a in odo.agreements[1:])
    out = tmp_path / "00.txt"
    odo.write_kitti(str(out))
    lines = out.read_text().splitlines()
    assert len(lines) == 5 and lines[0].split() == ["1", "0", "0", "0", "0", "1", "0", "0", "0", "0", "1", "0"]
    assert all(len(l.split()) == 12 for l in lines)
    odo.close()


# ---- row 3: camera projection of the rings + keypoint depth -----------------------------------------------------------------
def _check_projection(c, xyz, off, cam_t, window, of_target):
    n = c.project_lidar(of_target, cam_t, window)
    got = c.projection()
    want = O.project_lidar(xyz, off, cam_t, window)
    assert n == len(want[0])
    assert np.array_equal(got[2], want[2])
    assert np.array_equal(got[0].view(np.uint32), want[0].view(np.uint32))
    assert np.array_equal(got[1].view(np.uint32), want[1].view(np.uint32))
    return want


def _check_depth(c, want_proj, kps, thresh):
    kd, has = c.depth_association(kps, thresh)
    wkd, whas = O.depth_association(*want_proj, kps, thresh)
    assert np.array_equal(has, whas)
    assert np.array_equal(kd.view(np.uint32), wkd.view(np.uint32))
    return int((has >= 0).sum())


@pytest.mark.parametrize("cam", [0, 1])
def test_projection_and_depth_match_oracle_on_the_street_scan(hip_lib, cam):
    d = synth.scan_pair()                                   # 64 x 1875
    w = synth.cam_window()
    c = api.Context(0)
    c.set_target(d["tgt_xyz"], d["tgt_off"])
    c.set_source(d["src_xyz"], d["src_off"])
    for of_target, xyz, off in ((True, d["tgt_xyz"], d["tgt_off"]), (False, d["src_xyz"], d["src_off"])):
        want = _check_projection(c, xyz, off, synth.CAM_TRANS[cam], w, of_target)
        assert _check_depth(c, want, synth.keypoints_in_window(5000, seed=31 + cam), synth.DEPTH_ASSOC_THRESH) > 1000
    c.close()


def test_projection_stack_rules_on_crafted_rings(hip_lib):
    from test_depth_oracle import crafted_rings
    xyz, off = crafted_rings(n_rings=150)                    # > 64 rings: the keypoint search carries state across lane chunks
    w = synth.cam_window()
    c = api.Context(0)
    c.set_source(xyz, off)
    for cam in (0, 1):
        want = _check_projection(c, xyz, off, synth.CAM_TRANS[cam], w, False)
        kps = synth.keypoints_in_window(700, seed=3)
        kps[5] = (np.nan, 0.0)
        kps[6] = (0.0, np.nan)
        assert _check_depth(c, want, kps, 0.2) > 50
        _check_depth(c, want, kps, synth.DEPTH_ASSOC_THRESH)
        _check_depth(c, want, kps[:1], 0.2)
    c.close()


def test_depth_golden_fixture_and_call_order(hip_lib):
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "depth_mini.npz"))
    c = api.Context(0)
    with pytest.raises(api.VeloError):
        c.depth_association(g["keypoints"])                  # nothing projected yet
    with pytest.raises(api.VeloError):
        c.project_lidar(True, g["cam_t"], g["window"])       # no target loaded
    c.set_target(g["xyz"], g["off"])
    assert c.project_lidar(True, g["cam_t"], g["window"]) == len(g["proj_xy"])
    xy, pts, off = c.projection()
    assert np.array_equal(off, g["proj_off"])
    assert np.array_equal(xy.view(np.uint32), g["proj_xy"].view(np.uint32))
    assert np.array_equal(pts.view(np.uint32), g["proj_pts"].view(np.uint32))
    kd, has = c.depth_association(g["keypoints"], float(g["thresh"]))
    assert np.array_equal(has, g["has_depth"])
    assert np.array_equal(kd.view(np.uint32), g["kp_with_depth"].view(np.uint32))
    kd0, has0 = c.depth_association(np.zeros((0, 2), np.float32))
    assert len(kd0) == 0 and len(has0) == 0
    # a window nothing falls into: empty lists, no keypoint gets depth
    assert c.project_lidar(True, g["cam_t"], [5.0, 6.0, 5.0, 6.0]) == 0
    kd1, has1 = c.depth_association(g["keypoints"])
    assert len(kd1) == 0 and np.all(has1 == -1)
    c.close()


def test_depth_feeds_the_visual_rows(hip_lib, oracle):
    """Depth from the device feeds rows R2/R4 like the reference's pipeline does: 3-D keypoints of both frames from the two
    scans, matched by construction (same keypoints seen from both poses), go through frame_to_frame on both sides."""
    d = synth.scan_pair(n_beams=64, n_azimuth=1875)
    w = synth.cam_window()
    c = api.Context(0, icp_skip=50)
    o = O.Oracle(icp_skip=50)
    H.load_both(c, o, d)
    c.project_lidar(True, synth.CAM_TRANS[0], w)
    kps = synth.keypoints_in_window(800, seed=77)
    kd, has = c.depth_association(kps)
    sel = np.nonzero(has >= 0)[0]
    assert len(sel) > 200
    # current-frame 3-D point = the previous-frame one moved by the true motion (x_true maps frame1 -> frame2, so invert it)
    R = synth.rotvec_to_matrix(d["x_true"][:3])
    P2 = kd[has[sel]].astype(np.float64)
    P1 = (P2 - d["x_true"][3:]) @ R
    q1 = P1[:, :2] / P1[:, 2:3]
    n = len(sel)
    rec = dict(cam=np.zeros(n, np.int32), point1=np.arange(n, dtype=np.int32), point2=np.arange(n, dtype=np.int32),
               d1=np.ones(n, np.uint8), d2=np.ones(n, np.uint8), p3_1=P1.astype(np.float32), p3_2=P2.astype(np.float32),
               p2_1=q1.astype(np.float32), p2_2=kps[sel], t_cam=np.zeros((n, 3), np.float32))
    vis = api.matches_from_dict(rec)
    c.set_visual(vis)
    o.set_visual(vis)
    xg, _, sg = c.frame_to_frame(d["x0"])
    xo, _, so = o.frame_to_frame(d["x0"])
    assert H.pose_close(xg, xo)
    assert sg.solves[0].n_visual_blocks == so.solves[0].n_visual_blocks > 0
    c.close()


# ---- row 4: batched landmark triangulation ----------------------------------------------------------------------------------
TRI_TOL = 1e-4          # |dp| <= TRI_TOL * max(1, |p|): float32 points out of a double LM, same bar as the pose (1e-4 m)


def _tri_both(c, pr, **params):
    args = (pr["camera_poses"], pr["cam_trans"], pr["obs"], pr["obs_offsets"], pr["points0"], pr["initial_guess"])
    P = O.default_params()
    if params:
        c.set_params(**params)
        for k, v in params.items():
            setattr(P, k, v)
    got = c.triangulate_points(*args)
    want = O.triangulate_points(*args, params=P)
    return got, want


def _assert_tri_close(got, want, min_identical=0.99):
    (gp, gr), (wp, wr) = got, want
    scale = np.maximum(1.0, np.linalg.norm(wp, axis=1))
    # landmarks the solver gave up on (50 iterations / failure) sit on flat valleys: only converged ones are held to the tolerance
    conv = wr["termination"] == 0
    d = np.linalg.norm(gp.astype(np.float64) - wp, axis=1) / scale
    assert np.all(d[conv] <= TRI_TOL), (np.nonzero(d > TRI_TOL)[0][:10], d.max())
    same = np.all(gp.view(np.uint32) == wp.view(np.uint32), axis=1)
    assert same.mean() >= min_identical, same.mean()
    assert np.array_equal(gr["n_solves"], wr["n_solves"])
    agree = (gr["termination"] == wr["termination"]) & (gr["lm_iterations"] == wr["lm_iterations"]) & (gr["evaluations"] == wr["evaluations"])
    assert agree.mean() >= min_identical, agree.mean()
    np.testing.assert_allclose(gr["final_cost"][agree], wr["final_cost"][agree], rtol=1e-9, atol=1e-14)


def test_triangulation_matches_oracle(hip_lib):
    pr = synth.triangulation_problem(3000, n_frames=12, seed=13)
    c = api.Context(0)
    got, want = _tri_both(c, pr)
    _assert_tri_close(got, want)
    n_obs = np.diff(pr["obs_offsets"])
    err = np.linalg.norm(got[0] - pr["truth"], axis=1)[n_obs >= 6]
    assert np.median(err) < 0.1                               # and the points are the landmarks, not just equal to the oracle's
    # deterministic: the same call again gives the same bits
    again = c.triangulate_points(pr["camera_poses"], pr["cam_trans"], pr["obs"], pr["obs_offsets"], pr["points0"], pr["initial_guess"])
    assert np.array_equal(again[0].view(np.uint32), got[0].view(np.uint32))
    c.close()


def test_triangulation_kernels_agree_bit_for_bit(hip_lib, diag_lib, monkeypatch):
    # one wave per landmark (default) and one thread per landmark sum every accumulator in the same order
    pr = synth.triangulation_problem(1500, n_frames=40, seed=5)       # up to ~100 observations: more than one 64-lane chunk
    assert np.diff(pr["obs_offsets"]).max() > 64
    args = (pr["camera_poses"], pr["cam_trans"], pr["obs"], pr["obs_offsets"], pr["points0"], pr["initial_guess"])
    out = []
    for variant in ("1", "0"):
        monkeypatch.setenv("VELO_TRI_VARIANT", variant)        # "0" exists in the diagnostics build only
        c = api.Context(0, lib=diag_lib if variant == "0" else None)
        out.append(c.triangulate_points(*args))
        c.close()
    assert np.array_equal(out[0][0].view(np.uint32), out[1][0].view(np.uint32))
    assert np.array_equal(out[0][1], out[1][1])
    _assert_tri_close(out[0], O.triangulate_points(*args))


def test_triangulation_edge_cases(hip_lib):
    c = api.Context(0)
    # golden fixture
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "triangulation_mini.npz"))
    pts, res = c.triangulate_points(g["camera_poses"], g["cam_trans"], g["obs"], g["obs_offsets"], g["points0"], g["initial_guess"])
    scale = np.maximum(1.0, np.linalg.norm(g["points"], axis=1))
    assert np.all(np.linalg.norm(pts - g["points"], axis=1) / scale <= TRI_TOL)
    assert np.array_equal(res["n_solves"], g["results"]["n_solves"])
    # no initial-guess array at all == all zeros; other solver settings travel through velo_params
    pr = synth.triangulation_problem(200, n_frames=6, seed=3)
    pr0 = dict(pr, initial_guess=None)
    pr1 = dict(pr, initial_guess=np.zeros(200, np.uint8))
    a = c.triangulate_points(pr0["camera_poses"], pr0["cam_trans"], pr0["obs"], pr0["obs_offsets"], pr0["points0"], None)
    b = c.triangulate_points(pr1["camera_poses"], pr1["cam_trans"], pr1["obs"], pr1["obs_offsets"], pr1["points0"], pr1["initial_guess"])
    assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))
    got, want = _tri_both(c, pr, max_num_iterations=4, loss_thresh_3D2D=0.02, weight_3D2D=3.0)
    _assert_tri_close(got, want, min_identical=0.97)
    assert got[1]["lm_iterations"].max() <= 4
    # nothing to do / bad input
    e = c.triangulate_points(pr["camera_poses"], pr["cam_trans"], pr["obs"][:0], np.zeros(1, np.int32), np.zeros((0, 3), np.float32))
    assert len(e[0]) == 0
    bad = pr["obs"].copy()
    bad["frame"][0] = 99
    with pytest.raises(api.VeloError):
        c.triangulate_points(pr["camera_poses"], pr["cam_trans"], bad, pr["obs_offsets"], pr["points0"], pr["initial_guess"])
    bad = pr["obs"].copy()
    bad["kind"][1] = 7
    with pytest.raises(api.VeloError):
        c.triangulate_points(pr["camera_poses"], pr["cam_trans"], bad, pr["obs_offsets"], pr["points0"], pr["initial_guess"])
    c.close()


@pytest.mark.gpu
def test_device_memory_returns_after_destroy():
    """Create / upload / solve / promote / destroy cycles leave device memory where it was (every buffer a context owns,
    including the grid of each replaced target and the warm-start table, goes with the context)."""
    import torch
    d = synth.scan_pair(n_beams=16, n_azimuth=400)

    def cycle():
        ctxs = [api.Context(0, icp_skip=1) for _ in range(2)]
        for c in ctxs:
            c.set_target(d["tgt_xyz"], d["tgt_off"]); c.set_source(d["src_xyz"], d["src_off"])
        api.frame_to_frame_batch(ctxs, [d["x0"]] * 2)
        for _ in range(4):                      # a new target per frame must reuse / free the old grid
            ctxs[0].source_to_target(); ctxs[0].set_source(d["src_xyz"], d["src_off"])
        ctxs[0].frame_to_frame(d["x0"])
        cache = api.ScanCache(0, capacity=2)        # three stores into two nodes: one eviction, one recycled node
        for f in range(3):
            cache.store(f, ctxs[0], True)
        cache.load(2, ctxs[1], True)
        cache.close()
        # round 5: a frame announced ahead (velo_hint_next_source: two landing buffers, a copy stream, an event) and the drive loop in one call
        host = (np.ascontiguousarray(d["src_xyz"]), d["src_off"])
        refs = api.scan_refs([host], 0)
        api.hint_next_sources(ctxs[:1], refs[0])
        api.register_batch(ctxs[:1], None, None, np.asarray(d["x0"])[None, :], refs=(api.promote_refs(1), refs))
        seq_refs, _keep, _n = api.sequence_refs([[host, host, host]], 0, first=1)
        api.register_sequences(ctxs[:1], seq_refs, 2, np.tile(np.eye(4), (1, 1, 1)), np.asarray(d["x0"], dtype=np.float64)[None, :].copy())
        for _ in range(4):                      # every export hands out a new slab; the retired ones are capped
            ctxs[1].comm_peer_export()
        for c in ctxs: c.close()

    for _ in range(3): cycle()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(0)[0]
    for _ in range(25): cycle()
    torch.cuda.synchronize()
    drift = free0 - torch.cuda.mem_get_info(0)[0]
    assert drift < 16 << 20, f"device memory drift {drift / 1e6:.1f} MB over 25 cycles"


@pytest.mark.gpu
def test_scan_cache_lru_order_and_eviction(hip_lib):
    """lru.h:31-61 on the device: most-recently-used first, a hit moves to the front, the oldest entry goes beyond `capacity`,
    a stored frame replaces its older copy, a miss is an error the caller answers by reading the scan."""
    d = synth.scan_pair(n_beams=8, n_azimuth=200)
    c = api.Context(0, icp_skip=1)
    c.set_target(d["tgt_xyz"], d["tgt_off"])
    cache = api.ScanCache(0, capacity=3)
    model = []                                   # python model of the std::list, front = most recent

    def store(f):
        cache.store(f, c, True)
        if f in model: model.remove(f)
        model.insert(0, f)
        del model[3:]

    def load(f):
        cache.load(f, c, True)
        model.remove(f); model.insert(0, f)

    for f in (10, 11, 12):
        store(f)
    assert cache.frames() == model == [12, 11, 10]
    load(10)
    assert cache.frames() == model == [10, 12, 11]
    store(13)                                    # evicts 11, the least recently used
    assert cache.frames() == model == [13, 10, 12] and 11 not in cache and 10 in cache
    store(12)                                    # replaced and moved to the front, not duplicated
    assert cache.frames() == model == [12, 13, 10]
    with pytest.raises(api.VeloError):
        cache.load(11, c, True)
    c2 = api.Context(0)
    with pytest.raises(api.VeloError):
        cache.store(1, c2, False)                # the context holds no source scan
    c2.close(); cache.close(); c.close()


@pytest.mark.gpu
def test_scan_cache_serves_targets_and_sources_like_fresh_uploads(hip_lib):
    """A registration against a cached scan equals the registration against a fresh upload bit for bit: entries stored from a
    target (cloud + index reused), from a source (index built on load), loaded into several contexts, after an eviction cycle,
    and under different gates (index rebuilt); the dframe pattern of main.cpp:306-350 (frame k against k-1 and k-2)."""
    recs, _ = synth.velodyne_sequence(3, n_beams=16, n_azimuth=300)
    (f0, o0), (f1, o1), (f2, o2) = [synth.segment_points(r[:, :3]) for r in recs]
    x0 = np.array([0.0, 0.0, 0.0, 0.0, 0.0, 0.8])

    def table(c):                                # the full record (indices, distances) of an explicit round at x0
        c.associate(x0, 1)
        return c.correspondences().tobytes()

    def fresh(tgt, toff, src, soff, **params):
        c = api.Context(0, icp_skip=1, **params)
        c.set_target(tgt, toff); c.set_source(src, soff)
        tab = table(c)
        x, T, S = c.frame_to_frame(x0)
        c.close()
        return x, tab, [S.solves[k].lm_iterations for k in range(S.n_solves)]

    cache = api.ScanCache(0, capacity=4)
    a = api.Context(0, icp_skip=1)
    a.set_target(f0, o0); a.set_source(f1, o1)
    cache.store(0, a, True)                      # frame 0 with its index
    cache.store(1, a, False)                     # frame 1 from the source side: cloud only
    # frame 2 arrives: register it against frame 1 and frame 0 (dframe = 1, 2), both served by the cache
    b = api.Context(0, icp_skip=1)
    b.set_source(f2, o2)
    for frame, tgt, toff in ((1, f1, o1), (0, f0, o0)):
        cache.load(frame, b, True)
        tab = table(b)
        x, T, S = b.frame_to_frame(x0)
        xr, tabr, itr = fresh(tgt, toff, f2, o2)
        assert np.array_equal(x, xr) and tab == tabr
        assert [S.solves[k].lm_iterations for k in range(S.n_solves)] == itr
    # the same entry in two contexts at once, one of them taking its SOURCE from the cache too
    c1, c2 = api.Context(0, icp_skip=1), api.Context(0, icp_skip=1)
    cache.load(0, c1, True); cache.load(0, c2, True)
    c1.set_source(f1, o1); cache.load(1, c2, False)
    t1, t2 = table(c1), table(c2)
    xs, Ts, Ss = api.frame_to_frame_batch([c1, c2], [x0, x0])
    xr, tabr, _ = fresh(f0, o0, f1, o1)
    assert np.array_equal(xs[0], xr) and np.array_equal(xs[1], xr)
    assert t1 == tabr == t2
    # other gates than the entry's index was built for: rebuilt from the cached cloud
    g = api.Context(0, icp_skip=1, correspondence_thresh_icp=0.3)
    cache.load(0, g, True); g.set_source(f1, o1)
    tg = table(g)
    xg, _, _ = g.frame_to_frame(x0)
    xr, tabr, _ = fresh(f0, o0, f1, o1, correspondence_thresh_icp=0.3)
    assert np.array_equal(xg, xr) and tg == tabr
    # survive an eviction cycle: fill the cache, frame 0 was used last -> stays; frame 1 goes
    for f in (5, 6, 7):
        cache.store(f, a, True)
    assert 0 in cache and 1 not in cache
    cache.load(0, b, True)
    x, _, _ = b.frame_to_frame(x0)
    xr, _, _ = fresh(f0, o0, f2, o2)
    assert np.array_equal(x, xr)
    for c in (a, b, c1, c2, g):
        c.close()
    cache.close()


@pytest.mark.gpu
def test_odometry_with_diagonal_edges_from_the_scan_cache(hip_lib):
    """ndiagonal = 3 (main.cpp:148-152,306-350): every frame is also registered against k-2 and k-3, served by the device scan
    cache.  The chain itself is unchanged, every extra edge equals a registration against a fresh upload of that frame, and agrees
    with the chained poses to a few centimetres."""
    frames, truth = synth.velodyne_sequence(5, n_beams=32, n_azimuth=400)
    plain = odometry.LidarOdometer(0, icp_skip=1)
    diag = odometry.LidarOdometer(0, icp_skip=1, ndiagonal=3, cache_capacity=3)
    for rec in frames:
        plain.push(rec); diag.push(rec)
    assert all(np.array_equal(a, b) for a, b in zip(plain.poses, diag.poses))
    assert [(a, b) for a, b, _ in diag.edges] == [(0, 2), (1, 3), (0, 3), (2, 4), (1, 4)]
    clouds = [synth.segment_points(r[:, :3]) for r in frames]
    for a, b, dpose in diag.edges:
        c = api.Context(0, icp_skip=1)
        c.set_target(*clouds[a]); c.set_source(*clouds[b])
        dT = np.linalg.inv(diag.poses[a]) @ diag.poses[b]
        _, want, _ = c.frame_to_frame(api.pose_mat_to_vec(dT))
        c.close()
        assert np.array_equal(dpose, want)
        assert np.linalg.norm((dpose @ np.linalg.inv(dT))[:3, 3]) < 0.05
    assert diag.rejected_edges == []
    plain.close(); diag.close()

    # main.cpp:426-437: an edge over dframe > 1 frames whose registration disagrees with the chained poses is skipped (`continue`) and
    # never reaches the graph.  Inject two bad registrations: 0.5 m off over 2 frames (limit 0.2 m), 0.08 rad off over 3 (limit 0.05).
    class Faulty(odometry.LidarOdometer):
        def _register_edge(self, a, b, x0):
            x, dpose, s = super()._register_edge(a, b, x0)
            if (a, b) == (1, 3):
                dpose = dpose.copy(); dpose[0, 3] += 0.5
            if (a, b) == (1, 4):
                R = np.eye(4); R[:3, :3] = synth.rotvec_to_matrix([0.0, 0.08, 0.0])
                dpose = R @ dpose
            return x, dpose, s

    bad = Faulty(0, icp_skip=1, ndiagonal=3, cache_capacity=3)
    for rec in frames:
        bad.push(rec)
    assert all(np.array_equal(a, b) for a, b in zip(plain.poses, bad.poses))            # the chain itself never skips (dframe == 1)
    assert [(a, b) for a, b, _ in bad.edges] == [(0, 2), (0, 3), (2, 4)]
    assert [(a, b, why) for a, b, _, _, why in bad.rejected_edges] == [(1, 3, "poor t agreement"), (1, 4, "poor r agreement")]
    assert abs(np.linalg.norm(bad.rejected_edges[0][3][3:]) - 0.5) < 0.05 and abs(np.linalg.norm(bad.rejected_edges[1][3][:3]) - 0.08) < 0.01
    bad.close()


@pytest.mark.gpu
def test_odometry_over_a_directory_of_kitti_sweeps(hip_lib, tmp_path):
    """KITTI layout in (velodyne/*.bin + calib.txt Tr row), KITTI pose file out: same poses as pushing the records directly."""
    frames, _ = synth.velodyne_sequence(4, n_beams=16, n_azimuth=300)
    vdir = tmp_path / "velodyne"
    vdir.mkdir()
    for k, rec in enumerate(frames):
        synth.write_kitti_bin(str(vdir / f"{k:06d}.bin"), rec[:, :3])
    calib = tmp_path / "calib.txt"
    calib.write_text("P0: 1 0 0 0 0 1 0 0 0 0 1 0\nTr: " + " ".join(f"{v:.9e}" for v in synth.VELO_TO_CAM[:3, :4].reshape(-1)) + "\n")
    assert np.array_equal(odometry.read_velo_to_cam(str(calib))[:3], synth.VELO_TO_CAM[:3].astype(np.float32))
    odo = odometry.run_directory(str(vdir), str(tmp_path / "poses.txt"), str(calib), icp_skip=1)
    ref = odometry.LidarOdometer(0, icp_skip=1)
    for rec in frames:
        ref.push(rec)
    assert all(np.array_equal(a, b) for a, b in zip(odo.poses, ref.poses))
    lines = (tmp_path / "poses.txt").read_text().splitlines()
    assert len(lines) == 4 and lines[1] == odometry.kitti_pose_line(ref.poses[1]).rstrip("\n")
    odo.close(); ref.close()


@pytest.mark.gpu
def test_randomized_next_rows_sweep(hip_lib):
    """Eight seeds of tools/fuzz_next_rows.py: projection, keypoint depth and triangulation on random sizes, bit for bit vs the oracle."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("fuzz_next_rows", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_next_rows.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    assert m.run(8, first_seed=900) == 48


def test_bench_drive_step_promotes_on_the_device_and_matches_the_oracle_pair_by_pair(hip_lib, oracle):
    """bench.py's default step (DriveWalker): B drives advance one frame through ONE velo_register_batch call -- targets flagged
    VELO_SCAN_PROMOTE (the previous frame, held as source, becomes the target on the device), sources = the new frames, guesses from the
    constant-velocity hand-off (main.cpp:311-331,408).  Every registration must be the oracle's on the same two frames and the same guess,
    and the same as the explicit calls velo_source_to_target + velo_set_source + velo_frame_to_frame."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    B, n_frames = 4, 5
    drives = [synth.drive(n_frames, seed=s, n_beams=16, n_azimuth=128) for s in range(B)]
    ctxs = [api.Context(0, icp_skip=1) for _ in range(B)]
    w = bench.DriveWalker(api, ctxs, [d["frames"] for d in drives], 0)
    solo = api.Context(0, icp_skip=1)
    solo.set_source(*drives[1]["frames"][0])
    P = np.tile(np.eye(4), (B, 1, 1))
    for k in range(1, n_frames):
        x0 = w.x0.copy()
        if k == 1:
            assert np.array_equal(x0, np.tile(synth.INITIAL_GUESS, (B, 1)))
        xs, Ts, Ss = w.step()
        for i in range(B):
            orc = oracle.Oracle(threads=4, icp_skip=1)
            orc.set_target(*drives[i]["frames"][k - 1])
            orc.set_source(*drives[i]["frames"][k])
            xo, To, so = orc.frame_to_frame(x0[i])
            assert H.pose_close(xs[i], xo, 1e-9, 1e-10), (k, i, xs[i], xo)
            assert [Ss[i].solves[j].evaluations for j in range(6)] == [so.solves[j].evaluations for j in range(6)]
            # the hand-off itself against hand-written matrix arithmetic: T[k] = T[k-1] dpose; next guess = vec(T[k-1]^-1 T[k])
            P_new = P[i] @ To
            assert np.allclose(w.P_prev[i], P_new, atol=1e-9)
            assert H.pose_close(w.x0[i], api.pose_mat_to_vec(np.linalg.inv(P[i]) @ P_new), 1e-9, 1e-10)
            assert H.pose_close(w.x0[i], xs[i], 1e-9, 1e-9)              # ... which is the pair's own motion: constant velocity
            P[i] = P_new
        # drive 1 once more, with the explicit calls
        solo.source_to_target()
        solo.set_source(*drives[1]["frames"][k])
        x1, _T1, s1 = solo.frame_to_frame(x0[1])
        assert np.array_equal(x1, xs[1]) and [s1.solves[j].evaluations for j in range(6)] == [Ss[1].solves[j].evaluations for j in range(6)]
        assert np.array_equal(solo.ring_offsets(True), drives[1]["frames"][k - 1][1])
        assert np.array_equal(solo.cloud(True).view(np.uint32), np.asarray(drives[1]["frames"][k - 1][0]).view(np.uint32))
    # the estimated motion follows the simulated one at the noise level of these 2,048-point sweeps
    for i in range(B):
        assert np.linalg.norm(xs[i][3:] - drives[i]["x_true"][-1][3:]) < 0.15 and np.linalg.norm(xs[i][:3] - drives[i]["x_true"][-1][:3]) < 3e-2
    for c in ctxs + [solo]:
        c.close()
