"""-m gpu: the rows SURVEY.md 8(f) marks "next", each to the same bar as the path:
  row 1  device-side scan ingestion  (kitti.h:121-185)  -- bit-exact against the numpy restatement of the ring segmenter;
  row 2  pose hand-off / odometry loop (main.cpp:305-331,407-437, kitti.h:202-216) -- same chain as the oracle driven on the CPU."""
import numpy as np
import pytest

import helpers as H
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
    want_xyz, want_off = synth.segment_points(pts)
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
    ta, oa = synth.segment_points(a)
    tb, ob = synth.segment_points(b)
    c2.set_target(ta, oa)
    c2.set_source(tb, ob)
    x1, _, _ = c1.frame_to_frame(synth.INITIAL_GUESS)
    x2, _, _ = c2.frame_to_frame(synth.INITIAL_GUESS)
    assert np.array_equal(x1, x2)
    c1.close()
    c2.close()


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
