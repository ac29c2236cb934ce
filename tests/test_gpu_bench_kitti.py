"""-m gpu: bench.py's configs[0] hook ("run it if a dataset root is supplied by env var").  VELO_KITTI_ROOT names a KITTI
odometry root; here a synthetic drive is written in exactly that layout (sequences/00/velodyne/%06d.bin + calib.txt), and the pairs the
bench would register -- segmented on the device by velo_set_scan_velodyne -- must be the reference segmenter's rings (kitti.h:121-185)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_kitti_root_pairs_are_the_reference_segmenters_rings(hip_lib, tmp_path):
    from velo_amd import synth
    sys.path.insert(0, ROOT)
    import bench
    frames, _ = synth.velodyne_sequence(3, n_beams=16, n_azimuth=200)
    vd = tmp_path / "sequences" / "00" / "velodyne"
    vd.mkdir(parents=True)
    for k, rec in enumerate(frames):
        rec.astype(np.float32).tofile(vd / f"{k:06d}.bin")
    M = synth.VELO_TO_CAM.astype(np.float64)
    (tmp_path / "sequences" / "00" / "calib.txt").write_text("P0: 1 0 0 0 0 1 0 0 0 0 1 0\nTr: " + " ".join(f"{v:.9e}" for v in M[:3].reshape(-1)) + "\n")
    pairs = bench.kitti_pairs(str(tmp_path), 2)
    assert pairs is not None and len(pairs) == 2
    import segmenter_ref
    want = [segmenter_ref.segment_points(f[:, :3], np.vstack([M[:3], [0, 0, 0, 1]])) for f in frames]
    for k, d in enumerate(pairs):                              # pair k registers frame k+1 (source) against frame k (target), main.cpp:388-405
        assert np.array_equal(d["tgt_off"], want[k][1]) and np.array_equal(d["src_off"], want[k + 1][1])
        assert np.array_equal(d["tgt_xyz"].view(np.uint32), want[k][0].view(np.uint32))
        assert np.array_equal(d["src_xyz"].view(np.uint32), want[k + 1][0].view(np.uint32))
    assert bench.kitti_pairs(str(tmp_path / "nothing_here"), 2) is None
    # and through make_workload: the c1 leg's label says which data it ran on
    os.environ["VELO_KITTI_ROOT"] = str(tmp_path)
    try:
        bench._cache.clear()
        W = bench.make_workload("c1", 2)
        assert "KITTI" in W["label"] and W["icp_skip"] == 200 and W["distinct"] == 2
    finally:
        os.environ.pop("VELO_KITTI_ROOT", None)
        bench._cache.clear()
