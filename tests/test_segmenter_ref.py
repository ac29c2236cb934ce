"""CPU: the checker of the device ring segmenter is pinned to a literal scalar transcription of kitti.h:158-183, and so is the
generator's copy in the package (synth.segment_points) -- three independent writings, bit-identical."""
import numpy as np
import pytest

import segmenter_ref as R
import velo_amd  # noqa: F401
from velo_amd import synth


def _same(a, b):
    return np.array_equal(a[1], b[1]) and np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))


@pytest.mark.parametrize("shape", [(16, 200), (64, 1875)])
def test_three_writings_agree_on_a_synthetic_sweep(shape):
    d = synth.velodyne_sequence(1, n_beams=shape[0], n_azimuth=shape[1])[0][0][:, :3]
    M = synth.VELO_TO_CAM.astype(np.float32)
    want = R.segment_points_scalar(d, M)
    assert len(want[1]) - 1 == shape[0] + 1 or len(want[1]) - 1 == shape[0]        # 64 beams -> 64 or 65 rings (the first break)
    assert _same(R.segment_points(d, M), want)
    assert _same(synth.segment_points(d, M), want)


def test_ragged_and_degenerate_inputs():
    rng = np.random.default_rng(5)
    M = np.eye(4, dtype=np.float32)
    M[:3, :] = rng.normal(size=(3, 4)).astype(np.float32)
    # random points: ring breaks wherever x > 0 and the sign of y flips -> rings of 1, 2, 3 ... points, odd and even lengths
    for n in (0, 1, 2, 7, 500):
        p = rng.normal(size=(n, 3)).astype(np.float32)
        want = R.segment_points_scalar(p, M)
        assert _same(R.segment_points(p, M), want), n
        if n:
            assert _same(synth.segment_points(p, M), want), n
    # y == 0 counts as "not positive"; x == 0 never breaks (strict comparisons, kitti.h:166)
    p = np.array([[1, 1, 0], [1, 0, 0], [1, 1, 0], [0, -1, 0], [1, -1, 0], [1, 1, 0]], dtype=np.float32)
    want = R.segment_points_scalar(p, np.eye(4, dtype=np.float32))
    assert list(want[1]) == [0, 1, 2, 5, 6]
    assert _same(R.segment_points(p, np.eye(4, dtype=np.float32)), want)
    # the reorder: new i <- old (n - 1 - (i + n/2) % n), a reversed rotation by half (kitti.h:180)
    ring = np.array([[-1, 1, k] for k in range(5)], dtype=np.float32)
    got = R.segment_points_scalar(ring, np.eye(4, dtype=np.float32))[0][:, 2]
    assert list(got) == [2, 1, 0, 4, 3]
