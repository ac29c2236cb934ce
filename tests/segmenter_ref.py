"""The CHECKER of the device ring segmenter: restatements of the reference's segmentPoints (kitti.h:154-185).
Lives under tests/ -- the package's own copy (synth.segment_points) GENERATES every bench and test input, so it must not also be
the thing the device kernels are checked against.  Two forms:

  segment_points_scalar   a literal, point-by-point transcription of kitti.h:158-183 (plain Python loops, float32 scalars) -- the pin;
  segment_points          the same in vectorised numpy, fast enough for 120k-point scans; tests/test_segmenter_ref.py holds it, and the
                          generator's copy, to the scalar form bit for bit.

pcl::transformPointCloud<PointXYZ> [3P, PCL 1.7/1.8] for an affine float matrix evaluates  p' = L p + t  with Eigen's coefficient
product: ((m_i0 x + m_i1 y) + m_i2 z) + m_i3 in float, which is what both forms do."""
import numpy as np


def segment_points_scalar(pts_velo, velo_to_cam):
    f32 = np.float32
    p = np.asarray(pts_velo, dtype=np.float32)
    M = np.asarray(velo_to_cam, dtype=np.float32)
    n = p.shape[0]
    cloud_tmp = []                                             # kitti.h:162  pcl::transformPointCloud(*point_cloud, *cloud_tmp, velo_to_cam)
    for i in range(n):
        x, y, z = p[i, 0], p[i, 1], p[i, 2]
        cloud_tmp.append(tuple(f32(f32(f32(M[r, 0] * x) + f32(M[r, 1] * y)) + f32(M[r, 2] * z)) + M[r, 3] for r in range(3)))
    prev_y = f32(0)                                            # kitti.h:158
    scan_id = 0                                                # kitti.h:159
    scan_ids = []                                              # kitti.h:163
    for i in range(n):                                         # kitti.h:164
        px, py = p[i, 0], p[i, 1]
        if i > 0 and px > 0 and (py > 0) != (prev_y > 0):      # kitti.h:166
            scan_id += 1
        if scan_id >= len(scan_ids):                           # kitti.h:169-173
            scan_ids.append([])
        scan_ids[scan_id].append(i)                            # kitti.h:174
        prev_y = py                                            # kitti.h:175
    out, offsets = [], [0]
    for s in range(len(scan_ids)):                             # kitti.h:178
        _i = len(scan_ids[s])
        for i in range(_i):                                    # kitti.h:179
            out.append(cloud_tmp[scan_ids[s][_i - 1 - (i + _i // 2) % _i]])   # kitti.h:180
        offsets.append(len(out))
    return np.array(out, dtype=np.float32).reshape(-1, 3), np.array(offsets, dtype=np.int32)


def segment_points(pts_velo, velo_to_cam):
    p = np.asarray(pts_velo, dtype=np.float32)
    M = np.asarray(velo_to_cam, dtype=np.float32)
    n = p.shape[0]
    cam = np.empty((n, 3), dtype=np.float32)
    for r in range(3):
        cam[:, r] = ((M[r, 0] * p[:, 0] + M[r, 1] * p[:, 1]) + M[r, 2] * p[:, 2]) + M[r, 3]
    if n == 0:
        return cam, np.zeros(1, dtype=np.int32)
    flips = np.zeros(n, dtype=bool)
    flips[1:] = (p[1:, 0] > 0) & ((p[1:, 1] > 0) != (p[:-1, 1] > 0))
    ring = np.cumsum(flips)
    counts = np.bincount(ring)
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    out = np.empty_like(cam)
    for s, m in enumerate(counts):
        i = np.arange(m)
        out[off[s]:off[s + 1]] = cam[off[s] + (m - 1 - (i + m // 2) % m)]
    return out, off
