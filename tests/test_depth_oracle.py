"""CPU: the oracle's restatement of projectLidarToCamera + featureDepthAssociation (velo.h:329-497) against a plain-Python
re-derivation on crafted rings (pops, drops, equal depth, short rings, > 64 rings) and against geometric properties on the
synthetic street scan.  Also builds the tests/golden/depth_mini.npz fixture check (oracle-generated: parity unpinned)."""
import os

import numpy as np
import pytest

import oracle_lib as O
from velo_amd import synth

F = np.float32
GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "depth_mini.npz")


def py_project(xyz, off, t, w):
    proj, pts, ooff = [], [], [0]
    for s in range(len(off) - 1):
        stack = []                                           # (cx, cy, ppz, i)
        for i in range(off[s], off[s + 1]):
            p = xyz[i]
            pp = (F(p[0] + t[0]), F(p[1] + t[1]), F(p[2] + t[2]))
            with np.errstate(all="ignore"):
                c = (F(pp[0] / pp[2]), F(pp[1] / pp[2]))
            if pp[2] > 0 and w[0] <= float(c[0]) < w[1] and w[2] <= float(c[1]) < w[3]:
                while stack and c[0] < stack[-1][0] and pp[2] < stack[-1][2]:
                    stack.pop()
                if stack and c[0] < stack[-1][0] and pp[2] > stack[-1][2]:
                    continue
                stack.append((c[0], c[1], pp[2], i))
        for e in stack:
            proj.append((e[0], e[1]))
            pts.append(xyz[e[3]])
        ooff.append(len(proj))
    return (np.array(proj, dtype=F).reshape(-1, 2), np.array(pts, dtype=F).reshape(-1, 3), np.array(ooff, dtype=np.int32))


def lerp(p1, p2, start, end, mid):
    a = F(F(mid - start) / F(end - start))
    b = F(F(1) - a)
    return F(F(p1 * b) + F(p2 * a))


def py_depth(proj, pts, off, kps, thresh):
    out, has = [], []
    for kp in kps:
        h, last = -1, -1
        for s in range(len(off) - 1):
            b, n = off[s], off[s + 1] - off[s]
            found = False
            if n <= 1:
                last = -1
                continue
            lo, hi = 0, n - 2
            while lo <= hi:
                mid = (lo + hi) // 2
                if proj[b + mid, 0] > kp[0]:
                    hi = mid - 1
                elif proj[b + mid + 1, 0] <= kp[0]:
                    lo = mid + 1
                else:
                    found = True
                    if last != -1:
                        pb = off[s - 1]
                        a0, a1, b0, b1 = proj[b + mid], proj[b + mid + 1], proj[pb + last], proj[pb + last + 1]
                        if ((a0[1] > kp[1]) != (b0[1] > kp[1]) and abs(float(F(a0[0] - a1[0]))) < thresh
                                and abs(float(F(b0[0] - b1[0]))) < thresh):
                            i1 = [lerp(pts[b + mid, k], pts[b + mid + 1, k], a0[0], a1[0], kp[0]) for k in range(3)]
                            i2 = [lerp(pts[pb + last, k], pts[pb + last + 1, k], b0[0], b1[0], kp[0]) for k in range(3)]
                            i1y = lerp(a0[1], a1[1], a0[0], a1[0], kp[0])
                            i2y = lerp(b0[1], b1[1], b0[0], b1[0], kp[0])
                            out.append([lerp(i1[k], i2[k], i1y, i2y, kp[1]) for k in range(3)])
                            h = len(out) - 1
                    last = mid
                    break
            if not found:
                last = -1
            if h != -1:
                break
        has.append(h)
    return np.array(out, dtype=F).reshape(-1, 3), np.array(has, dtype=np.int32)


def crafted_rings(n_rings=70, seed=11):
    """Rings of a few dozen points in front of the camera with deliberate depth discontinuities: foreground slabs that occlude
    what was pushed before (pops), background points behind the stack top (drops), exact depth ties, empty and 1-point rings."""
    rng = np.random.default_rng(seed)
    xyz, off = [], [0]
    for s in range(n_rings):
        if s % 17 == 5:
            n = 0
        elif s % 17 == 9:
            n = 1
        else:
            n = int(rng.integers(20, 60))
        az = np.sort(rng.uniform(-0.9, 0.9, n))              # tan(azimuth): some fall outside the window on purpose
        z = np.where(rng.uniform(size=n) < 0.3, 6.0, 14.0) + rng.normal(0, 0.05, n)
        z[rng.uniform(size=n) < 0.1] = 14.0                  # exact ties in depth
        jitter = rng.normal(0, 0.03, n)                      # makes x non-monotone now and then
        y = (0.2 - 0.4 * s / n_rings) * z + rng.normal(0, 0.01, n)
        ring = np.stack([(az + jitter) * z, y, z], axis=1)
        if n > 4:
            ring[3, 2] = -1.0                                # behind the camera
        xyz.append(ring)
        off.append(off[-1] + n)
    return np.concatenate(xyz).astype(F), np.array(off, dtype=np.int32)


@pytest.mark.parametrize("cam", [0, 1])
def test_projection_matches_python_restatement_on_crafted_rings(cam):
    xyz, off = crafted_rings()
    w = synth.cam_window()
    got = O.project_lidar(xyz, off, synth.CAM_TRANS[cam], w)
    want = py_project(xyz, off, synth.CAM_TRANS[cam], w)
    assert np.array_equal(got[2], want[2])
    assert np.array_equal(got[0].view(np.uint32), want[0].view(np.uint32))
    assert np.array_equal(got[1].view(np.uint32), want[1].view(np.uint32))
    # the crafted data really exercises the stack: fewer survivors than points inside the window, and non-monotone x remains
    assert 0 < len(got[0]) < len(xyz)


def test_depth_association_matches_python_restatement_on_crafted_rings():
    xyz, off = crafted_rings()
    w = synth.cam_window()
    proj, pts, poff = O.project_lidar(xyz, off, synth.CAM_TRANS[0], w)
    kps = synth.keypoints_in_window(300, seed=3)
    kps[5] = (np.nan, 0.0)
    kps[6] = (0.0, np.nan)
    for thresh in (0.015, 0.2):
        got = O.depth_association(proj, pts, poff, kps, thresh)
        want = py_depth(proj, pts, poff, kps, thresh)
        assert np.array_equal(got[1], want[1])
        assert np.array_equal(got[0].view(np.uint32), want[0].view(np.uint32))
    assert (got[1] >= 0).sum() > 20


def test_depth_on_street_scan_reprojects_onto_the_keypoint():
    d = synth.scan_pair(n_beams=64, n_azimuth=1875)
    w = synth.cam_window()
    for cam in (0, 1):
        t = synth.CAM_TRANS[cam]
        proj, pts, off = O.project_lidar(d["tgt_xyz"], d["tgt_off"], t, w)
        assert np.all(np.diff(off) >= 0) and off[-1] == len(proj)
        q = pts + t
        assert np.array_equal((q[:, 0] / q[:, 2]).astype(F), proj[:, 0])       # every kept entry is its point's projection
        assert np.all(q[:, 2] > 0)
        kps = synth.keypoints_in_window(2000, seed=21)
        kd, has = O.depth_association(proj, pts, off, kps)
        sel = has >= 0
        assert 0.3 < sel.mean() < 0.9
        assert np.array_equal(has[sel], np.arange(sel.sum()))                # appended in keypoint order
        r = kd + t
        err = np.abs(np.stack([r[:, 0] / r[:, 2], r[:, 1] / r[:, 2]], axis=1) - kps[sel])
        # bilinear interpolation between four points that straddle the keypoint: exact on planes, loose at depth edges
        assert np.median(err) < 2e-3 and np.percentile(err, 90) < 0.02


def test_golden_depth_fixture():
    g = np.load(GOLDEN)
    proj, pts, off = O.project_lidar(g["xyz"], g["off"], g["cam_t"], g["window"])
    assert np.array_equal(off, g["proj_off"])
    assert np.array_equal(proj.view(np.uint32), g["proj_xy"].view(np.uint32))
    assert np.array_equal(pts.view(np.uint32), g["proj_pts"].view(np.uint32))
    kd, has = O.depth_association(proj, pts, off, g["keypoints"], float(g["thresh"]))
    assert np.array_equal(has, g["has_depth"])
    assert np.array_equal(kd.view(np.uint32), g["kp_with_depth"].view(np.uint32))
