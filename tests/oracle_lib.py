"""ctypes wrapper of oracle/_build/libvelo_oracle.so -- the CHECKER.  Lives under tests/ because only tests,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may touch the oracle."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

import velo_amd  # noqa: F401  (import shim)
from velo_amd.api import (CORR_DTYPE, GOOD_DTYPE, MATCH_DTYPE, VeloParams, VeloSolveSummary, VeloSummary,
                          matches_from_dict)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "_build", "libvelo_oracle.so")
# VELO_ORACLE_SO: load another build of the same source instead (the ASan/UBSan one of `make -C oracle asan`, see tests/README)
ORACLE_SO_OVERRIDE = os.environ.get("VELO_ORACLE_SO")

_dp = C.POINTER(C.c_double)
# layouts of the oracle's test hooks (oracle/velo_oracle.cpp: vo_trace_row, vo_block)
TRACE_DTYPE = np.dtype([("iteration", "<i4"), ("status", "<i4"), ("cost", "<f8"), ("candidate_cost", "<f8"), ("radius", "<f8"),
                        ("model_change", "<f8"), ("step_norm", "<f8"), ("relative_decrease", "<f8"), ("gradient_max", "<f8")])
BLOCK_DTYPE = np.dtype([("kind", "<i4"), ("loss_type", "<i4"), ("c", "<f8", (9,)), ("loss_a", "<f8"), ("loss_w", "<f8")])
TRACE_ACCEPTED, TRACE_REJECTED, TRACE_INVALID, TRACE_PARAMETER_TOL, TRACE_FUNCTION_TOL, TRACE_GRADIENT_TOL = 1, 0, -1, 2, 3, 4
ORACLE_SO_LIBM = os.path.join(ORACLE_DIR, "_build", "libvelo_oracle_libm.so")   # -DVELO_ORACLE_LIBM: the host libm's sin / cos
_libs = {}


def build_oracle(force: bool = False, libm: bool = False) -> str:
    """Builds (make -C oracle) when stale.  libm=True: the build that calls the host libm's sin / cos -- what the reference's
    ceres::AngleAxisRotatePoint does -- instead of the pinned routine the HIP library shares (tests/test_oracle_libm.py)."""
    if ORACLE_SO_OVERRIDE and not libm:
        return ORACLE_SO_OVERRIDE
    src = os.path.join(ORACLE_DIR, "velo_oracle.cpp")
    hdr = os.path.join(ROOT, "include", "velo_hip.h")
    so = ORACLE_SO_LIBM if libm else ORACLE_SO
    if (force or not os.path.exists(so)
            or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr))):
        subprocess.run(["make", "-C", ORACLE_DIR, "-B"], check=True, stdout=subprocess.DEVNULL)
    return so


def lib(libm: bool = False) -> C.CDLL:
    if libm not in _libs:
        l = C.CDLL(build_oracle(libm=libm))
        l.vo_create.restype = C.c_void_p
        for name in ("vo_destroy", "vo_set_params", "vo_set_threads", "vo_set_target", "vo_set_source",
                     "vo_set_visual", "vo_associate", "vo_get_correspondences", "vo_build_visual",
                     "vo_get_good_matches", "vo_evaluate", "vo_evaluate_rows", "vo_solve", "vo_frame_to_frame",
                     "vo_ring_nn", "vo_set_query_shard", "vo_max_threads", "vo_uses_libm", "vo_sincos",
                     "vo_set_variant", "vo_set_variant_normal", "vo_solve_trace", "vo_get_blocks", "vo_tie_census"):
            getattr(l, name).restype = C.c_int
        l.vo_set_variant_normal.restype = C.c_longlong
        assert l.vo_uses_libm() == (1 if libm else 0)
        _libs[libm] = l
    return _libs[libm]


def sincos(x, libm: bool = False):
    """sin / cos as this oracle build computes them inside AngleAxisRotatePoint (pinned routine, or the host libm)."""
    xv = np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(-1))
    s, c = np.zeros_like(xv), np.zeros_like(xv)
    lib(libm).vo_sincos(xv.ctypes.data_as(_dp), C.c_int32(xv.size), s.ctypes.data_as(_dp), c.ctypes.data_as(_dp))
    return s, c


def _d(a, n):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1))
    assert a.size == n
    return a


def default_params() -> VeloParams:
    p = VeloParams()
    lib().vo_default_params(C.byref(p))
    return p


def max_threads() -> int:
    return lib().vo_max_threads()


class Oracle:
    def __init__(self, threads: int = 1, libm: bool = False, **params):
        self._l = lib(libm)
        self._h = C.c_void_p(self._l.vo_create())
        self._l.vo_set_threads(self._h, int(threads))
        self.params = default_params()
        if params:
            self.set_params(**params)

    def __del__(self):
        try:
            self._l.vo_destroy(self._h)
        except Exception:
            pass

    def set_params(self, **kw):
        for k, v in kw.items():
            assert hasattr(self.params, k), k
            setattr(self.params, k, v)
        self._l.vo_set_params(self._h, C.byref(self.params))

    def set_threads(self, n):
        self._l.vo_set_threads(self._h, int(n))

    def set_query_shard(self, rank, world):
        assert self._l.vo_set_query_shard(self._h, int(rank), int(world)) == 0

    def set_target(self, xyz, off):
        a = np.ascontiguousarray(xyz, dtype=np.float32)
        o = np.ascontiguousarray(off, dtype=np.int32)
        self._l.vo_set_target(self._h, C.c_void_p(a.ctypes.data), C.c_int64(a.strides[0]),
                              C.c_void_p(o.ctypes.data), C.c_int32(len(o) - 1))

    def set_source(self, xyz, off):
        a = np.ascontiguousarray(xyz, dtype=np.float32)
        o = np.ascontiguousarray(off, dtype=np.int32)
        self._l.vo_set_source(self._h, C.c_void_p(a.ctypes.data), C.c_int64(a.strides[0]),
                              C.c_void_p(o.ctypes.data), C.c_int32(len(o) - 1))

    def set_visual(self, matches):
        if isinstance(matches, dict):
            matches = matches_from_dict(matches)
        m = np.ascontiguousarray(matches, dtype=MATCH_DTYPE)
        self._l.vo_set_visual(self._h, C.c_void_p(m.ctypes.data), C.c_int32(len(m)))

    def associate(self, x, it):
        n = C.c_int32(0)
        xv = _d(x, 6)
        self._l.vo_associate(self._h, xv.ctypes.data_as(_dp), C.c_int32(it), C.byref(n))
        return n.value

    def correspondences(self):
        n = C.c_int32(0)
        self._l.vo_get_correspondences(self._h, None, C.c_int32(0), C.byref(n))
        out = np.zeros(n.value, dtype=CORR_DTYPE)
        self._l.vo_get_correspondences(self._h, C.c_void_p(out.ctypes.data), n, C.byref(n))
        return out

    def build_visual(self, x, it):
        n = C.c_int32(0)
        xv = _d(x, 6)
        self._l.vo_build_visual(self._h, xv.ctypes.data_as(_dp), C.c_int32(it), C.byref(n))
        return n.value

    def good_matches(self):
        n = C.c_int32(0)
        self._l.vo_get_good_matches(self._h, None, C.c_int32(0), C.byref(n))
        out = np.zeros(n.value, dtype=GOOD_DTYPE)
        self._l.vo_get_good_matches(self._h, C.c_void_p(out.ctypes.data), n, C.byref(n))
        return out

    def evaluate(self, x):
        xv = _d(x, 6)
        cost = C.c_double(0)
        H = np.zeros(36)
        g = np.zeros(6)
        self._l.vo_evaluate(self._h, xv.ctypes.data_as(_dp), C.byref(cost), H.ctypes.data_as(_dp),
                            g.ctypes.data_as(_dp))
        return cost.value, H.reshape(6, 6), g

    def evaluate_rows(self, x):
        xv = _d(x, 6)
        n = C.c_int32(0)
        self._l.vo_evaluate_rows(self._h, xv.ctypes.data_as(_dp), None, None, C.c_int32(0), C.byref(n))
        r = np.zeros(n.value)
        J = np.zeros((n.value, 6))
        self._l.vo_evaluate_rows(self._h, xv.ctypes.data_as(_dp), r.ctypes.data_as(_dp), J.ctypes.data_as(_dp),
                                 n, C.byref(n))
        return r, J

    def solve(self, x):
        xv = _d(x, 6).copy()
        s = VeloSolveSummary()
        self._l.vo_solve(self._h, xv.ctypes.data_as(_dp), C.byref(s))
        return xv, s

    # --- parity-budget hooks (tests/test_parity_budget.py) ---
    def set_variant(self, qr=False, ftol_apply=False, tie_high=False):
        """Alternatives for un-pinned third-party behaviour: the LM step by Householder QR of [J; D] (Ceres' DENSE_QR) instead of
        the 6x6 Cholesky; a successful step that meets the function tolerance applied before terminating; exact in-ring distance
        ties to the HIGHEST index (FLANN's traversal order is unpinned) instead of the lowest."""
        self._l.vo_set_variant(self._h, C.c_int(1 if qr else 0), C.c_int(1 if ftol_apply else 0), C.c_int(1 if tie_high else 0))

    def set_variant_normal(self, norm_split=False, cross_fma=False):
        """The plane normal's float arithmetic under the other readings of Eigen (velo.h:868-874): Vector3f::norm() summed as
        x^2 + (y^2 + z^2) (the unrolled redux splitting 3 as 1 + 2) instead of (x^2 + y^2) + z^2; the cross product's a*b - c*d contracted
        to one fma.  Returns the number of correspondences the ||N|| < 1e-5 test (velo.h:873) dropped since the last call."""
        self._l.vo_set_variant_normal.restype = C.c_longlong
        return int(self._l.vo_set_variant_normal(self._h, C.c_int(1 if norm_split else 0), C.c_int(1 if cross_fma else 0)))

    def solve_trace(self, x):
        """solve() plus one TRACE_DTYPE row per LM iteration."""
        xv = _d(x, 6).copy()
        s = VeloSolveSummary()
        rows = np.zeros(64, dtype=TRACE_DTYPE)
        n = self._l.vo_solve_trace(self._h, xv.ctypes.data_as(_dp), C.byref(s), C.c_void_p(rows.ctypes.data), C.c_int32(len(rows)))
        assert 0 <= n <= len(rows)
        return xv, s, rows[:n]

    def blocks(self):
        """The residual blocks of the current problem in solver order (visual first, then point-to-plane)."""
        n = self._l.vo_get_blocks(self._h, None, C.c_int32(0))
        out = np.zeros(n, dtype=BLOCK_DTYPE)
        self._l.vo_get_blocks(self._h, C.c_void_p(out.ctypes.data), C.c_int32(n))
        return out

    def tie_census(self, x, it):
        """Exact-distance ties of one association round at pose x: dict of counts (see vo_tie_census)."""
        xv = _d(x, 6)
        out = np.zeros(8, dtype=np.int64)
        self._l.vo_tie_census(self._h, xv.ctypes.data_as(_dp), C.c_int32(it), C.c_void_p(out.ctypes.data))
        keys = ("queries", "gated_ring_pairs", "in_ring_ties", "in_ring_ties_on_a_winner", "cross_ring_tie_first",
                "cross_ring_tie_second", "queries_with_two_rings")
        return {k: int(v) for k, v in zip(keys, out)}

    def set_residual_stats(self, enable=True):
        self._l.vo_set_residual_stats(self._h, C.c_int(1 if enable else 0))

    def residual_stats(self, x):
        from velo_amd.api import VeloResidualStats
        xv = _d(x, 6).copy()
        out = VeloResidualStats()
        self._l.vo_residual_stats_at(self._h, xv.ctypes.data_as(_dp), C.byref(out))
        return out

    def frame_to_frame(self, x0):
        xv = _d(x0, 6).copy()
        T = np.zeros(16)
        s = VeloSummary()
        self._l.vo_frame_to_frame(self._h, xv.ctypes.data_as(_dp), T.ctypes.data_as(_dp), C.byref(s))
        return xv, T.reshape(4, 4), s

    def ring_nn(self, ring, q):
        q = np.ascontiguousarray(q, dtype=np.float32)
        it, ib = C.c_int(0), C.c_int(0)
        dt, db = C.c_float(0), C.c_float(0)
        f = self._l.vo_ring_nn(self._h, C.c_int(ring), C.c_void_p(q.ctypes.data), C.byref(it), C.byref(dt),
                               C.byref(ib), C.byref(db))
        return f, it.value, dt.value, ib.value, db.value


def functor(kind: int, c, x):
    """(residuals, jacobian dim x 6) of one residual functor at x; kind 0..3 = ResidualType order, 4 = cost3DPD."""
    cv = np.zeros(9)
    cv[:len(c)] = c
    xv = _d(x, 6)
    r = np.zeros(3)
    J = np.zeros(18)
    d = lib().vo_functor(C.c_int(kind), cv.ctypes.data_as(_dp), xv.ctypes.data_as(_dp), r.ctypes.data_as(_dp),
                         J.ctypes.data_as(_dp))
    return r[:d].copy(), J[:6 * d].reshape(d, 6).copy()


def loss(type_: int, a: float, w: float, s: float):
    rho = np.zeros(3)
    lib().vo_loss(C.c_int(type_), C.c_double(a), C.c_double(w), C.c_double(s), rho.ctypes.data_as(_dp))
    return rho


def transform_point(p, x):
    p = np.ascontiguousarray(p, dtype=np.float32)
    out = np.zeros(3, dtype=np.float32)
    xv = _d(x, 6)
    lib().vo_transform_point(C.c_void_p(p.ctypes.data), xv.ctypes.data_as(_dp), C.c_void_p(out.ctypes.data))
    return out


def rotate_point(w, p):
    wv, pv = _d(w, 3), _d(p, 3)
    out = np.zeros(3)
    lib().vo_rotate_point(wv.ctypes.data_as(_dp), pv.ctypes.data_as(_dp), out.ctypes.data_as(_dp))
    return out


def pose_vec_to_mat(x):
    T = np.zeros(16)
    lib().vo_pose_vec_to_mat(_d(x, 6).ctypes.data_as(_dp), T.ctypes.data_as(_dp))
    return T.reshape(4, 4)


def pose_mat_to_vec(T):
    x = np.zeros(6)
    lib().vo_pose_mat_to_vec(_d(T, 16).ctypes.data_as(_dp), x.ctypes.data_as(_dp))
    return x


# ---- SURVEY.md 8(f) row 3 (stateless oracle functions) -----------------------------------------------------------------
def project_lidar(xyz, ring_off, cam_t, window):
    """-> (proj_xy [n,2], points_xyz [n,3], ring_offsets [Rs+1]) per velo.h:329-374."""
    l = lib()
    xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
    off = np.ascontiguousarray(ring_off, dtype=np.int32)
    t = np.ascontiguousarray(cam_t, dtype=np.float32).reshape(3)
    w = _d(window, 4)
    n = len(xyz)
    proj = np.zeros((max(n, 1), 2), dtype=np.float32)
    pts = np.zeros((max(n, 1), 3), dtype=np.float32)
    ooff = np.zeros(len(off), dtype=np.int32)
    l.vo_project_lidar.restype = C.c_int
    m = l.vo_project_lidar(C.c_void_p(xyz.ctypes.data), C.c_int64(12), C.c_void_p(off.ctypes.data), C.c_int32(len(off) - 1),
                           C.c_void_p(t.ctypes.data), w.ctypes.data_as(_dp), C.c_void_p(proj.ctypes.data), C.c_void_p(pts.ctypes.data),
                           C.c_void_p(ooff.ctypes.data))
    return proj[:m].copy(), pts[:m].copy(), ooff


def depth_association(proj_xy, points_xyz, ring_off, keypoints_xy, thresh=0.015):
    """-> (kp_with_depth [m,3], has_depth [n]) per velo.h:376-497."""
    l = lib()
    proj = np.ascontiguousarray(proj_xy, dtype=np.float32).reshape(-1, 2)
    pts = np.ascontiguousarray(points_xyz, dtype=np.float32).reshape(-1, 3)
    off = np.ascontiguousarray(ring_off, dtype=np.int32)
    kp = np.ascontiguousarray(keypoints_xy, dtype=np.float32).reshape(-1, 2)
    n = len(kp)
    out = np.zeros((max(n, 1), 3), dtype=np.float32)
    has = np.full(max(n, 1), -1, dtype=np.int32)
    l.vo_depth_association.restype = C.c_int
    m = l.vo_depth_association(C.c_void_p(proj.ctypes.data), C.c_void_p(pts.ctypes.data), C.c_void_p(off.ctypes.data),
                               C.c_int32(len(off) - 1), C.c_void_p(kp.ctypes.data), C.c_int32(n), C.c_double(thresh),
                               C.c_void_p(out.ctypes.data), C.c_void_p(has.ctypes.data))
    return out[:m].copy(), has[:n].copy()


# ---- SURVEY.md 8(f) row 4 ----------------------------------------------------------------------------------------------
def triangulate_points(camera_poses, cam_trans, obs, obs_offsets, points_xyz, initial_guess=None, params=None):
    """-> (points [n,3] f32, results [n]) per velo.h:1027-1130, one landmark after the other."""
    from velo_amd.api import TRI_RESULT_DTYPE, pack_triangulation
    l = lib()
    poses, ct, ob, off, pts, init = pack_triangulation(camera_poses, cam_trans, obs, obs_offsets, points_xyz, initial_guess)
    P = params if params is not None else default_params()
    n = len(off) - 1
    res = np.zeros(n, dtype=TRI_RESULT_DTYPE)
    vp = lambda a: C.c_void_p(a.ctypes.data) if a is not None and a.size else None   # noqa: E731
    l.vo_triangulate_points.restype = C.c_int
    l.vo_triangulate_points(C.byref(P), vp(poses), C.c_int32(len(poses)), vp(ct), C.c_int32(len(ct)), vp(ob), vp(off), C.c_int32(n),
                            vp(pts), vp(init), vp(res))
    return pts, res


def tri_functor(kind, cam_pose, s, t, x):
    """One triangulation functor (costfunctions.h:288-375) with its dual-number Jacobian: (r [d], J [d,3])."""
    l = lib()
    r = np.zeros(3)
    J = np.zeros((3, 3))
    l.vo_tri_functor.restype = C.c_int
    d = l.vo_tri_functor(C.c_int(kind), _d(cam_pose, 6).ctypes.data_as(_dp), _d(s, 3).ctypes.data_as(_dp), _d(t, 3).ctypes.data_as(_dp),
                         _d(x, 3).ctypes.data_as(_dp), r.ctypes.data_as(_dp), J.ctypes.data_as(_dp))
    return r[:d].copy(), J[:d].copy()
