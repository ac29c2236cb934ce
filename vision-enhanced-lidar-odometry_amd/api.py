"""ctypes mirror of include/velo_hip.h -- plumbing only, no arithmetic happens in Python.

The shared library (csrc/libvelo_hip.so, built by build.py / __graft_entry__.build()) is the product.
If it is missing or no MI355X is visible every entry point raises: there is NO CPU fallback and this
module never imports anything from oracle/.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libvelo_hip.so")

VELO_MAX_SOLVES = 64
RESIDUAL_NAMES = {0: "3D3D", 1: "3D2D", 2: "2D3D", 3: "2D2D"}
TERMINATION_NAMES = {0: "CONVERGENCE", 1: "NO_CONVERGENCE", 2: "FAILURE"}


class VeloError(RuntimeError):
    pass


class VeloParams(C.Structure):
    _fields_ = [
        ("icp_skip", C.c_int32), ("f2f_iterations", C.c_int32), ("icp_iterations", C.c_int32),
        ("enable_icp", C.c_int32), ("enable_2d2d", C.c_int32), ("enable_3d2d", C.c_int32),
        ("max_num_iterations", C.c_int32), ("max_consecutive_invalid_steps", C.c_int32),
        ("weight_3D2D", C.c_double), ("weight_2D2D", C.c_double), ("weight_3DPD", C.c_double),
        ("loss_thresh_3D2D", C.c_double), ("loss_thresh_2D2D", C.c_double),
        ("loss_thresh_3DPD", C.c_double), ("loss_thresh_3D3D", C.c_double),
        ("outlier_reject", C.c_double), ("correspondence_thresh_icp", C.c_double),
        ("icp_norm_condition", C.c_double),
        ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double),
        ("parameter_tolerance", C.c_double), ("initial_trust_region_radius", C.c_double),
        ("max_trust_region_radius", C.c_double), ("min_trust_region_radius", C.c_double),
        ("min_relative_decrease", C.c_double), ("min_lm_diagonal", C.c_double),
        ("max_lm_diagonal", C.c_double),
    ]


class VeloSolveSummary(C.Structure):
    _fields_ = [
        ("termination", C.c_int32), ("lm_iterations", C.c_int32), ("evaluations", C.c_int32),
        ("n_icp_valid", C.c_int32), ("n_visual_blocks", C.c_int32), ("n_visual_residuals", C.c_int32),
        ("initial_cost", C.c_double), ("final_cost", C.c_double),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class VeloScanRef(C.Structure):
    _fields_ = [("xyz", C.c_void_p), ("stride_bytes", C.c_int64), ("ring_offsets", C.c_void_p), ("n_rings", C.c_int32),
                ("on_device", C.c_int32)]


VELO_MAX_STATS = 4
RESIDUAL_TYPE_NAMES = ("3D3D", "3D2D", "2D3D", "2D2D", "3DPD")


class VeloResidualStat(C.Structure):
    _fields_ = [("median", C.c_double), ("mean", C.c_double), ("count", C.c_int64)]


class VeloResidualStats(C.Structure):
    """residualStats (velo.h:921-1025): per residual type median / mean / count of the block norms, loss not applied."""
    _fields_ = [("type", VeloResidualStat * 5), ("cost", C.c_double), ("n_blocks", C.c_int32), ("n_residuals", C.c_int32)]

    def as_dict(self):
        d = {n: dict(median=self.type[k].median, mean=self.type[k].mean, count=self.type[k].count) for k, n in enumerate(RESIDUAL_TYPE_NAMES)}
        d.update(cost=self.cost, n_blocks=self.n_blocks, n_residuals=self.n_residuals)
        return d


class VeloSummary(C.Structure):
    _fields_ = [
        ("n_solves", C.c_int32), ("n_assoc_rounds", C.c_int32), ("n_queries", C.c_int32),
        ("n_target", C.c_int32),
        ("algorithmic_bytes", C.c_uint64), ("assoc_bytes", C.c_uint64),
        ("assoc_kernel_ms", C.c_double), ("assoc_kernel_launches", C.c_int32),
        ("eval_kernel_launches", C.c_int32), ("eval_kernel_ms", C.c_double),
        ("solves", VeloSolveSummary * VELO_MAX_SOLVES),
        ("n_residual_stats", C.c_int32), ("reserved", C.c_int32),
        ("residual_stats", VeloResidualStats * VELO_MAX_STATS),
    ]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_ if k not in ("solves", "residual_stats", "reserved")}
        d["solves"] = [self.solves[i].as_dict() for i in range(min(self.n_solves, VELO_MAX_SOLVES))]
        d["residual_stats"] = [self.residual_stats[i].as_dict() for i in range(min(self.n_residual_stats, VELO_MAX_STATS))]
        return d


MATCH_DTYPE = np.dtype([
    ("p3_1", np.float32, 3), ("p3_2", np.float32, 3), ("p2_1", np.float32, 2), ("p2_2", np.float32, 2),
    ("t_cam", np.float32, 3), ("cam", np.int32), ("point1", np.int32), ("point2", np.int32),
    ("d1", np.uint8), ("d2", np.uint8), ("pad", np.uint8, 2)], align=True)
GOOD_DTYPE = np.dtype([("cam", np.int32), ("point1", np.int32), ("point2", np.int32),
                       ("residual_type", np.int32)], align=True)
CORR_DTYPE = np.dtype([
    ("valid", np.int32), ("ring_i", np.int32), ("idx_i", np.int32), ("ring_j", np.int32),
    ("idx_j", np.int32), ("idx_k", np.int32), ("src_ring", np.int32), ("src_idx", np.int32),
    ("dist_i", np.float32), ("dist_j", np.float32),
    ("p", np.float32, 3), ("n", np.float32, 3), ("v0", np.float32, 3)], align=True)
PARTIAL_DTYPE = np.dtype([
    ("key1", np.uint64), ("key2", np.uint64), ("ring1", np.int32), ("ring2", np.int32), ("idx1", np.int32),
    ("idx_k", np.int32), ("idx2", np.int32), ("pad", np.int32), ("v0", np.float32, 3), ("v2", np.float32, 3),
    ("v1", np.float32, 3), ("pad2", np.float32)], align=True)
TRI_OBS_DTYPE = np.dtype([("kind", np.int32), ("frame", np.int32), ("cam", np.int32), ("s", np.float32, 3)])
TRI_RESULT_DTYPE = np.dtype([("n_solves", np.int32), ("termination", np.int32), ("lm_iterations", np.int32),
                             ("evaluations", np.int32), ("final_cost", np.float64)])
TRI_OBS_3D, TRI_OBS_2D = 0, 1
FUNCTOR_DTYPE = np.dtype([("kind", np.int32), ("reserved", np.int32), ("c", np.float64, 9)])
FUNCTOR_3DPD = 4
assert MATCH_DTYPE.itemsize == 68 and GOOD_DTYPE.itemsize == 16 and CORR_DTYPE.itemsize == 76 and PARTIAL_DTYPE.itemsize == 80
assert TRI_OBS_DTYPE.itemsize == 24 and TRI_RESULT_DTYPE.itemsize == 24 and FUNCTOR_DTYPE.itemsize == 80


KERNEL_TIME_DTYPE = np.dtype([("name", "S48"), ("ms", "<f8"), ("launches", "<i8"), ("sampled", "<i8"), ("algorithmic_bytes", "<u8")])
assert KERNEL_TIME_DTYPE.itemsize == 80


def matches_from_dict(rec: dict) -> np.ndarray:
    """synth.stereo_matches() dict -> packed velo_match array."""
    n = len(rec["cam"])
    m = np.zeros(n, dtype=MATCH_DTYPE)
    for k in ("p3_1", "p3_2", "p2_1", "p2_2", "t_cam", "cam", "point1", "point2", "d1", "d2"):
        m[k] = rec[k]
    return m


def pack_triangulation(camera_poses, cam_trans, obs, obs_offsets, points_xyz, initial_guess):
    """Contiguous, typed copies of the arguments of velo_triangulate_points (shared with the test-side oracle wrapper)."""
    poses = np.ascontiguousarray(np.asarray(camera_poses, dtype=np.float64).reshape(-1, 6))
    ct = np.ascontiguousarray(np.asarray(cam_trans, dtype=np.float32).reshape(-1, 3))
    ob = np.ascontiguousarray(obs, dtype=TRI_OBS_DTYPE)
    off = np.ascontiguousarray(obs_offsets, dtype=np.int32)
    pts = np.array(np.asarray(points_xyz, dtype=np.float32).reshape(-1, 3), copy=True, order="C")
    if len(off) - 1 != len(pts):
        raise ValueError("obs_offsets must have one more entry than there are landmarks")
    init = None if initial_guess is None else np.ascontiguousarray(np.asarray(initial_guess).astype(np.uint8))
    if init is not None and len(init) != len(pts):
        raise ValueError("initial_guess: one flag per landmark")
    return poses, ct, ob, off, pts, init


def default_params() -> VeloParams:
    """The reference's constants (kitti.h:8-10,20-32; main.cpp:43-45) + Ceres defaults, filled in Python so
    that it also works on a box without the library; tests check it against velo_default_params()."""
    p = VeloParams()
    p.icp_skip, p.f2f_iterations, p.icp_iterations = 200, 2, 3
    p.enable_icp = p.enable_2d2d = p.enable_3d2d = 1
    p.max_num_iterations, p.max_consecutive_invalid_steps = 50, 5
    p.weight_3D2D, p.weight_2D2D, p.weight_3DPD = 10.0, 500.0, 1.0
    p.loss_thresh_3D2D, p.loss_thresh_2D2D, p.loss_thresh_3DPD, p.loss_thresh_3D3D = 0.01, 0.00002, 0.1, 0.04
    p.outlier_reject, p.correspondence_thresh_icp, p.icp_norm_condition = 5.0, 0.5, 1e-5
    p.function_tolerance, p.gradient_tolerance, p.parameter_tolerance = 1e-6, 1e-10, 1e-8
    p.initial_trust_region_radius, p.max_trust_region_radius, p.min_trust_region_radius = 1e4, 1e16, 1e-32
    p.min_relative_decrease, p.min_lm_diagonal, p.max_lm_diagonal = 1e-3, 1e-6, 1e32
    return p


# ---- every symbol include/velo_hip.h declares (tests check the .so exports all of them) -----------
_P = C.POINTER
_ctx = C.c_void_p
_dp = _P(C.c_double)
SIGNATURES = {
    "velo_create": (C.c_int, [_P(_ctx), C.c_int]),
    "velo_destroy": (C.c_int, [_ctx]),
    "velo_last_error": (C.c_char_p, []),
    "velo_version": (C.c_char_p, []),
    "velo_default_params": (C.c_int, [_P(VeloParams)]),
    "velo_set_params": (C.c_int, [_ctx, _P(VeloParams)]),
    "velo_get_params": (C.c_int, [_ctx, _P(VeloParams)]),
    "velo_set_timing": (C.c_int, [_ctx, C.c_int]),
    "velo_get_kernel_times": (C.c_int, [_ctx, C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.c_int32]),
    "velo_set_target": (C.c_int, [_ctx, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int]),
    "velo_set_target_part": (C.c_int, [_ctx, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int]),
    "velo_set_source": (C.c_int, [_ctx, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int]),
    "velo_set_scan_velodyne": (C.c_int, [_ctx, C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int]),
    "velo_source_to_target": (C.c_int, [_ctx]),
    "velo_share_target": (C.c_int, [_ctx, _ctx]),
    "velo_cache_create": (C.c_int, [_P(C.c_void_p), C.c_int32, C.c_int32]),
    "velo_cache_destroy": (C.c_int, [C.c_void_p]),
    "velo_cache_store": (C.c_int, [C.c_void_p, C.c_int32, _ctx, C.c_int32]),
    "velo_cache_load": (C.c_int, [C.c_void_p, C.c_int32, _ctx, C.c_int32]),
    "velo_cache_contains": (C.c_int, [C.c_void_p, C.c_int32]),
    "velo_cache_frames": (C.c_int, [C.c_void_p, _P(C.c_int32), C.c_int32]),
    "velo_get_ring_offsets": (C.c_int, [_ctx, C.c_int32, C.c_void_p, C.c_int32, _P(C.c_int32)]),
    "velo_get_cloud": (C.c_int, [_ctx, C.c_int32, C.c_void_p, C.c_int32, _P(C.c_int32)]),
    "velo_set_visual": (C.c_int, [_ctx, C.c_void_p, C.c_int32]),
    "velo_associate": (C.c_int, [_ctx, _dp, C.c_int32, _P(C.c_int32)]),
    "velo_associate_partial": (C.c_int, [_ctx, _dp, C.c_int32]),
    "velo_get_partials": (C.c_int, [_ctx, C.c_void_p, C.c_int32, _P(C.c_int32)]),
    "velo_merge_partials": (C.c_int, [_ctx, _P(C.c_void_p), C.c_int32, _P(C.c_int32)]),
    "velo_get_correspondences": (C.c_int, [_ctx, C.c_void_p, C.c_int32, _P(C.c_int32)]),
    "velo_build_visual": (C.c_int, [_ctx, _dp, C.c_int32, _P(C.c_int32)]),
    "velo_get_good_matches": (C.c_int, [_ctx, C.c_void_p, C.c_int32, _P(C.c_int32)]),
    "velo_evaluate": (C.c_int, [_ctx, _dp, _dp, _dp, _dp]),
    "velo_evaluate_rows": (C.c_int, [_ctx, _dp, _dp, _dp, C.c_int32, _P(C.c_int32)]),
    "velo_evaluate_functors": (C.c_int, [_ctx, C.c_void_p, C.c_int32, _dp, _dp, _dp]),
    "velo_solve": (C.c_int, [_ctx, _dp, _P(VeloSolveSummary)]),
    "velo_frame_to_frame": (C.c_int, [_ctx, _dp, _dp, _P(VeloSummary)]),
    "velo_frame_to_frame_batch": (C.c_int, [_P(_ctx), C.c_int32, _dp, _dp, _P(VeloSummary)]),
    "velo_register_batch": (C.c_int, [_P(_ctx), C.c_int32, C.c_void_p, C.c_void_p, _dp, _dp, _P(VeloSummary)]),
    "velo_register_batch_visual": (C.c_int, [_P(_ctx), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, _dp, _dp, _P(VeloSummary)]),
    "velo_hint_next_source": (C.c_int, [_ctx, C.c_void_p]),
    "velo_hint_next_frame": (C.c_int, [_ctx, C.c_void_p]),
    "velo_register_sequences": (C.c_int, [_P(_ctx), C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, _dp, _dp, _dp, _dp, _P(VeloSummary), C.c_int32]),
    "velo_pose_vec_to_mat": (C.c_int, [_dp, _dp]),
    "velo_pose_mat_to_vec": (C.c_int, [_dp, _dp]),
    "velo_pose_handoff": (C.c_int, [C.c_int32, _dp, _dp, _dp]),
    "velo_comm_unique_id": (C.c_int, [C.c_char_p]),
    "velo_comm_init": (C.c_int, [_ctx, C.c_char_p, C.c_int32, C.c_int32]),
    "velo_comm_destroy": (C.c_int, [_ctx]),
    "velo_comm_peer_export": (C.c_int, [_ctx, C.c_char_p]),
    "velo_comm_peer_attach": (C.c_int, [_ctx, C.c_char_p, C.c_int32, C.c_int32]),
    "velo_comm_info": (C.c_int, [_ctx, _P(C.c_int32), _P(C.c_int32), _P(C.c_int32)]),
    "velo_chain_stats": (C.c_int, [_ctx, _P(C.c_int32), _P(C.c_int32)]),
    "velo_set_residual_stats": (C.c_int, [_ctx, C.c_int]),
    "velo_residual_stats_at": (C.c_int, [_ctx, _P(C.c_double), _P(VeloResidualStats)]),
    "velo_comm_peer_export_records": (C.c_int, [_ctx, C.c_int32, C.c_char_p]),
    "velo_comm_peer_attach_records": (C.c_int, [_ctx, C.c_char_p, C.c_int32]),
    "velo_comm_set_target_sharded": (C.c_int, [_ctx, C.c_int]),
    "velo_set_query_shard": (C.c_int, [_ctx, C.c_int32, C.c_int32]),
    "velo_synchronize": (C.c_int, [_ctx]),
    "velo_project_lidar": (C.c_int, [_ctx, C.c_int32, C.c_void_p, _dp, _P(C.c_int32)]),
    "velo_get_projection": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, _P(C.c_int32)]),
    "velo_depth_association": (C.c_int, [_ctx, C.c_void_p, C.c_int32, C.c_double, C.c_void_p, C.c_int32, C.c_void_p, _P(C.c_int32)]),
    "velo_triangulate_points": (C.c_int, [_ctx, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32,
                                          C.c_void_p, C.c_void_p, C.c_void_p]),
}

_lib = None
_libs_by_path = {}
LIB_PATH_DIAG = os.path.join(os.path.dirname(LIB_PATH), "libvelo_hip_diag.so")


def load_library(path: Optional[str] = None) -> C.CDLL:
    """dlopen the HIP library and type every entry point.  Raises VeloError when it has not been built.
    path: another build of the same source (the tools' / variant tests' diagnostics build, see load_diagnostics_library)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    if path is not None and path in _libs_by_path:
        return _libs_by_path[path]
    p = path or os.environ.get("VELO_LIB_PATH") or LIB_PATH      # VELO_LIB_PATH: A/B builds on one GPU box
    if not os.path.exists(p):
        raise VeloError(f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(p, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header/library drift; tests guard it
        fn.restype, fn.argtypes = res, args
    if path is None:
        _lib = lib
    else:
        _libs_by_path[path] = lib
    return lib


def load_diagnostics_library() -> C.CDLL:
    """The -DVELO_DIAGNOSTICS build of the same source (libvelo_hip_diag.so): the only build that honours the A/B environment switches
    (kernel variants, grid shapes, VELO_DEBUG_SKIP ...).  The product library reads five documented knobs and ignores the rest, so the
    parity tests that sweep variants, and the dev tools, create their contexts on this one: Context(device, lib=load_diagnostics_library())."""
    return load_library(LIB_PATH_DIAG)


def _dvec(a, n):
    arr = np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1))
    if arr.size != n:
        raise ValueError(f"expected {n} doubles, got {arr.size}")
    return arr


def _ptr(arr: np.ndarray):
    return arr.ctypes.data_as(_dp)


class Context:
    """One scan-matching context = one `velo_ctx*` (device buffers + a HIP stream on `device`)."""

    def __init__(self, device: int = 0, lib: Optional[C.CDLL] = None, **params):
        self._lib = lib if lib is not None else load_library()
        self._h = _ctx()
        self._device = int(device)
        self._check(self._lib.velo_create(C.byref(self._h), int(device)))
        if params:
            self.set_params(**params)

    # -- helpers ---------------------------------------------------------------------------------
    def _check(self, status: int):
        if status != 0:
            msg = self._lib.velo_last_error()
            raise VeloError(f"velo status {status}: {msg.decode() if msg else ''}")

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.velo_destroy(self._h)
            self._h = _ctx()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    # -- configuration -------------------------------------------------------------------------------
    def get_params(self) -> VeloParams:
        p = VeloParams()
        self._check(self._lib.velo_get_params(self._h, C.byref(p)))
        return p

    def set_params(self, **kw):
        p = self.get_params()
        for k, v in kw.items():
            if not hasattr(p, k):
                raise AttributeError(f"velo_params has no field {k}")
            setattr(p, k, v)
        self._check(self._lib.velo_set_params(self._h, C.byref(p)))

    def set_timing(self, enable):
        """0 / False: off; 1 / True: association launches into the summary; 2: launches counted by kernel name, every 8th bracketed
        (kernel_times); 3: every launch bracketed."""
        self._check(self._lib.velo_set_timing(self._h, int(enable)))

    def kernel_times(self, reset: bool = True):
        """{kernel name: (ms, launches, algorithmic bytes)} accumulated since the last reset (velo_set_timing(ctx, 2))."""
        n = C.c_int32(0)
        self._check(self._lib.velo_get_kernel_times(self._h, None, 0, C.byref(n), 0))
        out = np.zeros(max(n.value, 1), dtype=KERNEL_TIME_DTYPE)
        self._check(self._lib.velo_get_kernel_times(self._h, C.c_void_p(out.ctypes.data), n.value, C.byref(n), int(bool(reset))))
        return {bytes(r["name"]).split(b"\0")[0].decode(): (float(r["ms"]), int(r["launches"]), int(r["algorithmic_bytes"])) for r in out[:n.value]}

    # -- inputs ----------------------------------------------------------------------------------------
    @staticmethod
    def _device_tensor_args(t, device, cols=(3, 4), what="xyz"):
        """A torch tensor handed over as a device pointer (on_device = 1): checked here because the library only sees an address.
        The library reads it on the context's own stream, so everything torch has queued that produces the tensor must be
        complete first (include/velo_hip.h, "device pointers"): the producing torch stream is synchronised before the call."""
        import torch
        if not t.is_cuda:
            raise ValueError(f"{what}: a torch tensor must live on the GPU (pass a numpy array for host data)")
        if t.dtype != torch.float32:
            raise ValueError(f"{what}: device tensor must be float32, got {t.dtype}")
        if t.dim() != 2 or t.shape[1] not in cols:
            raise ValueError(f"{what}: device tensor must be (n, {' or '.join(str(c) for c in cols)}), got {tuple(t.shape)}")
        if t.shape[0] > 1 and (t.stride(1) != 1 or t.stride(0) < 3):
            raise ValueError(f"{what}: the coordinates of a point must be contiguous (stride(1) == 1), got strides {t.stride()}")
        if device is not None and t.device.index != int(device):
            raise ValueError(f"{what}: tensor is on cuda:{t.device.index}, the context on device {int(device)}")
        torch.cuda.current_stream(t.device).synchronize()
        stride = (t.stride(0) if t.shape[0] > 1 else t.shape[1]) * t.element_size()
        return C.c_void_p(t.data_ptr()), stride

    @staticmethod
    def _cloud_args(xyz, ring_offsets, device=None):
        off = np.ascontiguousarray(np.asarray(ring_offsets, dtype=np.int32))
        if hasattr(xyz, "data_ptr"):   # a torch tensor on the GPU: (n,3) or (n,4) float32, row stride in bytes
            ptr, stride = Context._device_tensor_args(xyz, device)
            if len(off) and int(off[-1]) > xyz.shape[0]:
                raise ValueError(f"ring_offsets end at {int(off[-1])} but the tensor holds {xyz.shape[0]} points")
            return ptr, stride, off, 1, xyz
        a = np.asarray(xyz, dtype=np.float32)
        if a.ndim != 2 or a.shape[1] not in (3, 4):
            raise ValueError("xyz must be (n,3) or (n,4) float32")
        a = np.ascontiguousarray(a)
        if len(off) and int(off[-1]) > a.shape[0]:
            raise ValueError(f"ring_offsets end at {int(off[-1])} but the array holds {a.shape[0]} points")
        return C.c_void_p(a.ctypes.data), a.strides[0], off, 0, a

    def set_target(self, xyz, ring_offsets):
        ptr, stride, off, dev, keep = self._cloud_args(xyz, ring_offsets, self._device)
        self._check(self._lib.velo_set_target(self._h, ptr, stride, C.c_void_p(off.ctypes.data), len(off) - 1, dev))

    def set_target_part(self, xyz, ring_offsets, first_ring: int, first_point: int):
        """Target-sharded mode: this context holds only a block of whole rings (local offsets, [0] == 0)."""
        ptr, stride, off, dev, keep = self._cloud_args(xyz, ring_offsets, self._device)
        self._check(self._lib.velo_set_target_part(self._h, ptr, stride, C.c_void_p(off.ctypes.data), len(off) - 1,
                                                   int(first_ring), int(first_point), dev))

    def associate_partial(self, x, iter: int):
        xv = _dvec(x, 6)
        self._check(self._lib.velo_associate_partial(self._h, _ptr(xv), int(iter)))

    def partials(self) -> np.ndarray:
        n = C.c_int32(0)
        self._check(self._lib.velo_get_partials(self._h, None, 0, C.byref(n)))
        out = np.zeros(n.value, dtype=PARTIAL_DTYPE)
        if n.value:
            self._check(self._lib.velo_get_partials(self._h, C.c_void_p(out.ctypes.data), n.value, C.byref(n)))
        return out

    def merge_partials(self, tables) -> int:
        tabs = [np.ascontiguousarray(t, dtype=PARTIAL_DTYPE) for t in tables]
        arr = (C.c_void_p * len(tabs))(*[t.ctypes.data for t in tabs])
        n = C.c_int32(0)
        self._check(self._lib.velo_merge_partials(self._h, arr, len(tabs), C.byref(n)))
        return n.value

    def comm_set_target_sharded(self, enable: bool):
        self._check(self._lib.velo_comm_set_target_sharded(self._h, int(bool(enable))))

    def set_source(self, xyz, ring_offsets):
        ptr, stride, off, dev, keep = self._cloud_args(xyz, ring_offsets, self._device)
        self._check(self._lib.velo_set_source(self._h, ptr, stride, C.c_void_p(off.ctypes.data), len(off) - 1, dev))

    def set_scan_velodyne(self, as_target: bool, records, velo_to_cam):
        """Raw Velodyne records (n,4) float32 in file order (or a torch tensor on the GPU) -> rings on the device."""
        M = np.ascontiguousarray(np.asarray(velo_to_cam, dtype=np.float32).reshape(4, 4))
        if hasattr(records, "data_ptr"):
            (ptr, stride), n, dev = self._device_tensor_args(records, self._device, cols=(3, 4), what="records"), records.shape[0], 1
        else:
            a = np.ascontiguousarray(np.asarray(records, dtype=np.float32))
            ptr, stride, n, dev = C.c_void_p(a.ctypes.data), a.strides[0], a.shape[0], 0
        self._check(self._lib.velo_set_scan_velodyne(self._h, int(bool(as_target)), ptr, stride, n, C.c_void_p(M.ctypes.data), dev))

    def ring_offsets(self, of_target: bool) -> np.ndarray:
        n = C.c_int32(0)
        self._check(self._lib.velo_get_ring_offsets(self._h, int(bool(of_target)), None, 0, C.byref(n)))
        out = np.zeros(n.value + 1, dtype=np.int32)
        self._check(self._lib.velo_get_ring_offsets(self._h, int(bool(of_target)), C.c_void_p(out.ctypes.data), len(out), C.byref(n)))
        return out

    def source_to_target(self):
        """The device-resident source scan becomes the target of the next registration (no upload, no second segmentation)."""
        self._check(self._lib.velo_source_to_target(self._h))

    def share_target(self, src: "Context"):
        """Take the target `src` holds (cloud + index) by reference: scan-to-map batches keep ONE map for all their contexts."""
        self._check(self._lib.velo_share_target(self._h, src.handle))

    def cloud(self, of_target: bool) -> np.ndarray:
        n = C.c_int32(0)
        self._check(self._lib.velo_get_cloud(self._h, int(bool(of_target)), None, 0, C.byref(n)))
        out = np.zeros((n.value, 3), dtype=np.float32)
        if n.value:
            self._check(self._lib.velo_get_cloud(self._h, int(bool(of_target)), C.c_void_p(out.ctypes.data), n.value, C.byref(n)))
        return out

    def set_visual(self, matches):
        if isinstance(matches, dict):
            matches = matches_from_dict(matches)
        m = np.ascontiguousarray(matches, dtype=MATCH_DTYPE) if matches is not None else np.zeros(0, MATCH_DTYPE)
        self._check(self._lib.velo_set_visual(self._h, C.c_void_p(m.ctypes.data) if len(m) else None, len(m)))

    # -- pieces ------------------------------------------------------------------------------------------
    def associate(self, x, iter: int) -> int:
        xv = _dvec(x, 6)
        n = C.c_int32(0)
        self._check(self._lib.velo_associate(self._h, _ptr(xv), int(iter), C.byref(n)))
        return n.value

    def correspondences(self) -> np.ndarray:
        n = C.c_int32(0)
        self._check(self._lib.velo_get_correspondences(self._h, None, 0, C.byref(n)))
        out = np.zeros(n.value, dtype=CORR_DTYPE)
        if n.value:
            self._check(self._lib.velo_get_correspondences(self._h, C.c_void_p(out.ctypes.data), n.value, C.byref(n)))
        return out

    def build_visual(self, x, iter: int) -> int:
        xv = _dvec(x, 6)
        n = C.c_int32(0)
        self._check(self._lib.velo_build_visual(self._h, _ptr(xv), int(iter), C.byref(n)))
        return n.value

    def good_matches(self) -> np.ndarray:
        n = C.c_int32(0)
        self._check(self._lib.velo_get_good_matches(self._h, None, 0, C.byref(n)))
        out = np.zeros(n.value, dtype=GOOD_DTYPE)
        if n.value:
            self._check(self._lib.velo_get_good_matches(self._h, C.c_void_p(out.ctypes.data), n.value, C.byref(n)))
        return out

    def evaluate(self, x):
        xv = _dvec(x, 6)
        cost = C.c_double(0)
        H = np.zeros(36)
        g = np.zeros(6)
        self._check(self._lib.velo_evaluate(self._h, _ptr(xv), C.byref(cost), _ptr(H), _ptr(g)))
        return cost.value, H.reshape(6, 6), g

    def evaluate_rows(self, x):
        xv = _dvec(x, 6)
        n = C.c_int32(0)
        self._lib.velo_evaluate_rows(self._h, _ptr(xv), None, None, 0, C.byref(n))
        r = np.zeros(n.value)
        J = np.zeros((n.value, 6))
        if n.value:
            self._check(self._lib.velo_evaluate_rows(self._h, _ptr(xv), _ptr(r), _ptr(J), n.value, C.byref(n)))
        return r, J

    def evaluate_functors(self, kinds, consts, x, want_jacobian: bool = True):
        """Seam 2 by value: raw residuals [n,3] and autodiff-equal Jacobians [n,3,6] of n functors (kind 0..3 = ResidualType
        order, 4 = cost3DPD; consts [n,<=9] = constructor arguments in the reference's order) at pose x."""
        kinds = np.asarray(kinds, dtype=np.int32).reshape(-1)
        consts = np.asarray(consts, dtype=np.float64)
        consts = consts.reshape(len(kinds), consts.size // max(len(kinds), 1))
        rec = np.zeros(len(kinds), dtype=FUNCTOR_DTYPE)
        rec["kind"] = kinds
        rec["c"][:, :consts.shape[1]] = consts
        xv = _dvec(x, 6)
        r = np.zeros((len(kinds), 3))
        J = np.zeros((len(kinds), 3, 6)) if want_jacobian else None
        self._check(self._lib.velo_evaluate_functors(self._h, C.c_void_p(rec.ctypes.data), len(kinds), _ptr(xv), _ptr(r),
                                                     _ptr(J) if want_jacobian else None))
        return r, J

    def solve(self, x):
        xv = _dvec(x, 6).copy()
        s = VeloSolveSummary()
        self._check(self._lib.velo_solve(self._h, _ptr(xv), C.byref(s)))
        return xv, s

    # -- the path ------------------------------------------------------------------------------------------
    def frame_to_frame(self, x0):
        xv = _dvec(x0, 6).copy()
        T = np.zeros(16)
        s = VeloSummary()
        self._check(self._lib.velo_frame_to_frame(self._h, _ptr(xv), _ptr(T), C.byref(s)))
        return xv, T.reshape(4, 4), s

    def synchronize(self):
        self._check(self._lib.velo_synchronize(self._h))

    # -- camera projection of a scan + keypoint depth (velo.h:329-497) -------------------------------------------
    def project_lidar(self, of_target: bool, cam_t, window) -> int:
        """projectLidarToCamera for one camera on the rings held as source/target; returns the number of kept points."""
        t = np.ascontiguousarray(np.asarray(cam_t, dtype=np.float32).reshape(-1))
        if t.size != 3:
            raise ValueError("cam_t: 3 floats")
        w = _dvec(window, 4)
        n = C.c_int32(0)
        self._check(self._lib.velo_project_lidar(self._h, int(bool(of_target)), C.c_void_p(t.ctypes.data), _ptr(w), C.byref(n)))
        return n.value

    def projection(self):
        """(proj_xy [n,2] f32, points_xyz [n,3] f32, ring_offsets [Rs+1] i32) of the last project_lidar."""
        nr = C.c_int32(0)
        self._check(self._lib.velo_get_projection(self._h, None, None, 0, None, 0, C.byref(nr)))
        off = np.zeros(nr.value + 1, dtype=np.int32)
        self._check(self._lib.velo_get_projection(self._h, None, None, 0, C.c_void_p(off.ctypes.data), off.size, C.byref(nr)))
        n = int(off[-1])
        xy = np.zeros((n, 2), dtype=np.float32)
        pts = np.zeros((n, 3), dtype=np.float32)
        if n:
            self._check(self._lib.velo_get_projection(self._h, C.c_void_p(xy.ctypes.data), C.c_void_p(pts.ctypes.data), n,
                                                      C.c_void_p(off.ctypes.data), off.size, C.byref(nr)))
        return xy, pts, off

    def depth_association(self, keypoints_xy, thresh: float = 0.015):
        """featureDepthAssociation against the last project_lidar: (kp_with_depth [m,3] f32, has_depth [n] i32)."""
        kp = np.ascontiguousarray(np.asarray(keypoints_xy, dtype=np.float32).reshape(-1, 2))
        n = len(kp)
        has = np.full(n, -1, dtype=np.int32)
        out = np.zeros((max(n, 1), 3), dtype=np.float32)
        m = C.c_int32(0)
        self._check(self._lib.velo_depth_association(self._h, C.c_void_p(kp.ctypes.data) if n else None, n, float(thresh),
                                                     C.c_void_p(out.ctypes.data), n, C.c_void_p(has.ctypes.data) if n else None,
                                                     C.byref(m)))
        return out[:m.value].copy(), has

    # -- batched landmark triangulation (velo.h:1027-1130) ---------------------------------------------------------
    def triangulate_points(self, camera_poses, cam_trans, obs, obs_offsets, points_xyz, initial_guess=None):
        """All landmarks of a frame in one call: (points [n,3] f32, results [n] TRI_RESULT_DTYPE).  `points_xyz` supplies the
        initial guesses of the landmarks flagged in `initial_guess`; it is not modified."""
        args = pack_triangulation(camera_poses, cam_trans, obs, obs_offsets, points_xyz, initial_guess)
        poses, ct, ob, off, pts, init = args
        n = len(off) - 1
        res = np.zeros(n, dtype=TRI_RESULT_DTYPE)
        vp = lambda a: C.c_void_p(a.ctypes.data) if a is not None and a.size else None   # noqa: E731
        self._check(self._lib.velo_triangulate_points(self._h, vp(poses), len(poses), vp(ct), len(ct), vp(ob), vp(off), n,
                                                      vp(pts), vp(init), vp(res)))
        return pts, res

    # -- multi-GPU -------------------------------------------------------------------------------------------
    def set_query_shard(self, rank: int, world: int):
        self._check(self._lib.velo_set_query_shard(self._h, int(rank), int(world)))

    def comm_init(self, unique_id: bytes, rank: int, world: int):
        buf = C.create_string_buffer(bytes(unique_id), 128)
        self._check(self._lib.velo_comm_init(self._h, buf, int(rank), int(world)))

    def comm_destroy(self):
        self._check(self._lib.velo_comm_destroy(self._h))

    def comm_peer_export(self) -> bytes:
        """IPC handle (64 bytes) of this context's all-reduce slab; gather the handles of all ranks, then comm_peer_attach."""
        buf = C.create_string_buffer(64)
        self._check(self._lib.velo_comm_peer_export(self._h, buf))
        return buf.raw

    def comm_peer_attach(self, handles, rank: int, world: int):
        blob = b"".join(bytes(h) for h in handles)
        if len(blob) != 64 * world:
            raise ValueError("one 64-byte handle per rank, in rank order")
        self._check(self._lib.velo_comm_peer_attach(self._h, C.create_string_buffer(blob, len(blob)), int(rank), int(world)))

    def comm_peer_export_records(self, max_queries: int) -> bytes:
        buf = C.create_string_buffer(64)
        self._check(self._lib.velo_comm_peer_export_records(self._h, int(max_queries), buf))
        return buf.raw

    def comm_peer_attach_records(self, handles, max_queries: int):
        blob = b"".join(bytes(h) for h in handles)
        self._check(self._lib.velo_comm_peer_attach_records(self._h, C.create_string_buffer(blob, len(blob)), int(max_queries)))

    def set_residual_stats(self, enable: bool = True):
        """residualStats after every f2f iteration into the summary (velo.h:909)."""
        self._check(self._lib.velo_set_residual_stats(self._h, 1 if enable else 0))

    def residual_stats(self, x):
        """residualStats (velo.h:921-1025) of the current blocks at x."""
        xx = np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(6))
        out = VeloResidualStats()
        self._check(self._lib.velo_residual_stats_at(self._h, _ptr(xx), C.byref(out)))
        return out

    def chain_stats(self):
        """(calls, misses) of the one-chain-per-call mode (velo_chain_stats): misses were repeated host-driven, same results."""
        a, b = C.c_int32(0), C.c_int32(0)
        self._check(self._lib.velo_chain_stats(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def comm_info(self):
        """(kind, rank, world): kind 0 = none, 1 = RCCL (world read back from the communicator), 2 = peer slabs."""
        k, r, w = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        self._check(self._lib.velo_comm_info(self._h, C.byref(k), C.byref(r), C.byref(w)))
        return k.value, r.value, w.value


class ScanCache:
    """Device-resident scan cache (velo_cache_*): the reference's ScansLRU (lru.h:31-61, 50 scans) with the scans and their
    search index kept in HBM.  store() copies the scan a context holds; load() hands it to any context as target or source."""

    def __init__(self, device: int = 0, capacity: int = 50, lib: Optional[C.CDLL] = None):
        self._lib = lib if lib is not None else load_library()
        self._h = C.c_void_p()
        status = self._lib.velo_cache_create(C.byref(self._h), int(device), int(capacity))
        if status != 0:
            msg = self._lib.velo_last_error()
            raise VeloError(f"velo status {status}: {msg.decode() if msg else ''}")

    def _check(self, status: int):
        if status != 0:
            msg = self._lib.velo_last_error()
            raise VeloError(f"velo status {status}: {msg.decode() if msg else ''}")

    def close(self):
        if self._h:
            self._lib.velo_cache_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def store(self, frame: int, ctx: "Context", of_target: bool):
        self._check(self._lib.velo_cache_store(self._h, int(frame), ctx.handle, int(bool(of_target))))

    def load(self, frame: int, ctx: "Context", as_target: bool):
        self._check(self._lib.velo_cache_load(self._h, int(frame), ctx.handle, int(bool(as_target))))

    def __contains__(self, frame: int) -> bool:
        return bool(self._lib.velo_cache_contains(self._h, int(frame)))

    def frames(self):
        """cached frame numbers, most recently used first"""
        n = self._lib.velo_cache_frames(self._h, None, 0)
        out = (C.c_int32 * max(n, 1))()
        n = self._lib.velo_cache_frames(self._h, out, n)
        return [int(out[i]) for i in range(n)]


def comm_unique_id() -> bytes:
    lib = load_library()
    buf = C.create_string_buffer(128)
    st = lib.velo_comm_unique_id(buf)
    if st != 0:
        raise VeloError(f"velo_comm_unique_id failed: {st}")
    return buf.raw


def frame_to_frame_batch(ctxs, x0s):
    """B independent scan pairs in flight, one context each.  The library advances them in lock-step on one stream with shared LM
    launches when it can (same device and parameters, no communicator, no visual blocks), else with a host thread per context."""
    lib = ctxs[0]._lib if len(ctxs) else load_library()      # the build the contexts were created on
    n = len(ctxs)
    arr = (_ctx * n)(*[c.handle for c in ctxs])
    x = np.ascontiguousarray(np.asarray(x0s, dtype=np.float64).reshape(n, 6)).copy()
    T = np.zeros((n, 16))
    S = (VeloSummary * n)()
    st = lib.velo_frame_to_frame_batch(arr, n, _ptr(x), _ptr(T), S)
    if st != 0:
        msg = lib.velo_last_error()
        raise VeloError(f"velo status {st}: {msg.decode() if msg else ''}")
    return x, T.reshape(n, 4, 4), list(S)


SCAN_ON_DEVICE, SCAN_SHARED, SCAN_PROMOTE = 1, 2, 4


def scan_refs(scans, device=None, shared=False):
    """[(xyz, ring_offsets), ...] -> (velo_scan_ref array, objects to keep alive while it is in use).  xyz: numpy (n,3|4) float32 or a
    torch tensor on the GPU, as for Context.set_target (device: the contexts' device, checked against the tensors').  Device tensors
    must not be written by torch between this call and the registration that uses the descriptors."""
    n = len(scans)
    arr = (VeloScanRef * n)()
    keep = []
    seen = {}                       # the same (cloud, offsets) objects give the same descriptor (what SCAN_SHARED keys on)
    for i, (xyz, off) in enumerate(scans):
        key = (id(xyz), id(off))
        if key not in seen:
            seen[key] = Context._cloud_args(xyz, off, device)
        ptr, stride, off_a, dev, k = seen[key]
        arr[i].xyz = ptr.value
        arr[i].stride_bytes = stride
        arr[i].ring_offsets = off_a.ctypes.data
        arr[i].n_rings = len(off_a) - 1
        arr[i].on_device = dev | (SCAN_SHARED if shared else 0)
        keep.append((k, off_a))
    return arr, keep


def promote_refs(n: int):
    """n target descriptors that say "this job's target is the scan the context holds as its source" (VELO_SCAN_PROMOTE): the step of a
    drive, where frame k -- registered as source in the last call -- is the target of frame k+1 (main.cpp:233,380)."""
    arr = (VeloScanRef * n)()
    for i in range(n):
        arr[i].xyz = None
        arr[i].stride_bytes = 16
        arr[i].ring_offsets = None
        arr[i].n_rings = 0
        arr[i].on_device = SCAN_ON_DEVICE | SCAN_PROMOTE
    return arr, []


def visual_refs(matches_per_job):
    """[matches of job 0, ...] (structured arrays / dicts / None) -> (pointer array, count array, objects to keep alive) for
    register_batch(..., visual=...)."""
    n = len(matches_per_job)
    ptrs = (C.c_void_p * n)()
    cnt = (C.c_int32 * n)()
    keep = []
    for i, m in enumerate(matches_per_job):
        if isinstance(m, dict):
            m = matches_from_dict(m)
        a = np.ascontiguousarray(m, dtype=MATCH_DTYPE) if m is not None else np.zeros(0, MATCH_DTYPE)
        ptrs[i] = a.ctypes.data if len(a) else None
        cnt[i] = len(a)
        keep.append(a)
    return ptrs, cnt, keep


def register_batch(ctxs, targets, sources, x0s, refs=None, visual=None):
    """velo_register_batch: job i's target / source scans go into context i and the batch is registered, all inside the library
    (the index builds run on the threads that drive the groups).  targets / sources: lists of (xyz, ring_offsets) or None to keep
    what the contexts hold; refs = (target_refs, source_refs) from scan_refs() to reuse prepared descriptors across calls;
    visual = visual_refs(...) hands the jobs' matches over in the same call (velo_register_batch_visual)."""
    lib = ctxs[0]._lib if len(ctxs) else load_library()      # the build the contexts were created on
    n = len(ctxs)
    arr = (_ctx * n)(*[c.handle for c in ctxs])
    if refs is None:
        dev = ctxs[0]._device if n else None
        tr = scan_refs(targets, dev) if targets is not None else (None, None)
        sr = scan_refs(sources, dev) if sources is not None else (None, None)
    else:
        tr, sr = refs
    x = np.ascontiguousarray(np.asarray(x0s, dtype=np.float64).reshape(n, 6)).copy()
    T = np.zeros((n, 16))
    S = (VeloSummary * n)()
    tp = C.cast(tr[0], C.c_void_p) if tr[0] is not None else None
    sp = C.cast(sr[0], C.c_void_p) if sr[0] is not None else None
    if visual is not None:
        st = lib.velo_register_batch_visual(arr, n, tp, sp, C.cast(visual[0], C.c_void_p), C.cast(visual[1], C.c_void_p), _ptr(x), _ptr(T), S)
    else:
        st = lib.velo_register_batch(arr, n, tp, sp, _ptr(x), _ptr(T), S)
    if st != 0:
        msg = lib.velo_last_error()
        raise VeloError(f"velo status {st}: {msg.decode() if msg else ''}")
    return x, T.reshape(n, 4, 4), list(S)


def hint_next_sources(ctxs, refs):
    """velo_hint_next_source for every context: refs = scan_refs(...)[0] of the clouds the NEXT call will hand over as sources (host clouds are
    uploaded under the current call's launches; device clouds ignore the hint)."""
    lib = ctxs[0]._lib if len(ctxs) else load_library()
    for i, c in enumerate(ctxs):
        st = lib.velo_hint_next_source(c.handle, C.addressof(refs[i]))
        if st != 0:
            msg = lib.velo_last_error()
            raise VeloError(f"velo status {st}: {msg.decode() if msg else ''}")


def hint_next_frames(ctxs, refs):
    """velo_hint_next_frame for every context: refs = scan_refs(...)[0] of the frames the NEXT call will bring as sources behind a promotion
    (the step of a drive): their loads are enqueued behind the current call's launches."""
    lib = ctxs[0]._lib if len(ctxs) else load_library()
    for i, c in enumerate(ctxs):
        st = lib.velo_hint_next_frame(c.handle, C.addressof(refs[i]))
        if st != 0:
            msg = lib.velo_last_error()
            raise VeloError(f"velo status {st}: {msg.decode() if msg else ''}")


def sequence_refs(frames_per_drive, device=None, first=1, count=None):
    """Descriptors of the frames a register_sequences call walks: frames_per_drive[i] = [(xyz, ring_offsets), ...] of drive i; frames
    first .. first + count - 1 of every drive, laid out [frame][drive] as velo_register_sequences takes them.  -> (array, keep-alive)"""
    n = len(frames_per_drive)
    count = (min(len(f) for f in frames_per_drive) - first) if count is None else count
    flat = [frames_per_drive[i][first + f] for f in range(count) for i in range(n)]
    arr, keep = scan_refs(flat, device)
    return arr, keep, count


def sequence_visual_refs(matches_per_drive, first=0, count=None):
    """matches_per_drive[i][k] = the matches of drive i's pair k (frame k+1 against frame k) -> pointer / count arrays laid out [frame][drive]"""
    n = len(matches_per_drive)
    count = (min(len(m) for m in matches_per_drive) - first) if count is None else count
    return visual_refs([matches_per_drive[i][first + f] for f in range(count) for i in range(n)])


SEQ_LOCKSTEP, SEQ_ANNOUNCE = 1, 2


def register_sequences(ctxs, frame_refs, n_frames, poses, x_guess, visual=None, summaries=True, lockstep=False):
    """velo_register_sequences: the drive loop of len(ctxs) sequences for n_frames frames in one call (every context holds its drive's
    current frame as source).  frame_refs: sequence_refs(...)[0]; poses (n,4,4) and x_guess (n,6) float64 C-contiguous, UPDATED IN PLACE
    (the drives' accumulated poses and the next frame's constant-velocity guesses).  lockstep: the groups start every frame together
    (VELO_SEQ_LOCKSTEP).  -> xs (F,n,6), Ts (F,n,4,4), summaries [F][n] or None"""
    lib = ctxs[0]._lib if len(ctxs) else load_library()
    n = len(ctxs)
    assert poses.dtype == np.float64 and poses.flags.c_contiguous and poses.shape == (n, 4, 4)
    assert x_guess.dtype == np.float64 and x_guess.flags.c_contiguous and x_guess.shape == (n, 6)
    arr = (_ctx * n)(*[c.handle for c in ctxs])
    xs = np.zeros((n_frames, n, 6))
    Ts = np.zeros((n_frames, n, 16))
    S = (VeloSummary * (n * n_frames))() if summaries else None
    vm = C.cast(visual[0], C.c_void_p) if visual is not None else None
    vn = C.cast(visual[1], C.c_void_p) if visual is not None else None
    st = lib.velo_register_sequences(arr, n, n_frames, C.cast(frame_refs, C.c_void_p), vm, vn, _ptr(poses), _ptr(x_guess), _ptr(xs), _ptr(Ts), S,
                                     SEQ_LOCKSTEP if lockstep else 0)
    if st != 0:
        msg = lib.velo_last_error()
        raise VeloError(f"velo status {st}: {msg.decode() if msg else ''}")
    return xs, Ts.reshape(n_frames, n, 4, 4), ([[S[f * n + i] for i in range(n)] for f in range(n_frames)] if summaries else None)


class DriveStep:
    """The step of len(ctxs) drives as ONE C call with every argument prepared once: velo_register_sequences for one frame, the frame after it
    announced (VELO_SEQ_ANNOUNCE) -- promote, load, register, chain the pose, predict the motion (main.cpp:305-413 for n sequences).  What a
    compiled caller's loop body costs on the host; register_batch + pose_handoff build a dozen numpy / ctypes objects per step.
    poses (n,4,4) and x_guess (n,6): float64 C-contiguous, UPDATED IN PLACE by every step."""

    def __init__(self, ctxs, poses, x_guess):
        self._lib = ctxs[0]._lib
        self.n = n = len(ctxs)
        assert poses.dtype == np.float64 and poses.flags.c_contiguous and poses.shape == (n, 4, 4)
        assert x_guess.dtype == np.float64 and x_guess.flags.c_contiguous and x_guess.shape == (n, 6)
        self._keep = (ctxs, poses, x_guess)
        self._arr = (_ctx * n)(*[c.handle for c in ctxs])
        self._pp, self._xg = _ptr(poses), _ptr(x_guess)

    def outputs(self, n_steps):
        """result blocks for n_steps steps, allocated once: (xs (S,n,6), Ts (S,n,4,4), summaries [S] of VeloSummary * n, per-step pointers)"""
        n = self.n
        xs, Ts = np.zeros((n_steps, n, 6)), np.zeros((n_steps, n, 16))
        S = [(VeloSummary * n)() for _ in range(n_steps)]
        ptrs = [(C.cast(xs[k].ctypes.data, _dp), C.cast(Ts[k].ctypes.data, _dp)) for k in range(n_steps)]
        return xs, Ts.reshape(n_steps, n, 4, 4), S, ptrs

    def __call__(self, frames_ptr, out_ptrs, summaries, visual=None, announce=False):
        """frames_ptr: c_void_p of n (announce: 2 n) velo_scan_ref -- this step's frames, then the next step's; out_ptrs / summaries: one step's
        entries of outputs(); visual: (pointer array, count array) cast to c_void_p, or None"""
        st = self._lib.velo_register_sequences(self._arr, self.n, 1, frames_ptr, visual[0] if visual else None, visual[1] if visual else None, self._pp, self._xg,
                                               out_ptrs[0], out_ptrs[1], summaries, SEQ_LOCKSTEP | (SEQ_ANNOUNCE if announce else 0))
        if st != 0:
            msg = self._lib.velo_last_error()
            raise VeloError(f"velo status {st}: {msg.decode() if msg else ''}")


def pose_handoff(poses: np.ndarray, dpose: np.ndarray):
    """velo_pose_handoff: poses (n,4,4) float64 C-contiguous, UPDATED IN PLACE to poses @ dpose (main.cpp:408); returns the (n,6) guesses
    of the next frame, the 6-vectors of poses_old^-1 poses_new (main.cpp:311-331)."""
    lib = load_library()
    assert poses.dtype == np.float64 and poses.flags.c_contiguous and poses.shape[1:] == (4, 4)
    n = poses.shape[0]
    d = np.ascontiguousarray(np.asarray(dpose, dtype=np.float64).reshape(n, 16))
    x = np.zeros((n, 6))
    st = lib.velo_pose_handoff(n, _ptr(poses), _ptr(d), _ptr(x))
    if st != 0:
        msg = lib.velo_last_error()
        raise VeloError(f"velo status {st}: {msg.decode() if msg else ''}")
    return x


def pose_vec_to_mat(x) -> np.ndarray:
    lib = load_library()
    xv = _dvec(x, 6)
    T = np.zeros(16)
    lib.velo_pose_vec_to_mat(_ptr(xv), _ptr(T))
    return T.reshape(4, 4)


def pose_mat_to_vec(T) -> np.ndarray:
    lib = load_library()
    Tv = _dvec(T, 16)
    x = np.zeros(6)
    lib.velo_pose_mat_to_vec(_ptr(Tv), _ptr(x))
    return x
