"""Shared builders for the parity tests: the same seeded inputs go to the HIP path and to the oracle."""
from __future__ import annotations

import numpy as np

import velo_amd  # noqa: F401
from velo_amd import synth

CORR_INDEX_FIELDS = ("valid", "ring_i", "idx_i", "ring_j", "idx_j", "idx_k", "src_ring", "src_idx")


def small_pair(n_beams=16, n_azimuth=128, **kw):
    return synth.scan_pair(n_beams=n_beams, n_azimuth=n_azimuth, **kw)


def load_both(ctx, orc, d, visual=None, **params):
    if params:
        ctx.set_params(**params)
        orc.set_params(**params)
    ctx.set_target(d["tgt_xyz"], d["tgt_off"])
    orc.set_target(d["tgt_xyz"], d["tgt_off"])
    ctx.set_source(d["src_xyz"], d["src_off"])
    orc.set_source(d["src_xyz"], d["src_off"])
    if visual is not None:
        ctx.set_visual(visual)
        orc.set_visual(visual)


def assert_corr_equal(a: np.ndarray, b: np.ndarray):
    """Index-exact and bit-exact comparison of two correspondence tables."""
    assert len(a) == len(b)
    for f in CORR_INDEX_FIELDS:
        bad = np.nonzero(a[f] != b[f])[0]
        # ring/idx fields of j,k are only defined when both rings were found; compare where meaningful
        if f in ("idx_i",):
            bad = bad[(a["ring_i"][bad] >= 0)]
        if f in ("idx_j",):
            bad = bad[(a["ring_j"][bad] >= 0)]
        if f in ("idx_k",):
            bad = bad[(a["ring_i"][bad] >= 0) & (a["ring_j"][bad] >= 0)]
        assert bad.size == 0, f"{f} differs at {bad[:10]}: {a[f][bad[:10]]} vs {b[f][bad[:10]]}"
    v = a["valid"] == 1
    for f in ("p", "n", "v0"):
        assert np.array_equal(a[f][v].view(np.uint32), b[f][v].view(np.uint32)), f"{f} not bit-identical"
    has_i = a["ring_i"] >= 0
    assert np.array_equal(a["dist_i"][has_i].view(np.uint32), b["dist_i"][has_i].view(np.uint32))
    has_j = a["ring_j"] >= 0
    assert np.array_equal(a["dist_j"][has_j].view(np.uint32), b["dist_j"][has_j].view(np.uint32))


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def pose_close(x, y, t_tol=1e-4, r_tol=1e-5):
    """north_star tolerance: 1e-4 m translation, 1e-5 rad rotation."""
    x = np.asarray(x)
    y = np.asarray(y)
    return np.linalg.norm(x[3:] - y[3:]) <= t_tol and np.linalg.norm(x[:3] - y[:3]) <= r_tol
