"""Numbers on "parity unpinned" (DESIGN.md section 2): one CPU test per third-party behaviour the reference does not pin.

  (a) the LM iterate SEQUENCE of the oracle against a second, separately written restatement of SURVEY.md Appendix B1
      (tests/lm_independent.py: numpy duals, an SVD least-squares step on the augmented system, its own bookkeeping);
  (b) solver variants: the step by Householder QR in row space (Ceres' DENSE_QR) instead of the 6x6 Cholesky, and a function-tolerance
      exit that applies the converging step -- pose delta of the whole frame-to-frame call vs the default, against the
      north_star tolerance 1e-4 m / 1e-5 rad;
  (c) a census of exact float-distance ties (FLANN's traversal order decides them inside a ring, and is unpinned) and the pose delta
      when they go to the HIGHEST index instead of the lowest.
The full-size numbers for C1-C4 come from tools/parity_budget.py and are committed as profiles/r04_parity_budget.json; the tests
below re-derive the C1 / C2 ones that are affordable here and check the committed file against the claims DESIGN.md makes.
Reference call sites: velo.h:897-902 (ceres::Solve), velo.h:825-848 (per-ring nearest neighbour + strict '<')."""
from __future__ import annotations

import json
import os

import numpy as np
import pytest

import helpers as H
import lm_independent as LI
import oracle_lib as ol
from velo_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STATUS = {ol.TRACE_ACCEPTED: "accepted", ol.TRACE_REJECTED: "rejected", ol.TRACE_INVALID: "invalid",
          ol.TRACE_PARAMETER_TOL: "parameter", ol.TRACE_FUNCTION_TOL: "function", ol.TRACE_GRADIENT_TOL: "gradient"}
T_TOL, R_TOL = 1e-4, 1e-5           # north_star


def _oracle(d, skip=1, vis=None, threads=4, **params):
    o = ol.Oracle(threads=threads, icp_skip=skip, **params)
    o.set_target(d["tgt_xyz"], d["tgt_off"])
    o.set_source(d["src_xyz"], d["src_off"])
    if vis is not None:
        o.set_visual(vis)
    return o


def _compare_traces(o, x0, **lm_kw):
    """The oracle's solve at x0 and the independent one on the same blocks: identical decisions, numbers to 1e-7."""
    x, s, tr = o.solve_trace(x0)
    xi, label, tri, evaluations = LI.solve(o.blocks(), x0, **lm_kw)
    assert [STATUS[int(r["status"])] for r in tr] == [r["status"] for r in tri]
    assert evaluations == s.evaluations and len(tri) == s.lm_iterations
    for a, b in zip(tr, tri):
        assert a["iteration"] == b["iteration"]
        assert a["radius"] == pytest.approx(b["radius"], rel=1e-8)             # the radius schedule: a chain of functions of q
        assert a["cost"] == pytest.approx(b["cost"], rel=1e-11)
        assert a["model_change"] == pytest.approx(b["model_change"], rel=1e-6)  # differences of nearly equal numbers late in a solve
        if "step_norm" in b:
            assert a["step_norm"] == pytest.approx(b["step_norm"], rel=1e-6)
            assert a["candidate_cost"] == pytest.approx(b["candidate_cost"], rel=1e-11)
        if "relative_decrease" in b and int(a["status"]) != ol.TRACE_PARAMETER_TOL:
            # q = (cost - candidate cost) / model change: near the optimum the numerator is a difference of equal numbers
            noise = 256 * np.finfo(np.float64).eps * a["cost"] / abs(a["model_change"])
            assert a["relative_decrease"] == pytest.approx(b["relative_decrease"], rel=1e-5, abs=1e-9 + noise)
    assert np.max(np.abs(x - xi)) <= 1e-11
    return x, s, tr, label


@pytest.mark.parametrize("it", [1, 2])
def test_lm_sequence_lidar_only_matches_independent_restatement(it):
    d = H.small_pair()
    o = _oracle(d)
    x0 = np.array(d["x0"] if it == 1 else d["x_true"], dtype=np.float64)
    assert o.associate(x0, it) > 300
    _x, s, tr, label = _compare_traces(o, x0)
    assert label == "function" and s.termination == 0 and len(tr) >= 3


@pytest.mark.parametrize("mix", ["all", "reproj"])
def test_lm_sequence_with_visual_blocks_matches_independent_restatement(mix):
    """All four visual functors (R2-R5), Arctan / scaled losses and the point-to-plane blocks in one problem."""
    d = H.small_pair()
    vis = synth.stereo_matches(100, mix=mix, outlier_frac=0.3, x_true=d["x_true"]) if mix == "all" \
        else synth.stereo_matches(100, outlier_frac=0.1, x_true=d["x_true"])
    o = _oracle(d, vis=vis)
    x0 = np.array(d["x0"], dtype=np.float64)
    assert o.build_visual(x0, 1) > 100
    o.associate(x0, 1)
    kinds = set(int(k) for k in o.blocks()["kind"])
    assert kinds >= ({0, 1, 2, 3, 4} if mix == "all" else {1, 2, 4})
    _compare_traces(o, x0)
    far = x0 + np.array([0.05, -0.05, 0.1, 0.5, -0.3, 0.4])
    o.build_visual(far, 1)
    o.associate(far, 1)
    _compare_traces(o, far)


def test_lm_sequence_with_rejected_steps_matches_independent_restatement():
    """A start far outside the basin: the radius shrinks by 2, 4, ... and the LM diagonal is reused after every rejection."""
    d = H.small_pair()
    o = _oracle(d)
    x0 = np.array(d["x0"]) + np.array([-0.095, 0.278, 0.07, -0.711, -2.916, -1.962])
    o.associate(x0, 1)
    _x, _s, tr, _label = _compare_traces(o, x0)
    st = [int(r["status"]) for r in tr]
    assert st.count(ol.TRACE_REJECTED) >= 3 and st[0] == ol.TRACE_REJECTED and st[1] == ol.TRACE_REJECTED
    # two rejections in a row: radius / 2, then / 4 (decrease factor doubles), back to 2 after an accepted step
    assert tr["radius"][1] == tr["radius"][0] / 2 and tr["radius"][2] == tr["radius"][1] / 4


def test_lm_each_tolerance_ends_a_solve_in_both_restatements():
    d = H.small_pair()
    x0 = np.array(d["x0"], dtype=np.float64)
    # function tolerance: the default exit
    o = _oracle(d)
    o.associate(x0, 1)
    assert _compare_traces(o, x0)[3] == "function"
    # parameter tolerance: function tolerance off
    o = _oracle(d, function_tolerance=0.0)
    o.associate(x0, 1)
    _x, _s, tr, label = _compare_traces(o, x0, f_tol=0.0)
    assert label == "parameter" and int(tr["status"][-1]) == ol.TRACE_PARAMETER_TOL
    # gradient tolerance: the other two off, gradient bar raised to where this problem reaches it
    o = _oracle(d, function_tolerance=0.0, parameter_tolerance=0.0, gradient_tolerance=1e-3)
    o.associate(x0, 1)
    _x, _s, tr, label = _compare_traces(o, x0, f_tol=0.0, p_tol=0.0, g_tol=1e-3)
    assert label == "gradient" and int(tr["status"][-1]) == ol.TRACE_GRADIENT_TOL
    # iteration cap
    o = _oracle(d, function_tolerance=0.0, parameter_tolerance=0.0, gradient_tolerance=0.0, max_num_iterations=4)
    o.associate(x0, 1)
    _x, s, _tr, label = _compare_traces(o, x0, f_tol=0.0, p_tol=0.0, g_tol=0.0, max_iterations=4)
    assert label == "max_iterations" and s.termination == 1 and s.lm_iterations == 4


def test_whole_call_driven_by_the_independent_solver_lands_on_the_oracle_pose():
    """frameToFrame's six rounds with the oracle's association but the INDEPENDENT solver: same pose, same evaluation counts."""
    d = H.small_pair()
    vis = synth.stereo_matches(60, outlier_frac=0.1, x_true=d["x_true"])
    xo, _T, so = _oracle(d, vis=vis).frame_to_frame(d["x0"])
    o = _oracle(d, vis=vis)
    x = np.array(d["x0"], dtype=np.float64)
    counts = []
    for it in (1, 2):
        o.build_visual(x, it)
        for _ in range(3):
            o.associate(x, it)
            x, _label, _tr, ev = LI.solve(o.blocks(), x)
            counts.append(ev)
    assert counts == [so.solves[i].evaluations for i in range(6)]
    assert np.linalg.norm(x[3:] - xo[3:]) <= 1e-9 and np.linalg.norm(x[:3] - xo[:3]) <= 1e-10


# ------------------------------------------------------------------------------------------------------------------------
# (b) solver variants
# ------------------------------------------------------------------------------------------------------------------------
def _pose_delta(d, skip, vis, **variant):
    x0, _T, s0 = _oracle(d, skip, vis, threads=ol.max_threads()).frame_to_frame(d["x0"])
    o = _oracle(d, skip, vis, threads=ol.max_threads())
    o.set_variant(**variant)
    x, _T, s = o.frame_to_frame(d["x0"])
    same = [s.solves[i].evaluations for i in range(6)] == [s0.solves[i].evaluations for i in range(6)]
    return float(np.linalg.norm(x[3:] - x0[3:])), float(np.linalg.norm(x[:3] - x0[:3])), same


def test_qr_step_is_not_observable():
    """Cholesky on the normal equations vs Householder QR of [J; D]: 1e-12 of the tolerance on mini, C1 and C2."""
    mini = H.small_pair()
    vis = synth.stereo_matches(60, mix="all", outlier_frac=0.2, x_true=mini["x_true"])
    full = synth.scan_pair()
    for d, skip, v in ((mini, 1, None), (mini, 1, vis), (full, 200, None), (full, 1, None)):
        dt, dw, same = _pose_delta(d, skip, v, qr=True)
        assert same and dt <= 1e-12 and dw <= 1e-13, (skip, dt, dw)


def test_function_tolerance_variant_budget():
    """If a Ceres version APPLIED the step that meets the function tolerance (the default restatement does not: the check precedes
    the update in trust_region_minimizer.cc as recalled in SURVEY B1), every solve would end one step further.  Measured on the
    whole call: inside the north_star tolerance at icp_skip = 1 (C2), OUTSIDE it at the reference's icp_skip = 200 (C1: 640 rows,
    a flatter problem) -- DESIGN.md section 2 says so."""
    full = synth.scan_pair()
    dt2, dw2, same2 = _pose_delta(full, 1, None, ftol_apply=True)
    assert same2 and dt2 <= 0.2 * T_TOL and dw2 <= 0.5 * R_TOL, (dt2, dw2)
    dt1, dw1, _same1 = _pose_delta(full, 200, None, ftol_apply=True)
    assert T_TOL < dt1 <= 3 * T_TOL and R_TOL < dw1 <= 5 * R_TOL, (dt1, dw1)        # observable: 1.8e-4 m / 3.0e-5 rad
    mini = H.small_pair()
    dtm, dwm, _ = _pose_delta(mini, 1, None, ftol_apply=True)
    assert dtm <= 5 * T_TOL and dwm <= 20 * R_TOL                                    # the 2k-point pair is flatter still


# ------------------------------------------------------------------------------------------------------------------------
# (c) ties
# ------------------------------------------------------------------------------------------------------------------------
def _rounds(d, skip, vis=None):
    o = _oracle(d, skip, vis, threads=ol.max_threads())
    x = np.array(d["x0"], dtype=np.float64)
    out = []
    for it in (1, 2):
        o.build_visual(x, it)
        for _ in range(3):
            out.append(o.tie_census(x, it))
            o.associate(x, it)
            x, _s = o.solve(x)
    return out


def test_tie_census_counts_a_planted_tie():
    """Two points of one ring mirrored about the query: the census sees the tie, the default takes the lower index, the variant the higher."""
    ring0 = np.array([[1.0, 0.25, 0.0], [1.0, -0.25, 0.0], [3.0, 0.0, 0.0]], dtype=np.float32)
    ring1 = np.array([[1.0, 0.0, 0.5], [2.0, 0.0, 0.5], [3.0, 0.0, 0.5]], dtype=np.float32)
    tgt = np.vstack([ring0, ring1])
    off = np.array([0, 3, 6], dtype=np.int32)
    src = np.array([[1.0, 0.0, 0.0]], dtype=np.float32)
    soff = np.array([0, 1], dtype=np.int32)
    o = ol.Oracle(icp_skip=1)
    o.set_target(tgt, off)
    o.set_source(src, soff)
    c = o.tie_census(np.zeros(6), 1)
    assert c["queries"] == 1 and c["gated_ring_pairs"] == 2 and c["in_ring_ties"] == 1 and c["in_ring_ties_on_a_winner"] == 1
    o.associate(np.zeros(6), 1)
    assert o.correspondences()["idx_i"][0] == 0
    o.set_variant(tie_high=True)
    o.associate(np.zeros(6), 1)
    assert o.correspondences()["idx_i"][0] == 1


def test_tie_census_c1_has_no_ties_and_c2_ties_stay_inside_tolerance():
    full = synth.scan_pair()
    c1 = _rounds(full, 200)
    assert sum(r["in_ring_ties"] for r in c1) == 0 and sum(r["cross_ring_tie_first"] + r["cross_ring_tie_second"] for r in c1) == 0
    c2 = _rounds(full, 1)
    gated = sum(r["gated_ring_pairs"] for r in c2)
    ties = sum(r["in_ring_ties"] for r in c2)
    on_winner = sum(r["in_ring_ties_on_a_winner"] for r in c2)
    assert gated > 8_000_000 and ties <= 50 and on_winner <= 3            # measured: 22 of 8.36 M pairs, 1 on a winning ring
    assert sum(r["cross_ring_tie_first"] + r["cross_ring_tie_second"] for r in c2) == 0
    dt, dw, same = _pose_delta(full, 1, None, tie_high=True)
    assert same and dt <= 0.01 * T_TOL and dw <= 0.05 * R_TOL, (dt, dw)  # measured 4.5e-7 m / 1.7e-7 rad


def test_plane_normal_arithmetic_variants_are_not_observable():
    """velo.h:868-874 in Eigen's float arithmetic is third-party: (a) Vector3f::norm() through the unrolled redux, which splits three
    elements as 1 + 2 -- x^2 + (y^2 + z^2) -- where the restatement and the kernel sum (x^2 + y^2) + z^2; (b) a reference built with FMA
    contraction would round a*b - c*d of the cross product once less.  Both feed the 1e-5 skip (velo.h:873) and the last bit of every
    normal.  Measured on the whole call: 1e-10 m / 1e-10 rad at most (six orders inside the tolerance), the same evaluation counts, the
    same rows per solve, the same number of ||N|| skips."""
    mini = H.small_pair()
    full = synth.scan_pair()
    for d, skip in ((mini, 1), (full, 200), (full, 1)):
        o0 = _oracle(d, skip, None, threads=ol.max_threads())
        x0, _T0, s0 = o0.frame_to_frame(d["x0"])
        k0 = o0.set_variant_normal()
        for normal in (dict(norm_split=True), dict(cross_fma=True), dict(norm_split=True, cross_fma=True)):
            o = _oracle(d, skip, None, threads=ol.max_threads())
            o.set_variant_normal(**normal)
            x, _T, s = o.frame_to_frame(d["x0"])
            k = o.set_variant_normal(**normal)
            assert [s.solves[i].evaluations for i in range(6)] == [s0.solves[i].evaluations for i in range(6)]
            assert [s.solves[i].n_icp_valid for i in range(6)] == [s0.solves[i].n_icp_valid for i in range(6)] and k == k0
            assert np.linalg.norm(x[3:] - x0[3:]) <= 1e-4 * T_TOL and np.linalg.norm(x[:3] - x0[:3]) <= 1e-4 * R_TOL, (skip, normal, x - x0)
    # the variants DO change bits: some normal of the mini pair differs in its last place (otherwise the rows above would measure nothing)
    o0 = _oracle(mini, 1, None, threads=1); o0.associate(mini["x0"], 1); n0 = o0.correspondences()["n"].copy()
    o1 = _oracle(mini, 1, None, threads=1); o1.set_variant_normal(norm_split=True, cross_fma=True); o1.associate(mini["x0"], 1)
    n1 = o1.correspondences()["n"]
    assert not np.array_equal(n0, n1) and np.abs(n0 - n1).max() <= 2e-6


def test_committed_budget_file_supports_the_design_table():
    """profiles/r05_parity_budget.json (tools/parity_budget.py, all of C1-C4 at full size) says what DESIGN.md section 2 claims."""
    path = os.path.join(ROOT, "profiles", "r05_parity_budget.json")
    B = json.load(open(path))
    assert set(B) >= {"c1", "c2", "c3", "c4"}
    for name, b in B.items():
        assert len(b["rounds"]) == 6
        v = b["variants"]
        assert v["qr"]["dt_m"] <= 1e-12 and v["qr"]["dw_rad"] <= 1e-13 and v["qr"]["same_evaluation_counts"]
        assert v["tie_high"]["dt_m"] <= 0.01 * T_TOL and v["tie_high"]["dw_rad"] <= 0.05 * R_TOL
        assert b["total"]["cross_ring_tie_first"] == 0 and b["total"]["cross_ring_tie_second"] == 0
        inside = v["ftol_apply"]["dt_m"] <= T_TOL and v["ftol_apply"]["dw_rad"] <= R_TOL
        assert inside == (name != "c1"), (name, v["ftol_apply"])
        for row in ("norm_split", "cross_fma"):                  # the plane normal's float arithmetic (velo.h:868-874): six orders inside the tolerance
            assert v[row]["dt_m"] <= 1e-4 * T_TOL and v[row]["dw_rad"] <= 1e-4 * R_TOL and v[row]["same_evaluation_counts"] and v[row]["same_valid_counts"]
            assert v[row]["norm_skips"] == v[row]["default_norm_skips"]
    assert B["c1"]["total"]["in_ring_ties"] == 0
