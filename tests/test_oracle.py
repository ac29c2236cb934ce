"""CPU tests (-m "not gpu") of the oracle: the restatement is checked against INDEPENDENT tools -- scipy
Rotation / cKDTree / least_squares, finite differences, closed forms -- because the reference ships no tests or
fixtures to pin it ("parity unpinned", see oracle/velo_oracle.cpp header)."""
import numpy as np
import pytest
from scipy.optimize import least_squares
from scipy.spatial import cKDTree
from scipy.spatial.transform import Rotation

import helpers as H
import oracle_lib as ol
from velo_amd import synth

RNG = np.random.default_rng(12345)


# ---- rotation (ceres::AngleAxisRotatePoint restated) ------------------------------------------------------
@pytest.mark.parametrize("scale", [1.0, 1e-3, 1e-7, 3.0])
def test_rotate_point_matches_scipy(scale):
    for _ in range(20):
        w = RNG.normal(size=3) * scale
        p = RNG.normal(size=3) * 10
        np.testing.assert_allclose(ol.rotate_point(w, p), Rotation.from_rotvec(w).apply(p), rtol=1e-13, atol=1e-13)


def test_rotate_point_small_angle_branch():
    w = np.array([1e-9, -2e-9, 3e-9])          # theta^2 < DBL_EPSILON -> p + w x p
    p = np.array([1.0, 2.0, 3.0])
    np.testing.assert_array_equal(ol.rotate_point(w, p), p + np.cross(w, p))
    assert np.array_equal(ol.rotate_point(np.zeros(3), p), p)


def test_transform_point_rounds_to_float_once():
    x = np.array([0.01, -0.02, 0.015, 0.3, -0.1, 1.0])
    for _ in range(50):
        p = (RNG.normal(size=3) * 20).astype(np.float32)
        want = (Rotation.from_rotvec(x[:3]).apply(p.astype(np.float64)) + x[3:])
        got = ol.transform_point(p, x)
        assert got.dtype == np.float32
        # float(double result): at most one float ulp away from the scipy double rounded the same way
        assert np.all(np.abs(got.astype(np.float64) - want) <= np.spacing(np.abs(want).astype(np.float32)) * 0.51 + 1e-12)


def test_pose_conversions_roundtrip_and_convention():
    for _ in range(20):
        x = np.concatenate([RNG.normal(size=3) * 0.5, RNG.normal(size=3)])
        T = ol.pose_vec_to_mat(x)
        np.testing.assert_allclose(T[:3, :3], Rotation.from_rotvec(x[:3]).as_matrix(), atol=1e-14)
        np.testing.assert_array_equal(T[:3, 3], x[3:])
        np.testing.assert_array_equal(T[3], [0, 0, 0, 1])
        np.testing.assert_allclose(ol.pose_mat_to_vec(T), x, atol=1e-13)
    # near pi the quaternion route must still invert
    x = np.array([3.1, 0.05, -0.02, 1, 2, 3.0])
    np.testing.assert_allclose(ol.pose_mat_to_vec(ol.pose_vec_to_mat(x)), x, atol=1e-10)
    np.testing.assert_allclose(synth.matrix_to_rotvec(synth.rotvec_to_matrix(x[:3])), x[:3], atol=1e-10)


# ---- residual functors: closed forms + finite-difference Jacobians -------------------------------------------
def _closed_form(kind, c, x):
    R = Rotation.from_rotvec(x[:3])
    t = x[3:]
    if kind == 4:      # cost3DPD costfunctions.h:39-54
        return np.array([np.dot(R.apply(c[0:3]) + t - c[6:9], c[3:6])])
    if kind == 0:      # cost3D3D :76-87
        return R.apply(c[0:3]) + t - c[3:6]
    if kind == 1:      # cost3D2D :111-126
        M = R.apply(c[0:3]) + t + c[5:8]
        return np.array([M[0] - c[3] * M[2], M[1] - c[4] * M[2]])
    if kind == 2:      # cost2D3D :151-168
        M = R.inv().apply(c[0:3] - t) + c[5:8]
        return np.array([M[0] - c[3] * M[2], M[1] - c[4] * M[2]])
    if kind == 3:      # cost2D2D :192-216
        M = R.apply([c[0], c[1], 1.0])
        tc = c[4:7]
        tt = -R.apply(tc) + t + tc
        tt = tt / np.linalg.norm(tt)
        sx, sy = c[2], c[3]
        return np.array([M[0] * (-sy * tt[2] + tt[1]) + M[1] * (sx * tt[2] - tt[0]) + M[2] * (-sx * tt[1] + sy * tt[0])])
    raise ValueError(kind)


N_CONST = {0: 6, 1: 8, 2: 8, 3: 7, 4: 9}


@pytest.mark.parametrize("kind", [0, 1, 2, 3, 4])
def test_functor_values_and_jacobians(kind):
    for trial in range(10):
        c = RNG.normal(size=N_CONST[kind]) * (3.0 if kind != 3 else 0.3)
        if kind == 3:
            c[4:7] = [-0.537, 0.0, 0.0]
        x = np.concatenate([RNG.normal(size=3) * (0.2 if trial else 1e-10), RNG.normal(size=3)])
        r, J = ol.functor(kind, c, x)
        np.testing.assert_allclose(r, _closed_form(kind, c, x), rtol=1e-11, atol=1e-12)
        if trial == 0:
            continue                         # finite differences across the small-angle branch switch are meaningless
        Jn = np.zeros_like(J)
        for k in range(6):
            h = 1e-6
            xp, xm = x.copy(), x.copy()
            xp[k] += h
            xm[k] -= h
            Jn[:, k] = (_closed_form(kind, c, xp) - _closed_form(kind, c, xm)) / (2 * h)
        np.testing.assert_allclose(J, Jn, rtol=2e-6, atol=2e-7)


def test_cost3dpd_jacobian_at_identity_is_p_cross_n_and_n():
    # at omega = 0, d r / d omega = p x N, d r / d t = N
    c = np.array([1, 2, 3, 0, 0, 1, 0.5, 0.5, 0.5], dtype=np.float64)
    r, J = ol.functor(4, c, np.zeros(6))
    np.testing.assert_allclose(J[0], [2, -1, 0, 0, 0, 1], atol=1e-15)
    assert abs(r[0] - 2.5) < 1e-15


# ---- losses (ceres CauchyLoss / ArctanLoss / ScaledLoss) ----------------------------------------------------------
def test_losses_closed_form_and_derivatives():
    for a, w in ((0.1, 1.0), (0.01, 10.0), (2e-5, 500.0)):
        for s in (0.0, 1e-9, 1e-4, 0.3, 7.0):
            rho = ol.loss(1, a, w, s)
            b = a * a
            sm = 1.0 + s * (1.0 / b)                               # ceres forms 1 + s*c and takes log(sum), not log1p
            np.testing.assert_allclose(rho, [w * b * np.log(sm), w / sm, -w / b / sm ** 2], rtol=1e-12)
            rho = ol.loss(2, a, w, s)
            np.testing.assert_allclose(rho[:2], [w * a * np.arctan2(s, a), w / (1 + s * s / (a * a))], rtol=1e-12)
            assert rho[2] <= 0.0
            for typ, scale in ((1, b), (2, a)):              # central difference, step small against the loss scale
                h = 1e-5 * (s + scale)
                if s - h < 0:
                    continue
                d = (ol.loss(typ, a, w, s + h)[0] - ol.loss(typ, a, w, s - h)[0]) / (2 * h)
                assert abs(d - ol.loss(typ, a, w, s)[1]) <= 1e-6 * abs(d) + 1e-12
    # scipy's 'cauchy' rho(z) = ln(1+z) with f_scale = a is ceres CauchyLoss(a) up to the factor b
    np.testing.assert_allclose(ol.loss(1, 0.1, 1.0, 0.02)[0], 0.01 * np.log1p(0.02 / 0.01))


# ---- exact per-ring 1-NN ------------------------------------------------------------------------------------------
def test_ring_kdtree_is_exact_against_bruteforce_and_ckdtree():
    d = H.small_pair(16, 256)
    orc = ol.Oracle()
    orc.set_target(d["tgt_xyz"], d["tgt_off"])
    qs = np.concatenate([d["src_xyz"][::37], (RNG.normal(size=(30, 3)) * 20).astype(np.float32)])
    for ring in range(0, 16, 3):
        pts = d["tgt_xyz"][d["tgt_off"][ring]:d["tgt_off"][ring + 1]]
        tree = cKDTree(pts.astype(np.float64))
        for q in qs:
            found, it, dt, ib, db = orc.ring_nn(ring, q)
            assert found == 1 and it == ib and dt == db                  # tree == brute force, bit-exact float distance
            dd, ii = tree.query(q.astype(np.float64))
            assert abs(np.sqrt(dt) - dd) <= 1e-5 * max(dd, 1.0)
            if ii != it:                                                  # only float-level near ties may differ
                assert abs(np.sum((pts[ii].astype(np.float64) - q) ** 2) - dt) <= 1e-5 * max(dt, 1e-12)


def test_ring_nn_ties_go_to_lowest_index():
    pts = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0]] * 8, dtype=np.float32)   # 32 points, many exact ties
    orc = ol.Oracle()
    orc.set_target(pts, np.array([0, 32], dtype=np.int32))
    found, it, dt, ib, db = orc.ring_nn(0, np.zeros(3, dtype=np.float32))
    assert (found, it, ib, dt) == (1, 0, 0, 1.0)


# ---- association semantics (velo.h:806-874) ----------------------------------------------------------------------------
def _associate_numpy(d, x, iter_, skip, gate0=0.5, norm_cond=1e-5):
    """Independent numpy restatement (per-ring brute force) used to cross-check the C++ oracle."""
    out = []
    gate = gate0 / iter_ ** 4
    for sm in range(len(d["src_off"]) - 1):
        ring_pts = d["src_xyz"][d["src_off"][sm]:d["src_off"][sm + 1]]
        for smi in range(0, len(ring_pts), skip):
            q = ol.transform_point(ring_pts[smi], x)
            best = []
            for ss in range(len(d["tgt_off"]) - 1):
                P = d["tgt_xyz"][d["tgt_off"][ss]:d["tgt_off"][ss + 1]]
                dx = (q[0] - P[:, 0]).astype(np.float32)
                dy = (q[1] - P[:, 1]).astype(np.float32)
                dz = (q[2] - P[:, 2]).astype(np.float32)
                d2 = ((dx * dx + dy * dy).astype(np.float32) + dz * dz).astype(np.float32)
                i = int(np.argmin(d2))
                if float(d2[i]) > gate:
                    continue
                best.append((float(d2[i]), ss, i))
            best.sort()
            if len(best) < 2:
                out.append((0, best[0][1] if best else -1, best[0][2] if best else 0, -1, 0, 0))
                continue
            (di, si, ii), (dj, sj, ij) = best[0], best[1]
            P = d["tgt_xyz"][d["tgt_off"][si]:d["tgt_off"][si + 1]]
            n = len(P)
            k1, k2 = (ii + 1) % n, (ii - 1 + n) % n

            def n2(a):
                v = (a - q).astype(np.float32)
                return np.float32(np.float32(v[0] * v[0] + v[1] * v[1]) + v[2] * v[2])
            ik = k1 if n2(P[k1]) < n2(P[k2]) else k2
            v0, v1, v2 = P[ii], d["tgt_xyz"][d["tgt_off"][sj] + ij], P[ik]
            N = np.cross((v1 - v0).astype(np.float32), (v2 - v0).astype(np.float32)).astype(np.float32)
            valid = int(np.sqrt(np.float32(np.float32(N[0] * N[0] + N[1] * N[1]) + N[2] * N[2])) >= norm_cond)
            out.append((valid, si, ii, sj, ij, ik))
    return out


@pytest.mark.parametrize("iter_", [1, 2])
def test_association_matches_independent_numpy(iter_):
    d = H.small_pair(8, 48)
    orc = ol.Oracle(icp_skip=3)
    orc.set_target(d["tgt_xyz"], d["tgt_off"])
    orc.set_source(d["src_xyz"], d["src_off"])
    x = d["x_true"]
    orc.associate(x, iter_)
    c = orc.correspondences()
    ref = _associate_numpy(d, x, iter_, 3)
    assert len(c) == len(ref)
    for row, (valid, si, ii, sj, ij, ik) in zip(c, ref):
        assert row["valid"] == valid and row["ring_i"] == si and row["ring_j"] == sj
        if si >= 0:
            assert row["idx_i"] == ii
        if sj >= 0:
            assert row["idx_j"] == ij and row["idx_k"] == ik
        if valid:
            assert abs(np.linalg.norm(row["n"]) - 1) < 1e-6 and np.array_equal(row["p"], d["src_xyz"][d["src_off"][row["src_ring"]] + row["src_idx"]])


def test_query_list_follows_icp_skip_and_enable_icp():
    d = H.small_pair(4, 50)
    orc = ol.Oracle(icp_skip=7)
    orc.set_target(d["tgt_xyz"], d["tgt_off"])
    orc.set_source(d["src_xyz"], d["src_off"])
    orc.associate(d["x0"], 1)
    c = orc.correspondences()
    assert len(c) == 4 * 8 and list(c["src_idx"][:8]) == [0, 7, 14, 21, 28, 35, 42, 49]      # smi += icp_skip, velo.h:807
    orc.set_params(enable_icp=0)                                                              # loop bound * enable_icp, velo.h:806
    assert orc.associate(d["x0"], 1) == 0 and len(orc.correspondences()) == 0


def test_gate_is_on_squared_distance_and_shrinks_with_iter4():
    tgt = np.array([[0, 0, 0], [1, 0, 0], [0, 0.6, 0], [1, 0.6, 0]], dtype=np.float32)
    off = np.array([0, 2, 4], dtype=np.int32)
    src = np.array([[0.1, 0.3, 0.0]], dtype=np.float32)
    orc = ol.Oracle(icp_skip=1)
    orc.set_target(tgt, off)
    orc.set_source(src, np.array([0, 1], dtype=np.int32))
    assert orc.associate(np.zeros(6), 1) == 1            # d^2 = 0.1 <= 0.5
    assert orc.associate(np.zeros(6), 2) == 0            # 0.5 / 16 = 0.03125 < 0.1  (velo.h:829)


# ---- evaluation / solver ------------------------------------------------------------------------------------------------
def test_evaluate_equals_sum_over_rows_and_manual_robustifier():
    d = H.small_pair(8, 64)
    vis = synth.stereo_matches(30, mix="all")
    orc = ol.Oracle(icp_skip=2)
    orc.set_target(d["tgt_xyz"], d["tgt_off"])
    orc.set_source(d["src_xyz"], d["src_off"])
    orc.set_visual(vis)
    x = d["x0"] + 0.01
    orc.build_visual(x, 1)
    orc.associate(x, 1)
    cost, Hm, g = orc.evaluate(x)
    r, J = orc.evaluate_rows(x)
    np.testing.assert_allclose(Hm, J.T @ J, rtol=1e-12)
    np.testing.assert_allclose(g, J.T @ r, rtol=1e-12, atol=1e-14)
    # ICP rows: sqrt(rho') * raw residual with rho' = 1 / (1 + r^2 / 0.01)   
    c = orc.correspondences()
    v = c[c["valid"] == 1]
    raw = np.array([ol.functor(4, np.concatenate([q["p"], q["n"], q["v0"]]).astype(np.float64), x)[0][0] for q in v])
    np.testing.assert_allclose(r[-len(v):], raw / np.sqrt(1 + raw ** 2 / 0.01), rtol=1e-12, atol=1e-15)
    assert cost > 0


def test_lm_reaches_the_scipy_robust_optimum():
    """Same optimum (not the same iterates): SciPy soft-cauchy least squares on identical fixed correspondences."""
    d = H.small_pair(16, 96)
    orc = ol.Oracle(icp_skip=1)
    orc.set_target(d["tgt_xyz"], d["tgt_off"])
    orc.set_source(d["src_xyz"], d["src_off"])
    orc.associate(d["x_true"], 1)
    c = orc.correspondences()
    v = c[c["valid"] == 1]
    p, n, v0 = (v[k].astype(np.float64) for k in ("p", "n", "v0"))

    def res(x):
        return np.einsum("ij,ij->i", Rotation.from_rotvec(x[:3]).apply(p) + x[3:] - v0, n)
    x0 = d["x_true"] + np.array([1e-3, -1e-3, 1e-3, 0.02, -0.02, 0.02])
    xs, s = orc.solve(x0)
    assert s.termination == 0 and s.final_cost <= s.initial_cost
    sp = least_squares(res, x0, loss="cauchy", f_scale=0.1, xtol=1e-14, ftol=1e-14, gtol=1e-14)
    cost_sp = 0.5 * np.sum(0.01 * np.log1p(res(sp.x) ** 2 / 0.01))
    cost_or = 0.5 * np.sum(0.01 * np.log1p(res(xs) ** 2 / 0.01))
    assert abs(cost_or - s.final_cost) <= 1e-12 * cost_or
    assert cost_or <= cost_sp * (1 + 1e-5)                       # ceres stops on function_tolerance 1e-6
    assert np.linalg.norm(xs - sp.x) < 2e-3


def test_solve_without_blocks_is_immediate_convergence():
    orc = ol.Oracle()
    d = H.small_pair(4, 16)
    orc.set_target(d["tgt_xyz"], d["tgt_off"])
    orc.set_source(d["src_xyz"], d["src_off"])
    x, s = orc.solve([0, 0, 0, 100.0, 0, 0])
    assert (s.termination, s.lm_iterations, s.evaluations) == (0, 0, 1) and x[3] == 100.0


def test_visual_gate_semantics():
    """iter 1 gates nothing; iter 2 drops outliers and `continue` skips the REMAINING types of that match."""
    vis = synth.stereo_matches(100, mix="all", outlier_frac=0.3)
    d = H.small_pair(4, 16)
    orc = ol.Oracle()
    orc.set_target(d["tgt_xyz"], d["tgt_off"])
    orc.set_source(d["src_xyz"], d["src_off"])
    orc.set_visual(vis)
    n1 = orc.build_visual(d["x_true"], 1)
    g1 = orc.good_matches()
    kinds = np.arange(200) % 4
    expect1 = int(np.sum(kinds == 0) + np.sum(kinds == 1) + 3 * np.sum(kinds == 2) + np.sum(kinds == 3))
    assert n1 == expect1 == len(g1)
    n2 = orc.build_visual(d["x_true"], 2)
    g2 = orc.good_matches()
    assert 0 < n2 < n1
    # a both-depth match emits 3D3D, 3D2D, 2D3D in that order, and never a later type without the earlier ones
    by_match = {}
    for row in g2:
        by_match.setdefault((int(row["cam"]), int(row["point1"])), []).append(int(row["residual_type"]))
    for (cam, p1), types in by_match.items():
        if kinds[p1] == 2:
            assert types in ([0], [0, 1], [0, 1, 2])
    orc.set_params(enable_2d2d=0, enable_3d2d=0)
    assert orc.build_visual(d["x_true"], 1) == int(np.sum(kinds == 2))      # only 3D3D is unconditional (velo.h:662)


def test_frame_to_frame_recovers_simulated_motion_and_counts_bytes():
    d = H.small_pair(32, 400)
    orc = ol.Oracle(threads=4, icp_skip=1)
    orc.set_target(d["tgt_xyz"], d["tgt_off"])
    orc.set_source(d["src_xyz"], d["src_off"])
    x, T, s = orc.frame_to_frame(d["x0"])
    assert s.n_solves == 6 and s.n_assoc_rounds == 6 and s.n_queries == 12800
    assert np.linalg.norm(x[3:] - d["x_true"][3:]) < 2e-3 and np.linalg.norm(x[:3] - d["x_true"][:3]) < 1e-3
    np.testing.assert_allclose(T, ol.pose_vec_to_mat(x))
    want = 6 * (12 * 12800 + 12 * 12800 + 28 * 12800)
    want += sum(s.solves[k].evaluations * (36 * s.solves[k].n_icp_valid + 224) for k in range(6))
    assert s.algorithmic_bytes == want and s.assoc_bytes == 6 * 52 * 12800


def test_query_shards_partition_the_work():
    d = H.small_pair(8, 64)
    x = d["x0"]
    whole = ol.Oracle(icp_skip=1)
    whole.set_target(d["tgt_xyz"], d["tgt_off"])
    whole.set_source(d["src_xyz"], d["src_off"])
    whole.associate(x, 1)
    c0, H0, g0 = whole.evaluate(x)
    acc = [0.0, np.zeros((6, 6)), np.zeros(6)]
    parts = []
    for r in range(3):
        o = ol.Oracle(icp_skip=1)
        o.set_query_shard(r, 3)
        o.set_target(d["tgt_xyz"], d["tgt_off"])
        o.set_source(d["src_xyz"], d["src_off"])
        o.associate(x, 1)
        parts.append(o.correspondences())
        c, Hm, g = o.evaluate(x)
        acc[0] += c
        acc[1] += Hm
        acc[2] += g
    assert np.array_equal(np.concatenate(parts), whole.correspondences())
    assert abs(acc[0] - c0) < 1e-12 * c0 and H.rel_err(acc[1], H0) < 1e-12 and H.rel_err(acc[2], g0) < 1e-12


def test_residual_stats_restatement_against_numpy(oracle):
    """residualStats (velo.h:921-1025) restated in the oracle against an independent numpy evaluation of the same rule: unrobustified
    rows (vo_evaluate_rows applies the loss, so the rows are rebuilt from the functor constants), block norms, median = sorted[n // 2],
    mean, count; cost = 1/2 sum r^2."""
    from velo_amd import api, synth
    d = H.small_pair(16, 128)
    o = oracle.Oracle(threads=2, icp_skip=3)
    o.set_target(d["tgt_xyz"], d["tgt_off"]); o.set_source(d["src_xyz"], d["src_off"])
    o.set_visual(api.matches_from_dict(synth.stereo_matches(60, mix="all")))
    x = d["x_true"] + np.array([1e-3, -2e-3, 5e-4, 0.01, -0.02, 0.015])
    o.build_visual(x, 1)
    o.associate(x, 1)
    st = o.residual_stats(x)
    # independent: loss-free evaluation through the trivial-loss parameters (weights 1, thresholds huge -> rho(s) = s up to 1e-15)
    good = o.good_matches()
    corr = o.correspondences()
    n_icp = int((corr["valid"] != 0).sum())
    counts = {k: int((good["residual_type"] == k).sum()) for k in range(4)}
    assert [st.type[k].count for k in range(4)] == [counts[k] for k in range(4)]
    assert st.type[4].count == n_icp and st.n_blocks == len(good) + n_icp
    assert st.n_residuals == 3 * counts[0] + 2 * counts[1] + 2 * counts[2] + counts[3] + n_icp
    # 3DPD by hand: r = N . (R p + t - v0) in numpy doubles
    from scipy.spatial.transform import Rotation
    R = Rotation.from_rotvec(x[:3]).as_matrix()
    v = corr[corr["valid"] != 0]
    r = np.einsum("ij,ij->i", v["n"].astype(np.float64), (R @ v["p"].astype(np.float64).T).T + x[3:] - v["v0"].astype(np.float64))
    a = np.abs(r)
    assert abs(st.type[4].mean - a.mean()) <= 1e-12 * a.mean()
    assert abs(st.type[4].median - np.sort(a)[len(a) // 2]) <= 1e-12
    assert st.cost >= 0.5 * float((r * r).sum()) * (1 - 1e-12)
    if counts[0] + counts[1] + counts[2] + counts[3] == 0:
        assert abs(st.cost - 0.5 * float((r * r).sum())) <= 1e-12 * st.cost
