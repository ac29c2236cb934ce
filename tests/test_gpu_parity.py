"""-m gpu: the HIP path (through the C-ABI, via ctypes) against the CPU oracle on identical seeded inputs.

Bars: correspondence tables index-exact and bit-exact, JtJ/Jtr/cost <= 1e-12 relative,
solved pose within 1e-4 m / 1e-5 rad (north_star)."""
import numpy as np
import pytest

import helpers as H
import oracle_lib as O
from velo_amd import api, synth

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ctx(hip_lib):
    c = api.Context(0)
    yield c
    c.close()


def test_default_params_match_header_python_and_oracle(hip_lib, oracle):
    import ctypes as C
    p = api.VeloParams()
    assert hip_lib.velo_default_params(C.byref(p)) == 0
    q = api.default_params()
    o = oracle.default_params()
    for name, _ in api.VeloParams._fields_:
        assert getattr(p, name) == getattr(q, name) == getattr(o, name), name


@pytest.mark.parametrize("iter_", [1, 2])
@pytest.mark.parametrize("skip", [1, 7])
def test_association_index_exact(ctx, oracle, iter_, skip):
    d = H.small_pair(16, 128)
    orc = oracle.Oracle(threads=4)
    H.load_both(ctx, orc, d, icp_skip=skip)
    for x in (d["x0"], d["x_true"], np.zeros(6), [0.3, -0.2, 0.1, 0.5, -0.25, 2.0]):
        n_gpu = ctx.associate(x, iter_)
        n_cpu = orc.associate(x, iter_)
        a, b = ctx.correspondences(), orc.correspondences()
        H.assert_corr_equal(a, b)
        assert n_gpu == n_cpu == int(a["valid"].sum())


def test_association_medium_cloud(ctx, oracle):
    d = H.small_pair(32, 512)
    orc = oracle.Oracle(threads=8)
    H.load_both(ctx, orc, d, icp_skip=1)
    for it in (1, 2, 3):
        assert ctx.associate(d["x0"], it) == orc.associate(d["x0"], it)
        H.assert_corr_equal(ctx.correspondences(), orc.correspondences())


def test_evaluate_normal_equations(ctx, oracle):
    d = H.small_pair(16, 128)
    orc = oracle.Oracle(threads=4)
    H.load_both(ctx, orc, d, icp_skip=1)
    ctx.associate(d["x0"], 1)
    orc.associate(d["x0"], 1)
    for x in (d["x0"], d["x_true"], np.zeros(6), [1e-9, 0, 0, 0, 0, 1.0], [0.3, -0.2, 0.1, 0.5, -0.25, 2.0]):
        c1, H1, g1 = ctx.evaluate(x)
        c2, H2, g2 = orc.evaluate(x)
        assert abs(c1 - c2) <= 1e-12 * max(abs(c2), 1e-300)
        assert H.rel_err(H1, H2) <= 1e-12
        assert H.rel_err(g1, g2) <= 1e-12


def test_evaluate_rows_match(ctx, oracle):
    d = H.small_pair(16, 64)
    vis = synth.stereo_matches(n_per_cam=40, mix="all")
    orc = oracle.Oracle()
    H.load_both(ctx, orc, d, visual=vis, icp_skip=3)
    x = d["x0"] + 0.01
    assert ctx.build_visual(x, 1) == orc.build_visual(x, 1)
    assert ctx.associate(x, 1) == orc.associate(x, 1)
    r1, J1 = ctx.evaluate_rows(x)
    r2, J2 = orc.evaluate_rows(x)
    assert r1.shape == r2.shape and J1.shape == J2.shape and len(r1) > 0
    np.testing.assert_allclose(r1, r2, rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(J1, J2, rtol=1e-12, atol=1e-13)


@pytest.mark.parametrize("mix", ["reproj", "all"])
@pytest.mark.parametrize("iter_", [1, 2])
def test_visual_gate_and_evaluate(ctx, oracle, mix, iter_):
    d = H.small_pair(8, 64)
    vis = synth.stereo_matches(n_per_cam=150, mix=mix)
    orc = oracle.Oracle()
    H.load_both(ctx, orc, d, visual=vis, icp_skip=4)
    for x in (d["x_true"], d["x0"]):
        assert ctx.build_visual(x, iter_) == orc.build_visual(x, iter_)
        g1, g2 = ctx.good_matches(), orc.good_matches()
        assert np.array_equal(g1, g2)
        c1, H1, gg1 = ctx.evaluate(x)     # no association yet on a fresh table -> visual blocks only
        c2, H2, gg2 = orc.evaluate(x)
        assert abs(c1 - c2) <= 1e-12 * abs(c2)
        assert H.rel_err(H1, H2) <= 1e-12 and H.rel_err(gg1, gg2) <= 1e-12


def test_single_solve_matches(ctx, oracle):
    d = H.small_pair(16, 128)
    orc = oracle.Oracle(threads=4)
    H.load_both(ctx, orc, d, icp_skip=1)
    ctx.associate(d["x0"], 1)
    orc.associate(d["x0"], 1)
    x1, s1 = ctx.solve(d["x0"])
    x2, s2 = orc.solve(d["x0"])
    assert s1.termination == s2.termination
    assert s1.lm_iterations == s2.lm_iterations and s1.evaluations == s2.evaluations
    assert s1.n_icp_valid == s2.n_icp_valid
    assert abs(s1.final_cost - s2.final_cost) <= 1e-10 * s2.final_cost
    assert H.pose_close(x1, x2, 1e-9, 1e-10)


@pytest.mark.parametrize("shape", [(16, 128), (32, 400)])
def test_frame_to_frame_pose_parity(ctx, oracle, shape):
    d = H.small_pair(*shape)
    orc = oracle.Oracle(threads=8)
    H.load_both(ctx, orc, d, icp_skip=1)
    x1, T1, s1 = ctx.frame_to_frame(d["x0"])
    x2, T2, s2 = orc.frame_to_frame(d["x0"])
    assert H.pose_close(x1, x2), (x1, x2)                       # 1e-4 m / 1e-5 rad
    assert s1.n_solves == s2.n_solves == 6
    for k in range(6):
        a, b = s1.solves[k], s2.solves[k]
        assert (a.termination, a.lm_iterations, a.evaluations, a.n_icp_valid) == \
               (b.termination, b.lm_iterations, b.evaluations, b.n_icp_valid)
    assert s1.algorithmic_bytes == s2.algorithmic_bytes
    np.testing.assert_allclose(T1, T2, atol=1e-6)
    # and both land near the simulated motion (coarse clouds: centimetres; the full-size check is in test_gpu_fullsize.py)
    assert np.linalg.norm(x1[3:] - d["x_true"][3:]) < 0.08 and np.linalg.norm(x1[:3] - d["x_true"][:3]) < 0.03


def test_frame_to_frame_with_visual(ctx, oracle):
    d = H.small_pair(16, 128)
    vis = synth.stereo_matches(n_per_cam=200, mix="all")
    orc = oracle.Oracle(threads=4)
    H.load_both(ctx, orc, d, visual=vis, icp_skip=2)
    x1, _, s1 = ctx.frame_to_frame(d["x0"])
    x2, _, s2 = orc.frame_to_frame(d["x0"])
    assert H.pose_close(x1, x2), (x1, x2)
    assert np.array_equal(ctx.good_matches(), orc.good_matches())
    for k in range(6):
        assert s1.solves[k].n_visual_blocks == s2.solves[k].n_visual_blocks
        assert s1.solves[k].n_visual_residuals == s2.solves[k].n_visual_residuals


def test_reference_constants_config1(ctx, oracle):
    """BASELINE config 1 shape: reference constants (icp_skip=200) on a full 64x1875 KITTI-layout pair."""
    d = synth.scan_pair()
    orc = oracle.Oracle(threads=8)
    H.load_both(ctx, orc, d)           # defaults = reference constants
    x1, _, s1 = ctx.frame_to_frame(d["x0"])
    x2, _, s2 = orc.frame_to_frame(d["x0"])
    assert s1.n_queries == s2.n_queries == 64 * 10
    assert H.pose_close(x1, x2), (x1, x2)


def test_query_shards_sum_to_whole(ctx, oracle):
    """Multi-GPU contract on one GPU: the per-shard normal equations add up to the unsharded ones."""
    d = H.small_pair(16, 128)
    ctx.set_params(icp_skip=1)
    ctx.set_target(d["tgt_xyz"], d["tgt_off"])
    ctx.set_source(d["src_xyz"], d["src_off"])
    x = d["x0"]
    ctx.associate(x, 1)
    c0, H0, g0 = ctx.evaluate(x)
    whole = ctx.correspondences()
    acc_c, acc_H, acc_g, parts = 0.0, np.zeros((6, 6)), np.zeros(6), []
    for r in range(3):
        ctx.set_query_shard(r, 3)
        ctx.associate(x, 1)
        parts.append(ctx.correspondences())
        c, Hm, g = ctx.evaluate(x)
        acc_c += c
        acc_H += Hm
        acc_g += g
    ctx.set_query_shard(0, 1)
    assert np.array_equal(np.concatenate(parts), whole)
    assert abs(acc_c - c0) <= 1e-12 * c0 and H.rel_err(acc_H, H0) <= 1e-12 and H.rel_err(acc_g, g0) <= 1e-12


def test_edge_cases(ctx, oracle):
    d = H.small_pair(4, 32)
    orc = oracle.Oracle()
    # far-away guess: no ring within the gate -> zero blocks, solve converges immediately, x unchanged
    H.load_both(ctx, orc, d, icp_skip=1)
    far = np.array([0, 0, 0, 500.0, 0, 0])
    assert ctx.associate(far, 1) == orc.associate(far, 1) == 0
    x1, s1 = ctx.solve(far)
    x2, s2 = orc.solve(far)
    assert np.array_equal(x1, far) and np.array_equal(x2, far)
    assert s1.termination == s2.termination == 0 and s1.lm_iterations == s2.lm_iterations == 0
    # single-ring target: never two distinct rings -> nothing valid
    one = dict(d)
    one["tgt_xyz"], one["tgt_off"] = d["tgt_xyz"][:32], np.array([0, 32], dtype=np.int32)
    H.load_both(ctx, orc, one, icp_skip=1)
    assert ctx.associate(d["x_true"], 1) == orc.associate(d["x_true"], 1) == 0
    H.assert_corr_equal(ctx.correspondences(), orc.correspondences())
    # ragged rings (different lengths, one of length 1) and 16-byte stride input
    offs = np.array([0, 1, 20, 52, 128], dtype=np.int32)
    rag = dict(d)
    rag["tgt_off"] = offs
    xyz4 = np.zeros((128, 4), dtype=np.float32)
    xyz4[:, :3] = d["tgt_xyz"]
    ctx.set_target(xyz4, offs)
    orc.set_target(d["tgt_xyz"], offs)
    ctx.set_source(d["src_xyz"], d["src_off"])
    orc.set_source(d["src_xyz"], d["src_off"])
    assert ctx.associate(d["x_true"], 1) == orc.associate(d["x_true"], 1)
    H.assert_corr_equal(ctx.correspondences(), orc.correspondences())


def test_tie_breaks_and_wraparound(ctx, oracle):
    """Equal distances across rings -> lower ring wins (velo.h:836,843); np_i = 0 / n-1 wrap (velo.h:852-854);
    equal neighbour distances -> the -1 neighbour (velo.h:859); degenerate triangle -> skipped (velo.h:873).
    Coordinates are multiples of 1/8 so that the ties are exact in float."""
    u = np.float32(0.125)

    def ring(y):
        return np.stack([np.arange(5, dtype=np.float32) * u, np.full(5, y, np.float32), np.zeros(5, np.float32)], 1)

    dup = np.tile(np.float32([[2 * u, 3 * u, 0]]), (5, 1))          # ring 3: five copies of one point
    tgt = np.concatenate([ring(u), ring(-u), ring(-u), dup])        # rings 1 and 2 tie with ring 0 for y = 0 queries
    off = np.array([0, 5, 10, 15, 20], dtype=np.int32)
    src = np.array([[0, 0, 0], [4 * u, 0, 0], [2 * u, 0, 0], [2 * u, 3 * u, 0]], dtype=np.float32)
    soff = np.array([0, 4], dtype=np.int32)
    orc = oracle.Oracle()
    for o in (ctx, orc):
        o.set_params(icp_skip=1)
        o.set_target(tgt, off)
        o.set_source(src, soff)
    x = np.zeros(6)
    assert ctx.associate(x, 1) == orc.associate(x, 1) == 3
    a, b = ctx.correspondences(), orc.correspondences()
    H.assert_corr_equal(a, b)
    assert list(a["ring_i"][:3]) == [0, 0, 0] and list(a["ring_j"][:3]) == [1, 1, 1]     # ties -> lowest rings
    assert (a["idx_i"][0], a["idx_k"][0]) == (0, 1)        # neighbours 1 and 4 (wrap): 1 is closer
    assert (a["idx_i"][1], a["idx_k"][1]) == (4, 3)        # neighbours 0 (wrap) and 3: 3 is closer
    assert (a["idx_i"][2], a["idx_k"][2]) == (2, 1)        # equidistant neighbours -> the -1 one
    assert a["valid"][3] == 0 and a["ring_i"][3] == 3 and a["idx_i"][3] == 0 and a["ring_j"][3] == 0   # |N| = 0
    # the same ties when the round is warm-started from the previous round's winners (seeds enter before the grid candidates):
    # exact multiples of 1/8 keep the distances tied after these translations too
    for xs in (np.array([0, 0, 0, u, 0, 0], dtype=np.float64), np.zeros(6), np.array([0, 0, 0, 0, 0, u], dtype=np.float64), np.zeros(6)):
        for it in (1, 2):
            assert ctx.associate(xs, it) == orc.associate(xs, it)
            H.assert_corr_equal(ctx.correspondences(), orc.correspondences())


def test_rccl_path_single_rank(ctx, oracle):
    """The query-sharded multi-GPU path (RCCL all-reduce of the 28-double block between sweep and LM step) with a
    1-rank communicator: same launches/collective calls as N ranks, result must equal the plain path and the oracle."""
    d = H.small_pair(16, 128)
    orc = oracle.Oracle(threads=4)
    H.load_both(ctx, orc, d, icp_skip=1)
    x_plain, _, s_plain = ctx.frame_to_frame(d["x0"])
    ctx.comm_init(api.comm_unique_id(), 0, 1)
    try:
        x_comm, _, s_comm = ctx.frame_to_frame(d["x0"])
        c1, H1, g1 = ctx.evaluate(d["x_true"])
    finally:
        ctx.comm_destroy()
    x_orc, _, s_orc = orc.frame_to_frame(d["x0"])
    c2, H2, g2 = orc.evaluate(d["x_true"])          # both sides hold the table of their last association round
    assert np.array_equal(x_plain, x_comm)
    assert H.pose_close(x_comm, x_orc)
    assert [s_comm.solves[k].evaluations for k in range(6)] == [s_orc.solves[k].evaluations for k in range(6)]
    assert abs(c1 - c2) <= 1e-12 * c2 and H.rel_err(H1, H2) <= 1e-12 and H.rel_err(g1, g2) <= 1e-12


def test_non_finite_points_and_degenerate_inputs(ctx, oracle):
    """Non-finite target points are never matched (PCL keeps only finite points in the tree), non-finite queries match
    nothing, enable_icp = 0 empties the query list (velo.h:806), and argument errors come back as statuses."""
    d = H.small_pair(8, 64)
    tgt = d["tgt_xyz"].copy()
    src = d["src_xyz"].copy()
    tgt[5] = [np.nan, 0, 0]
    tgt[100] = [np.inf, 1, 2]
    tgt[101, 2] = -np.inf
    src[7] = [np.nan, np.nan, np.nan]
    src[70, 1] = np.inf
    orc = oracle.Oracle()
    for o in (ctx, orc):
        o.set_params(icp_skip=1)
        o.set_target(tgt, d["tgt_off"])
        o.set_source(src, d["src_off"])
    for it in (1, 2):
        assert ctx.associate(d["x_true"], it) == orc.associate(d["x_true"], it)
        a, b = ctx.correspondences(), orc.correspondences()
        assert a["valid"][7] == 0 and a["valid"][70] == 0
        keep = np.ones(len(a), bool)
        keep[[7, 70]] = False                       # p of a NaN query is NaN on both sides; compare the rest bit for bit
        H.assert_corr_equal(a[keep], b[keep])
        for f in ("idx_i", "idx_j", "idx_k"):
            pass
        used = set()
        for r, i in zip(a["ring_i"][a["ring_i"] >= 0], a["idx_i"][a["ring_i"] >= 0]):
            used.add(int(d["tgt_off"][r] + i))
        assert not ({5, 100, 101} & used)
    x1, _, s1 = ctx.frame_to_frame(d["x0"])
    x2, _, s2 = orc.frame_to_frame(d["x0"])
    assert H.pose_close(x1, x2)
    # enable_icp = 0: no queries, six solves that converge immediately, x unchanged
    ctx.set_params(enable_icp=0)
    x3, _, s3 = ctx.frame_to_frame(d["x0"])
    assert np.array_equal(x3, d["x0"]) and s3.n_queries == 0 and all(s3.solves[k].evaluations == 1 for k in range(6))
    ctx.set_params(enable_icp=1)
    # statuses, not exceptions across the ABI
    with pytest.raises(api.VeloError):
        ctx.set_target(d["tgt_xyz"], np.array([0, 10, 10, 20], dtype=np.int32))      # empty ring
    with pytest.raises(api.VeloError):
        ctx.set_params(icp_skip=0)
    with pytest.raises(api.VeloError):
        ctx.associate(d["x0"], 0)


def test_small_problem_single_launch_solve_is_bit_identical(hip_lib, diag_lib, monkeypatch):
    """Problems whose sweep fits a few workgroups (the reference's icp_skip = 200) run a whole solve in one single-workgroup launch
    (lm_solve_small_kernel); it walks the same virtual blocks with the same arithmetic, so poses, costs and iteration counts
    equal the launch-per-iteration path bit for bit."""
    d = synth.scan_pair(n_beams=64, n_azimuth=1875)
    vis = api.matches_from_dict(synth.stereo_matches(40, mix="all"))
    out = []
    for small in ("1", "0"):
        monkeypatch.setenv("VELO_SMALL_SOLVE", small)         # honoured by the diagnostics build only: "1" runs on the product library
        c = api.Context(0, lib=diag_lib if small == "0" else None)   # reference constants: icp_skip = 200 -> 640 queries
        c.set_target(d["tgt_xyz"], d["tgt_off"]); c.set_source(d["src_xyz"], d["src_off"]); c.set_visual(vis)
        out.append(c.frame_to_frame(d["x0"]))
        c.close()
    (x1, T1, s1), (x0, T0, s0) = out
    assert s1.n_queries == 640
    assert np.array_equal(x1, x0) and np.array_equal(T1, T0)
    for k in range(s1.n_solves):
        a, b = s1.solves[k], s0.solves[k]
        assert (a.termination, a.lm_iterations, a.evaluations, a.n_icp_valid, a.n_visual_blocks) == (b.termination, b.lm_iterations, b.evaluations, b.n_icp_valid, b.n_visual_blocks)
        assert a.initial_cost == b.initial_cost and a.final_cost == b.final_cost


@pytest.mark.parametrize("env", [{}, {"VELO_ASKER_ROWS": "0"}, {"VELO_CLUSTER_W": "2"}, {"VELO_DENSE_REF": "300"},
                                 {"VELO_DENSE_REF": "300", "VELO_ASKER_ROWS": "0"}, {"VELO_WARM_START": "0"}, {"VELO_ASSOC_LANE": "1"},
                                 {"VELO_DENSE_REF": "300", "VELO_ASKER_QUEUE": "0"}, {"VELO_DENSE_REF": "300", "VELO_ASKER_ROWS": "0", "VELO_ASKER_QUEUE": "0"},
                                 {"VELO_DENSE_REF": "300", "VELO_DENSE_ROWS": "4"}, {"VELO_DENSE_REF": "300", "VELO_DENSE_ROWS": "40", "VELO_ASKER_QUEUE": "0"},
                                 {"VELO_DENSE_REF": "300", "VELO_DENSE_ROWS": "1", "VELO_WARM_START": "0"}, {"VELO_XCD_CHUNKS": "1"},
                                 {"VELO_XCD_CHUNKS": "1", "VELO_DENSE_REF": "300"}, {"VELO_GRID_COMPRESS": "1"},
                                 {"VELO_GRID_COMPRESS": "1", "VELO_DENSE_REF": "300"}, {"VELO_GRID_COMPRESS": "1", "VELO_DENSE_REF": "300", "VELO_ASKER_QUEUE": "0"}])
def test_association_random_geometry_all_paths(hip_lib, diag_lib, oracle, monkeypatch, env):
    """Every way through the tube kernel (tile pass / query-by-query second phase, one or many clusters, regular or
    density-shrunk grid, with or without seeds) against the oracle on clouds that look nothing like a street scan: ragged
    rings, dense clumps next to voids, points far outside the bulk, repeated rounds at moving poses."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    rng = np.random.default_rng(12)
    n_rings = 23
    lens = rng.integers(1, 140, n_rings)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    centres = rng.uniform(-4, 4, (6, 3))
    which = rng.integers(0, 6, off[-1])
    tgt = (centres[which] + rng.normal(0, 0.35, (off[-1], 3)) * rng.choice([0.1, 1.0, 3.0], (off[-1], 1))).astype(np.float32)
    tgt[rng.integers(0, off[-1], 5)] += np.float32(40.0)                      # far outliers stretch the bounding box
    slens = rng.integers(1, 90, 17)
    soff = np.concatenate([[0], np.cumsum(slens)]).astype(np.int32)
    src = (centres[rng.integers(0, 6, soff[-1])] + rng.normal(0, 0.5, (soff[-1], 3))).astype(np.float32)
    c = api.Context(0, lib=diag_lib if env else None, icp_skip=1)   # {}: the product library; the switches: the diagnostics build
    o = oracle.Oracle(icp_skip=1)
    for obj in (c, o):
        obj.set_target(tgt, off)
        obj.set_source(src, soff)
    poses = [np.zeros(6), np.array([0.01, -0.02, 0.015, 0.05, -0.03, 0.02]), np.array([0.01, -0.02, 0.015, 0.051, -0.03, 0.02]),
             np.array([0.3, 0.1, -0.2, 1.0, -0.5, 0.3]), np.zeros(6)]
    for it in (1, 2, 1):
        for x in poses:
            assert c.associate(x, it) == o.associate(x, it)
            H.assert_corr_equal(c.correspondences(), o.correspondences())
    c.close()


@pytest.mark.gpu
def test_functor_batch_matches_oracle_autodiff(hip_lib):
    """velo_evaluate_functors (seam 2 by value): 5 functor kinds x 400 random constant sets at large, tiny and zero
    rotations against the oracle's dual-number evaluation; rows beyond a functor's dimension are zero; bad kinds are refused."""
    rng = np.random.default_rng(77)
    ncon = {0: 6, 1: 8, 2: 8, 3: 7, 4: 9}
    dim = {0: 3, 1: 2, 2: 2, 3: 1, 4: 1}
    n = 2000
    kinds = rng.integers(0, 5, size=n).astype(np.int32)
    consts = np.zeros((n, 9))
    for i, k in enumerate(kinds):
        consts[i, :ncon[k]] = rng.normal(size=ncon[k]) * (0.3 if k == 3 else 3.0)
    c = api.Context(0)
    for w_scale in (0.0, 1e-9, 1e-4, 0.3, 2.5):
        x = np.concatenate([rng.normal(size=3) * w_scale, rng.normal(size=3)])
        r, J = c.evaluate_functors(kinds, consts, x)
        r_only, none = c.evaluate_functors(kinds, consts, x, want_jacobian=False)
        assert none is None and np.array_equal(r, r_only)
        for i in range(0, n, 7):
            ro, Jo = O.functor(int(kinds[i]), consts[i], x)
            d = dim[int(kinds[i])]
            np.testing.assert_allclose(r[i, :d], ro, rtol=1e-12, atol=1e-13)
            np.testing.assert_allclose(J[i, :d], Jo, rtol=1e-11, atol=1e-12)
            assert not r[i, d:].any() and not J[i, d:].any()
    r0, J0 = c.evaluate_functors(np.zeros(0, np.int32), np.zeros((0, 9)), np.zeros(6))
    assert r0.shape == (0, 3)
    with pytest.raises(api.VeloError):
        c.evaluate_functors([5], np.zeros((1, 9)), np.zeros(6))
    c.close()


def test_randomized_parity_sweep(hip_lib, oracle):
    """A dozen seeds of tools/fuzz_parity.py: random clumpy geometries (tables at moving poses through warm rounds) and small street
    pairs registered singly and in lock-step batches, everything against the oracle (the full sweep: 300 seeds, tools/README.md)."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_parity.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    assert m.run(12, first_seed=500) > 150


def test_shared_target_equals_own_copy_and_outlives_its_owner(hip_lib, oracle):
    """velo_share_target: contexts that register different scans against ONE target hold it by reference.  Same tables and poses as
    with a copy of their own; the shared target stays valid when the context that built it loads something else or is destroyed."""
    d = H.small_pair(24, 200)
    d2 = H.small_pair(24, 200, scene_seed=5)
    owner, own_copy, borrower = api.Context(0, icp_skip=1), api.Context(0, icp_skip=1), api.Context(0, icp_skip=1)
    owner.set_target(d["tgt_xyz"], d["tgt_off"])
    own_copy.set_target(d["tgt_xyz"], d["tgt_off"])
    borrower.share_target(owner)
    for c in (own_copy, borrower):
        c.set_source(d2["src_xyz"], d2["src_off"])                      # a different scan than the owner's
    assert borrower.associate(d["x0"], 1) == own_copy.associate(d["x0"], 1)
    H.assert_corr_equal(borrower.correspondences(), own_copy.correspondences())
    owner.set_target(d2["tgt_xyz"], d2["tgt_off"])                      # the owner moves on: the borrower keeps the old target
    xa, Ta, sa = own_copy.frame_to_frame(d["x0"])
    xb, Tb, sb = borrower.frame_to_frame(d["x0"])
    assert np.array_equal(xa, xb) and sa.n_solves == sb.n_solves
    owner.close()                                                       # ... and goes away
    xb2, _, _ = borrower.frame_to_frame(d["x0"])
    assert np.array_equal(xb2, xa)
    # batch entry: jobs flagged SCAN_SHARED with the same target descriptor are indexed once
    ctxs = [api.Context(0, icp_skip=1) for _ in range(4)]
    srcs = [(d["src_xyz"], d["src_off"]), (d2["src_xyz"], d2["src_off"])] * 2
    x0s = np.tile(d["x0"], (4, 1))
    tgt = (d["tgt_xyz"], d["tgt_off"])
    refs_sh = (api.scan_refs([tgt] * 4, 0, shared=True), api.scan_refs(srcs, 0))
    refs_own = (api.scan_refs([tgt] * 4, 0), api.scan_refs(srcs, 0))
    xs1, _, S1 = api.register_batch(ctxs, None, None, x0s, refs=refs_sh)
    xs2, _, S2 = api.register_batch(ctxs, None, None, x0s, refs=refs_own)
    assert np.array_equal(xs1, xs2) and np.array_equal(xs1[0], xs1[2]) and not np.array_equal(xs1[0], xs1[1])
    orc = oracle.Oracle(threads=4, icp_skip=1)
    orc.set_target(d["tgt_xyz"], d["tgt_off"]); orc.set_source(d2["src_xyz"], d2["src_off"])
    xo, _, _ = orc.frame_to_frame(d["x0"])
    assert H.pose_close(xs1[1], xo, 1e-9, 1e-10)
    # repeated shared calls rebuild the index IN PLACE: the sharers of the last call let go of it before their owner reloads (a target
    # others still hold is left to them and a new one allocated) -- no device memory is taken from call to call; a context OUTSIDE the batch
    # that borrowed the map keeps the old one when the batch moves on to another map, and still registers against it
    import torch
    outsider = api.Context(0, icp_skip=1)
    outsider.share_target(ctxs[0]); outsider.set_source(d2["src_xyz"], d2["src_off"])
    x_out, _, _ = outsider.frame_to_frame(d["x0"])
    for _ in range(2):
        api.register_batch(ctxs, None, None, x0s, refs=refs_sh)
    for _ in range(6):
        xs3, _, _ = api.register_batch(ctxs, None, None, x0s, refs=refs_sh)
    assert np.array_equal(xs3, xs1)
    # ... measured on FULL-SIZE scans, where an index is ~9 MB: were a new one allocated per call (what happened before the sharers let go
    # first), six calls would take ~50 MB; the runtime's own bookkeeping moves the device-wide figure by a megabyte or two at most
    big = synth.scan_pair()
    big_ctxs = [api.Context(0, icp_skip=1) for _ in range(4)]
    refs_big = (api.scan_refs([(big["tgt_xyz"], big["tgt_off"])] * 4, 0, shared=True), api.scan_refs([(big["src_xyz"], big["src_off"])] * 4, 0))
    x0b = np.tile(big["x0"], (4, 1))
    for _ in range(3):
        xb0, _, _ = api.register_batch(big_ctxs, None, None, x0b, refs=refs_big)
    torch.cuda.synchronize(); free0 = torch.cuda.mem_get_info()[0]
    for _ in range(6):
        xb, _, _ = api.register_batch(big_ctxs, None, None, x0b, refs=refs_big)
    torch.cuda.synchronize(); free1 = torch.cuda.mem_get_info()[0]
    assert np.array_equal(xb, xb0)
    assert free1 >= free0 - (16 << 20), (free0, free1)
    for c in big_ctxs:
        c.close()
    tgt2 = (d2["tgt_xyz"], d2["tgt_off"])
    refs_sh2 = (api.scan_refs([tgt2] * 4, 0, shared=True), api.scan_refs(srcs, 0))
    xs4, _, _ = api.register_batch(ctxs, None, None, x0s, refs=refs_sh2)            # another map for the batch
    x_out2, _, _ = outsider.frame_to_frame(d["x0"])                                 # the outsider still holds the first one
    assert np.array_equal(x_out2, x_out) and not np.array_equal(xs4, xs1)
    for c in ctxs + [own_copy, borrower, outsider]:
        c.close()


def _summary_tuple(s):
    return [(s.solves[k].termination, s.solves[k].lm_iterations, s.solves[k].evaluations, s.solves[k].n_icp_valid,
             s.solves[k].initial_cost, s.solves[k].final_cost) for k in range(s.n_solves)]


def test_chain_mode_is_bit_identical_and_survives_mispredictions(hip_lib, oracle, monkeypatch):
    """A LiDAR-only call is ONE chain of launches (pose scalars of the next round and the solve summaries stay on the device,
    velo_chain_stats).  Its results must equal the host-driven path's bit for bit -- pose, every solve's costs and counts -- also
    when the launch prediction is too short (margin 0, first call: the call is repeated host-driven) and for a pair the context
    has not seen before (different evaluation counts than predicted)."""
    a = synth.scan_pair(n_beams=32, n_azimuth=400, scene_seed=3)
    b = synth.scan_pair(n_beams=32, n_azimuth=400, scene_seed=4, sigma=0.05)
    b["x0"] = np.array([0.01, -0.02, 0.0, 0.3, 0.0, 0.4])        # a poor guess: other evaluation counts than the previous call's
    res = {}
    for name, env in (("host", {"VELO_CHAIN": "0"}), ("chain", {"VELO_CHAIN": "1"}), ("tight", {"VELO_CHAIN": "1", "VELO_CHAIN_MARGIN": "0"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        c = api.Context(0, icp_skip=1)
        monkeypatch.delenv("VELO_CHAIN_MARGIN", raising=False)
        out = []
        for d in (a, a, b, a):
            c.set_target(d["tgt_xyz"], d["tgt_off"]); c.set_source(d["src_xyz"], d["src_off"])
            x, T, s = c.frame_to_frame(d["x0"])
            out.append((x.copy(), T.copy(), _summary_tuple(s), s.n_assoc_rounds, s.assoc_kernel_launches, s.eval_kernel_launches, s.algorithmic_bytes))
        res[name] = (out, c.chain_stats())
        c.close()
    assert res["host"][1] == (0, 0)
    assert res["chain"][1][0] == 4 and res["tight"][1][0] == 4
    assert res["tight"][1][1] >= 1                              # margin 0: at least one call outran its prediction and was repeated
    for name in ("chain", "tight"):
        for (x0, T0, s0, *r0), (x1, T1, s1, *r1) in zip(res["host"][0], res[name][0]):
            assert np.array_equal(x0, x1) and np.array_equal(T0, T1) and s0 == s1 and r0 == r1
    # and the pose is the oracle's
    orc = oracle.Oracle(threads=8, icp_skip=1)
    orc.set_target(a["tgt_xyz"], a["tgt_off"]); orc.set_source(a["src_xyz"], a["src_off"])
    xo = orc.frame_to_frame(a["x0"])[0]
    assert np.abs(res["chain"][0][0][0] - xo).max() <= 1e-9


def test_chain_mode_lockstep_batch_is_bit_identical(hip_lib, diag_lib, monkeypatch):
    """Lock-step batch: host-driven vs chain mode, and sweep + LM step as ONE launch per iteration (eval_step_batch_kernel: the last
    workgroup of a context steps) vs two launches (VELO_LM_FUSED=0) -- poses, solves, counts and bytes equal bit for bit."""
    pairs = [synth.scan_pair(n_beams=32, n_azimuth=400, scene_seed=10 + k, sigma=0.02 * (k + 1)) for k in range(3)]
    pairs[2]["x0"] = np.array([0.01, -0.02, 0.0, 0.3, 0.0, 0.4])
    res = {}
    for name, env in (("host", {"VELO_CHAIN": "0", "VELO_LM_FUSED": "0"}), ("chain", {"VELO_CHAIN": "1", "VELO_LM_FUSED": "1"}),
                      ("tight", {"VELO_CHAIN": "1", "VELO_CHAIN_MARGIN": "0", "VELO_LM_FUSED": "1"}),
                      ("host_fused", {"VELO_CHAIN": "0", "VELO_LM_FUSED": "1"}), ("chain_two", {"VELO_CHAIN": "1", "VELO_LM_FUSED": "0"}),
                      # the lean one-launch iteration (every workgroup advances the state itself), with the default and with no margin
                      ("chain_iter", {"VELO_CHAIN": "1", "VELO_LM_FUSED": "1", "VELO_LM_ITER": "1"}),
                      ("tight_iter", {"VELO_CHAIN": "1", "VELO_CHAIN_MARGIN": "0", "VELO_LM_FUSED": "1", "VELO_LM_ITER": "1"}),
                      # a whole solve as ONE launch, all-gather form (lm_solve_ag_batch_kernel): no launch-count prediction, so no margin can miss
                      ("chain_ag", {"VELO_CHAIN": "1", "VELO_CHAIN_MARGIN": "0", "VELO_LM_FUSED": "1", "VELO_LM_PERSIST": "2"}),
                      ("chain_ag8", {"VELO_CHAIN": "1", "VELO_LM_FUSED": "1", "VELO_LM_PERSIST": "2"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        # VELO_LM_FUSED / VELO_LM_ITER / VELO_LM_PERSIST exist in the diagnostics build only; the product library fuses (its three variants run on the product library)
        ctxs = [api.Context(0, lib=diag_lib if (env["VELO_LM_FUSED"] == "0" or "VELO_LM_ITER" in env or "VELO_LM_PERSIST" in env) else None, icp_skip=1) for _ in pairs]
        monkeypatch.delenv("VELO_CHAIN_MARGIN", raising=False); monkeypatch.delenv("VELO_LM_ITER", raising=False); monkeypatch.delenv("VELO_LM_PERSIST", raising=False)
        out = []
        for rep in range(3):
            order = pairs if rep != 1 else pairs[::-1]           # the second call gives every context another pair than predicted
            for c, d in zip(ctxs, order):
                c.set_target(d["tgt_xyz"], d["tgt_off"]); c.set_source(d["src_xyz"], d["src_off"])
            x, T, S = api.frame_to_frame_batch(ctxs, [d["x0"] for d in order])
            out.append((x.copy(), T.copy(), [_summary_tuple(s) for s in S], [(s.n_assoc_rounds, s.eval_kernel_launches, s.algorithmic_bytes) for s in S]))
        res[name] = (out, [c.chain_stats() for c in ctxs])
        for c in ctxs:
            c.close()
    assert all(st == (0, 0) for st in res["host"][1]) and all(st == (0, 0) for st in res["host_fused"][1])
    assert all(st[0] == 3 for st in res["chain"][1]) and all(st[0] == 3 for st in res["chain_two"][1]) and all(st[0] == 3 for st in res["chain_iter"][1])
    assert any(st[1] >= 1 for st in res["tight"][1]) and any(st[1] >= 1 for st in res["tight_iter"][1])
    assert all(st == (3, 0) for st in res["chain_ag"][1]) and all(st == (3, 0) for st in res["chain_ag8"][1]), (res["chain_ag"][1], res["chain_ag8"][1])
    for name in ("chain", "tight", "host_fused", "chain_two", "chain_iter", "tight_iter", "chain_ag", "chain_ag8"):
        for (x0, T0, s0, r0), (x1, T1, s1, r1) in zip(res["host"][0], res[name][0]):
            assert np.array_equal(x0, x1) and np.array_equal(T0, T1) and s0 == s1 and r0 == r1


def test_lean_visual_sweep_in_shared_chip_groups_equals_single_calls(hip_lib):
    """Four contexts = two lock-step groups sharing the chip, so the groups' LM launches are the LEAN ones: a group whose contexts hold at
    most one visual block slot per thread runs the blocks inside the lean sweep + step launch (visual_sweep_one: block evaluated ahead of
    the accumulators, the epipolar block in three passes of two pose parameters); a group with more matches than that (6,000 > 5,461)
    keeps the visual sweep as a launch of its own.  Either way every pose, solve summary, block count and good_matches list equals the
    single-pair call's (one-launch iteration, 6-wide duals) bit for bit."""
    pairs = [synth.scan_pair(n_beams=32, n_azimuth=400, scene_seed=50 + k, sigma=0.02) for k in range(4)]
    vis = [api.matches_from_dict(synth.stereo_matches(6000, seed=11, mix="all", x_true=pairs[0]["x_true"])),
           api.matches_from_dict(synth.stereo_matches(100, seed=12, x_true=pairs[1]["x_true"])), None,
           api.matches_from_dict(synth.stereo_matches(700, seed=13, mix="all", x_true=pairs[3]["x_true"]))]

    def loaded():
        cs = [api.Context(0, icp_skip=1) for _ in pairs]
        for c, d, v in zip(cs, pairs, vis):
            c.set_target(d["tgt_xyz"], d["tgt_off"]); c.set_source(d["src_xyz"], d["src_off"])
            if v is not None:
                c.set_visual(v)
        return cs

    singles = loaded()
    ref = []
    for c, d in zip(singles, pairs):
        x, T, S = c.frame_to_frame(d["x0"])
        ref.append((x.copy(), T.copy(), _summary_tuple(S), [(S.solves[k].n_visual_blocks, S.solves[k].n_visual_residuals) for k in range(S.n_solves)], c.good_matches().tobytes()))
    batch = loaded()
    for rep in range(2):                                   # the second call chains from real history
        xs, Ts, Ss = api.frame_to_frame_batch(batch, [d["x0"] for d in pairs])
        for i, (x, T, st, vb, gm) in enumerate(ref):
            assert np.array_equal(xs[i], x) and np.array_equal(Ts[i], T) and _summary_tuple(Ss[i]) == st
            assert [(Ss[i].solves[k].n_visual_blocks, Ss[i].solves[k].n_visual_residuals) for k in range(Ss[i].n_solves)] == vb
            assert batch[i].good_matches().tobytes() == gm
        for c, d in zip(batch, pairs):
            c.set_source(d["src_xyz"], d["src_off"])
    assert all(nb > 0 for nb, _ in ref[0][3]) and all(nb == 0 for nb, _ in ref[2][3])
    for c in singles + batch:
        c.close()


def test_chain_mode_lockstep_batch_with_visual_blocks_is_bit_identical(hip_lib, diag_lib, oracle, monkeypatch):
    """Lock-step batches whose contexts carry stereo blocks (all four kinds) go down the chain as well: the residual-type choice and the
    outlier gate of every f2f iteration run on the device at the pose the device holds, the visual sweep rides ahead of every fused
    sweep + step launch.  Poses, solves, block counts and good_matches equal the host-driven batch bit for bit and the oracle's pose."""
    pairs = [synth.scan_pair(n_beams=32, n_azimuth=400, scene_seed=30 + k, sigma=0.02 * (k + 1)) for k in range(3)]
    vis = [api.matches_from_dict(synth.stereo_matches(50, seed=5, mix="all", x_true=pairs[0]["x_true"])), None,
           api.matches_from_dict(synth.stereo_matches(30, seed=6, x_true=pairs[2]["x_true"]))]
    res = {}
    for name, env in (("host", {"VELO_CHAIN": "0"}), ("chain", {"VELO_CHAIN": "1"}), ("tight", {"VELO_CHAIN": "1", "VELO_CHAIN_MARGIN": "0"}),
                      # the visual sweep as a launch of its own ahead of every fused sweep + step launch (diagnostics build; the product rides them in one)
                      ("separate", {"VELO_CHAIN": "1", "VELO_LM_VIS_MERGED": "0"}), ("separate_tight", {"VELO_CHAIN": "1", "VELO_CHAIN_MARGIN": "0", "VELO_LM_VIS_MERGED": "0"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        ctxs = [api.Context(0, lib=diag_lib if "VELO_LM_VIS_MERGED" in env else None, icp_skip=1) for _ in pairs]
        monkeypatch.delenv("VELO_CHAIN_MARGIN", raising=False); monkeypatch.delenv("VELO_LM_VIS_MERGED", raising=False)
        for c, v in zip(ctxs, vis):
            if v is not None:
                c.set_visual(v)
        out = []
        for rep in range(2):
            for c, d in zip(ctxs, pairs):
                c.set_target(d["tgt_xyz"], d["tgt_off"]); c.set_source(d["src_xyz"], d["src_off"])
            x, T, S = api.frame_to_frame_batch(ctxs, [d["x0"] for d in pairs])
            out.append((x.copy(), T.copy(), [_summary_tuple(s) for s in S], [(s.n_assoc_rounds, s.eval_kernel_launches, s.algorithmic_bytes) for s in S],
                        [c.good_matches().tobytes() for c in ctxs] + [[(s.solves[k].n_visual_blocks, s.solves[k].n_visual_residuals) for k in range(s.n_solves)] for s in S]))
        res[name] = (out, [c.chain_stats() for c in ctxs])
        for c in ctxs:
            c.close()
    assert all(st == (0, 0) for st in res["host"][1])
    assert all(st[0] == 2 for st in res["chain"][1])               # (margin 0 may or may not miss here; either way the results must not move)
    for name in ("chain", "tight", "separate", "separate_tight"):
        for (x0, T0, s0, r0, g0), (x1, T1, s1, r1, g1) in zip(res["host"][0], res[name][0]):
            assert np.array_equal(x0, x1) and np.array_equal(T0, T1) and s0 == s1 and r0 == r1 and g0 == g1
    assert all(nb > 0 for nb, _ in res["chain"][0][0][4][3]) and all(nb == 0 for nb, _ in res["chain"][0][0][4][4])   # context 0 solved with visual blocks, context 1 without
    orc = oracle.Oracle(threads=8, icp_skip=1)
    orc.set_target(pairs[0]["tgt_xyz"], pairs[0]["tgt_off"]); orc.set_source(pairs[0]["src_xyz"], pairs[0]["src_off"]); orc.set_visual(vis[0])
    xo = orc.frame_to_frame(pairs[0]["x0"])[0]
    assert H.pose_close(res["chain"][0][0][0][0], xo, 1e-9, 1e-10)


def test_lockstep_batch_at_the_reference_constants_uses_single_launch_solves(hip_lib, diag_lib, monkeypatch):
    """icp_skip = 200 (kitti.h:8) in a lock-step batch: every solve of the group is one single-workgroup launch per context
    (lm_solve_small_batch_kernel, the body of the single-pair kernel); poses, solves and counts equal single calls bit for bit, with the
    launch-per-iteration path (VELO_SMALL_SOLVE=0) and with the host-driven batch."""
    pairs = [synth.scan_pair(scene_seed=20 + k, sigma=0.01 * (k + 1)) for k in range(3)]
    singles = []
    for d in pairs:
        c = api.Context(0)                                   # reference constants: 640 queries
        c.set_target(d["tgt_xyz"], d["tgt_off"]); c.set_source(d["src_xyz"], d["src_off"])
        x, T, s = c.frame_to_frame(d["x0"])
        singles.append((x.copy(), T.copy(), _summary_tuple(s)))
        c.close()
    for env in ({}, {"VELO_SMALL_SOLVE": "0"}, {"VELO_CHAIN": "0"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        ctxs = [api.Context(0, lib=diag_lib if "VELO_SMALL_SOLVE" in env else None) for _ in pairs]
        for k in env:
            monkeypatch.delenv(k, raising=False)
        for rep in range(2):
            for c, d in zip(ctxs, pairs):
                c.set_target(d["tgt_xyz"], d["tgt_off"]); c.set_source(d["src_xyz"], d["src_off"])
            x, T, S = api.frame_to_frame_batch(ctxs, [d["x0"] for d in pairs])
            for i, (xs, Ts, ss) in enumerate(singles):
                assert np.array_equal(x[i], xs) and np.array_equal(T[i], Ts) and _summary_tuple(S[i]) == ss, (env, rep, i)
        for c in ctxs:
            c.close()
    # twelve contexts: two groups of six, so a group's solves take two launches (four items by value each)
    ctxs = [api.Context(0) for _ in range(12)]
    for c, d in zip(ctxs, pairs * 4):
        c.set_target(d["tgt_xyz"], d["tgt_off"]); c.set_source(d["src_xyz"], d["src_off"])
    x, T, S = api.frame_to_frame_batch(ctxs, [d["x0"] for d in pairs * 4])
    for i in range(12):
        xs, Ts, ss = singles[i % 3]
        assert np.array_equal(x[i], xs) and np.array_equal(T[i], Ts) and _summary_tuple(S[i]) == ss, i
    for c in ctxs:
        c.close()


def _stats_close(a, b):
    assert (a.n_blocks, a.n_residuals) == (b.n_blocks, b.n_residuals)
    assert abs(a.cost - b.cost) <= 1e-12 * max(abs(b.cost), 1e-300)
    for k in range(5):
        assert a.type[k].count == b.type[k].count, k
        if b.type[k].count:
            assert abs(a.type[k].median - b.type[k].median) <= 1e-12 * max(abs(b.type[k].median), 1e-30), (k, a.type[k].median, b.type[k].median)
            assert abs(a.type[k].mean - b.type[k].mean) <= 1e-12 * abs(b.type[k].mean), k
        else:
            assert a.type[k].median == 0.0 and a.type[k].mean == 0.0


def test_residual_stats_match_the_oracle_restatement(ctx, oracle):
    """residualStats (velo.h:921-1025) on the device -- block norms, radix-select median, means, counts, loss-free cost -- against the
    oracle's restatement: standalone at two poses with all five residual types present, and inside frame_to_frame (one record per
    f2f iteration, velo.h:909) where switching the statistics on must not change the pose or the solves."""
    d = H.small_pair(32, 400)
    vis = api.matches_from_dict(synth.stereo_matches(300, mix="all"))
    orc = oracle.Oracle(threads=8)
    H.load_both(ctx, orc, d, icp_skip=2)
    ctx.set_visual(vis); orc.set_visual(vis)
    for it, x in ((1, d["x0"]), (2, d["x_true"] + np.array([1e-3, -1e-3, 2e-3, 0.02, -0.01, 0.03]))):
        assert ctx.build_visual(x, it) == orc.build_visual(x, it)
        assert ctx.associate(x, it) == orc.associate(x, it)
        a, b = ctx.residual_stats(x), orc.residual_stats(x)
        assert all(b.type[k].count > 0 for k in range(5))
        _stats_close(a, b)
    x_plain, T_plain, s_plain = ctx.frame_to_frame(d["x0"])
    ctx.set_residual_stats(True); orc.set_residual_stats(True)
    x1, T1, s1 = ctx.frame_to_frame(d["x0"])
    x2, T2, s2 = orc.frame_to_frame(d["x0"])
    ctx.set_residual_stats(False)
    assert np.array_equal(x1, x_plain) and _summary_tuple(s1) == _summary_tuple(s_plain) and s_plain.n_residual_stats == 0
    assert s1.n_residual_stats == s2.n_residual_stats == 2
    assert H.pose_close(x1, x2)
    for k in range(2):
        a, b = s1.residual_stats[k], s2.residual_stats[k]
        assert (a.n_blocks, a.n_residuals) == (b.n_blocks, b.n_residuals)
        for t in range(5):
            assert a.type[t].count == b.type[t].count
            if b.type[t].count:          # the two sides end at poses that agree to ~1e-9, so do the statistics
                assert abs(a.type[t].median - b.type[t].median) <= 1e-6 * max(b.type[t].median, 1e-6)
                assert abs(a.type[t].mean - b.type[t].mean) <= 1e-6 * b.type[t].mean
    # LiDAR-only, larger: the chain is switched off by the statistics, the batch driver falls back to one call per context
    ctx.set_visual(None)
    ctx.set_residual_stats(True)
    xa, _, sa = ctx.frame_to_frame(d["x0"])
    ctx.set_residual_stats(False)
    xb, _, sb = ctx.frame_to_frame(d["x0"])
    assert np.array_equal(xa, xb) and sa.n_residual_stats == 2 and sa.residual_stats[1].type[4].count == sa.solves[5].n_icp_valid


def test_chain_mode_with_single_launch_solves_at_the_reference_constants(hip_lib, monkeypatch):
    """icp_skip = 200 (kitti.h:8): a solve is one single-workgroup launch, so the chain is 6 x (association + solve) with nothing to
    predict; pose, solves and summaries equal the host-driven path bit for bit, and no call is ever repeated."""
    d = synth.scan_pair()
    out = {}
    for name, chain in (("host", "0"), ("chain", "1")):
        monkeypatch.setenv("VELO_CHAIN", chain)
        c = api.Context(0)                                   # reference constants: 640 queries
        c.set_target(d["tgt_xyz"], d["tgt_off"]); c.set_source(d["src_xyz"], d["src_off"])
        res = []
        for x0 in (d["x0"], d["x_true"], d["x0"]):
            x, T, s = c.frame_to_frame(x0)
            res.append((x.copy(), T.copy(), _summary_tuple(s), s.n_queries, s.algorithmic_bytes))
        out[name] = (res, c.chain_stats())
        c.close()
    assert out["host"][1] == (0, 0) and out["chain"][1] == (3, 0)
    for (x0, T0, s0, *r0), (x1, T1, s1, *r1) in zip(out["host"][0], out["chain"][0]):
        assert np.array_equal(x0, x1) and np.array_equal(T0, T1) and s0 == s1 and r0 == r1 and r0[0] == 640


def test_chain_mode_with_visual_blocks_is_bit_identical(hip_lib, oracle, monkeypatch):
    """With stereo matches the residual-type choice + outlier gate of every f2f iteration runs on the device at the pose the device
    holds, and the chain carries sweep + visual sweep + step launches; pose, solves (incl. the visual block / residual counts per
    solve), good matches and bytes equal the host-driven path bit for bit -- large and single-launch-solve problems, tight margin."""
    cases = [(H.small_pair(32, 400), 1, synth.stereo_matches(300, mix="all")),          # 12,800 queries: launch-per-iteration solves
             (synth.scan_pair(), 200, synth.stereo_matches(40, mix="all")),             # reference constants: single-launch solves
             (H.small_pair(32, 400), 1, synth.stereo_matches(150))]                     # reprojection blocks only
    for d, skip, vis in cases:
        vis = api.matches_from_dict(vis)
        res = {}
        for name, env in (("host", {"VELO_CHAIN": "0"}), ("chain", {"VELO_CHAIN": "1"}), ("tight", {"VELO_CHAIN": "1", "VELO_CHAIN_MARGIN": "0"})):
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            c = api.Context(0, icp_skip=skip)
            monkeypatch.delenv("VELO_CHAIN_MARGIN", raising=False)
            c.set_target(d["tgt_xyz"], d["tgt_off"]); c.set_source(d["src_xyz"], d["src_off"]); c.set_visual(vis)
            out = []
            for x0 in (d["x0"], d["x_true"]):
                x, T, s = c.frame_to_frame(x0)
                vis_counts = [(s.solves[k].n_visual_blocks, s.solves[k].n_visual_residuals) for k in range(s.n_solves)]
                out.append((x.copy(), T.copy(), _summary_tuple(s), vis_counts, s.algorithmic_bytes, c.good_matches().tobytes()))
            res[name] = (out, c.chain_stats())
            c.close()
        assert res["host"][1] == (0, 0) and res["chain"][1][0] == 2
        for name in ("chain", "tight"):
            for (x0, T0, s0, v0, b0, g0), (x1, T1, s1, v1, b1, g1) in zip(res["host"][0], res[name][0]):
                assert np.array_equal(x0, x1) and np.array_equal(T0, T1) and s0 == s1 and v0 == v1 and b0 == b1 and g0 == g1
                assert v0[0][0] > 0
        # and against the oracle
        orc = oracle.Oracle(threads=8, icp_skip=skip)
        orc.set_target(d["tgt_xyz"], d["tgt_off"]); orc.set_source(d["src_xyz"], d["src_off"]); orc.set_visual(vis)
        xo, To, so = orc.frame_to_frame(d["x0"])
        assert H.pose_close(res["chain"][0][0][0], xo)
        assert [(so.solves[k].n_visual_blocks, so.solves[k].n_visual_residuals) for k in range(so.n_solves)] == res["chain"][0][0][3]


def test_kernel_times_count_every_launch_and_bracket_a_sample(hip_lib):
    """velo_set_timing(ctx, 2) / velo_get_kernel_times (what bench.py's `kernels` list is made of): every instrumented launch is counted
    by kernel name with its algorithmic bytes, every 8th is bracketed with HIP events and the time scaled up; level 3 brackets all of
    them; timing never changes a result."""
    d = synth.scan_pair(n_beams=32, n_azimuth=400)
    ref = api.Context(0, icp_skip=1)
    ref.set_target(d["tgt_xyz"], d["tgt_off"]); ref.set_source(d["src_xyz"], d["src_off"])
    x0, T0, s0 = ref.frame_to_frame(d["x0"])
    ref.close()
    evals = sum(s0.solves[k].evaluations for k in range(s0.n_solves))
    for level in (2, 3):
        c = api.Context(0, icp_skip=1)
        c.set_timing(level)
        for rep in range(3):
            c.set_target(d["tgt_xyz"], d["tgt_off"]); c.set_source(d["src_xyz"], d["src_off"])
            x, T, s = c.frame_to_frame(d["x0"])
            assert np.array_equal(x, x0) and np.array_equal(T, T0)
        kt = c.kernel_times(reset=True)
        assert c.kernel_times() == {}                                   # reset
        assert kt["assoc_search_v5_kernel"][1] == 18 and kt["grid_scatter_kernel"][1] == 3      # 3 calls x 6 rounds; one index build per load
        ms, n, b = kt["lm_iter_kernel"]
        assert n >= 3 * (evals + 6) and ms > 0.0                        # every solve takes its evaluations + 1 launches (+ the chain's margin)
        nq, nt = 32 * 400, 32 * 400
        assert kt["assoc_search_v5_kernel"][2] == 18 * (40 * nq + 12 * nt)                      # B_assoc per round served
        assert b == 3 * sum(s0.solves[k].evaluations * (36 * s0.solves[k].n_icp_valid + 224) for k in range(s0.n_solves))   # B_eval per evaluation
        assert all(v[0] >= 0.0 and v[1] > 0 for v in kt.values())
        c.close()
    # a lock-step batch logs its shared launches on the group's first context
    ctxs = [api.Context(0, icp_skip=1) for _ in range(4)]
    for c in ctxs:
        c.set_timing(2)
    xs, Ts, Ss = api.register_batch(ctxs, [(d["tgt_xyz"], d["tgt_off"])] * 4, [(d["src_xyz"], d["src_off"])] * 4, np.tile(d["x0"], (4, 1)))
    assert all(np.array_equal(xs[i], x0) for i in range(4))
    names = set()
    for c in ctxs:
        names |= set(c.kernel_times())
        c.close()
    assert "assoc_search_v5_batch_kernel" in names and any(nm.startswith("eval_step_batch") for nm in names)


def test_failed_shared_batch_leaves_sharers_in_a_defined_state(hip_lib):
    """A shared-map batch releases the sharers' hold on the old map before the owner reloads.  If the owner's load then fails (here: a
    ring table that decreases), the sharers must be left WITHOUT a target but in a defined state: the getters and the registration
    entry points report a status -- the C-ABI promises codes, never a crash -- and the contexts work again after the next good load."""
    d = H.small_pair(16, 128)
    ctxs = [api.Context(0, icp_skip=1) for _ in range(4)]
    srcs = [(d["src_xyz"], d["src_off"])] * 4
    x0s = np.tile(d["x0"], (4, 1))
    tgt = (d["tgt_xyz"], d["tgt_off"])
    good = (api.scan_refs([tgt] * 4, 0, shared=True), api.scan_refs(srcs, 0))
    xs, _, _ = api.register_batch(ctxs, None, None, x0s, refs=good)
    assert np.array_equal(ctxs[1].ring_offsets(True), d["tgt_off"])
    bad_off = np.array(d["tgt_off"], dtype=np.int32).copy()
    bad_off[3] = bad_off[2] - 1                                           # offsets decrease: velo_set_target rejects the owner's job
    bad = (api.scan_refs([(d["tgt_xyz"], bad_off)] * 4, 0, shared=True), api.scan_refs(srcs, 0))
    # (first with a fresh descriptor; further down with the SAME descriptor as the preceding good call, patched in place -- the case in
    #  which the sharers have let go of their map before the owner's load fails)
    with pytest.raises(api.VeloError):
        api.register_batch(ctxs, None, None, x0s, refs=bad)
    off_live = np.array(d["tgt_off"], dtype=np.int32).copy()
    live = (api.scan_refs([(d["tgt_xyz"], off_live)] * 4, 0, shared=True), api.scan_refs(srcs, 0))
    xs_live, _, _ = api.register_batch(ctxs, None, None, x0s, refs=live)   # a good call: ctxs 1..3 share ctx 0's map
    assert np.array_equal(xs_live, xs)
    off_a = live[0][1][0][1]                                               # the offsets array the descriptors point at
    assert np.shares_memory(off_a, off_live)
    off_a[3] = off_a[2] - 1                                                # the SAME descriptor turns bad: sharers are released, then the owner fails
    with pytest.raises(api.VeloError):
        api.register_batch(ctxs, None, None, x0s, refs=live)
    for c in ctxs[1:]:
        with pytest.raises(api.VeloError):
            c.ring_offsets(True)
        with pytest.raises(api.VeloError):
            c.cloud(True)
        with pytest.raises(api.VeloError):
            c.frame_to_frame(d["x0"])
    xs2, _, _ = api.register_batch(ctxs, None, None, x0s, refs=good)
    assert np.array_equal(xs2, xs)
    for c in ctxs:
        c.close()


def test_register_batch_visual_hands_matches_over_with_the_scans(hip_lib, oracle):
    """velo_register_batch_visual: job i's matches go into context i inside the call (seam 1 takes them per call, velo.h:599-605).  Same
    poses, solves and good-match lists as velo_set_visual per context followed by velo_register_batch; a job with zero matches registers
    LiDAR-only; the one-job form takes the single-pair path."""
    d = H.small_pair(24, 200)
    d2 = H.small_pair(24, 200, scene_seed=5)
    pairs = [d, d2, d, d2]
    vis = [synth.stereo_matches(80, seed=3 + k, mix="all", x_true=p["x_true"]) for k, p in enumerate(pairs)]
    vis[2] = None                                                                # this job: no matches
    tg = [(p["tgt_xyz"], p["tgt_off"]) for p in pairs]
    sr = [(p["src_xyz"], p["src_off"]) for p in pairs]
    x0s = np.stack([p["x0"] for p in pairs])
    a = [api.Context(0, icp_skip=1) for _ in range(4)]
    b = [api.Context(0, icp_skip=1) for _ in range(4)]
    b[2].set_visual(vis[0])                                                      # stale matches from an earlier frame: the call must replace them with none
    xa, Ta, Sa = api.register_batch(a, tg, sr, x0s, visual=api.visual_refs(vis))
    for c, m in zip(b, vis):
        c.set_visual(m)
    xb, Tb, Sb = api.register_batch(b, tg, sr, x0s)
    assert np.array_equal(xa, xb) and np.array_equal(Ta, Tb)
    for i in range(4):
        assert _summary_tuple(Sa[i]) == _summary_tuple(Sb[i])
        assert np.array_equal(a[i].good_matches(), b[i].good_matches())
        assert Sa[i].solves[0].n_visual_blocks == (0 if vis[i] is None else Sb[i].solves[0].n_visual_blocks)
    assert Sa[0].solves[0].n_visual_blocks > 0 and Sa[2].solves[0].n_visual_blocks == 0
    orc = oracle.Oracle(threads=4, icp_skip=1)
    H.load_both(api.Context(0, icp_skip=1), orc, d2, visual=vis[1])
    xo, _, so = orc.frame_to_frame(d2["x0"])
    assert H.pose_close(xa[1], xo, 1e-9, 1e-10) and so.solves[0].n_visual_blocks == Sa[1].solves[0].n_visual_blocks
    # one job
    x1, _, S1 = api.register_batch(a[:1], tg[1:2], sr[1:2], x0s[1:2], visual=api.visual_refs(vis[1:2]))
    assert H.pose_close(x1[0], xo, 1e-9, 1e-10)
    for c in a + b:
        c.close()


def test_batch_entry_points_report_bad_jobs_as_status_codes(hip_lib):
    """The drive entry points refuse what they cannot do with a status, and leave the contexts usable: a PROMOTE target on a context that
    holds no source, a negative match count, a null match pointer with a positive count, a missing count array."""
    import ctypes as C
    d = H.small_pair(16, 128)
    ctxs = [api.Context(0, icp_skip=1) for _ in range(2)]
    srcs = api.scan_refs([(d["src_xyz"], d["src_off"])] * 2, 0)
    x0s = np.tile(d["x0"], (2, 1))
    with pytest.raises(api.VeloError, match="no source cloud to promote"):
        api.register_batch(ctxs, None, None, x0s, refs=(api.promote_refs(2), srcs))
    tg = api.scan_refs([(d["tgt_xyz"], d["tgt_off"])] * 2, 0)
    ptrs, cnt, keep = api.visual_refs([None, None])
    cnt[1] = -3
    with pytest.raises(api.VeloError, match="bad visual arguments"):
        api.register_batch(ctxs, None, None, x0s, refs=(tg, srcs), visual=(ptrs, cnt, keep))
    cnt[1] = 5                                                             # five matches announced, no pointer
    with pytest.raises(api.VeloError, match="bad visual arguments"):
        api.register_batch(ctxs, None, None, x0s, refs=(tg, srcs), visual=(ptrs, cnt, keep))
    lib = ctxs[0]._lib
    arr = (C.c_void_p * 2)(*[c.handle for c in ctxs])
    x = np.zeros(12)
    assert lib.velo_register_batch_visual(arr, 2, None, None, None, None, x.ctypes.data_as(C.POINTER(C.c_double)), None, None) != 0
    # ... and the same contexts register normally afterwards, promotion included
    xs, _, _ = api.register_batch(ctxs, None, None, x0s, refs=(tg, srcs))
    xs2, _, _ = api.register_batch(ctxs, None, None, x0s, refs=(api.promote_refs(2), srcs))   # the source registered against itself
    assert np.all(np.isfinite(xs)) and np.all(np.isfinite(xs2)) and np.linalg.norm(xs2[0][3:]) < 0.05
    for c in ctxs:
        c.close()
