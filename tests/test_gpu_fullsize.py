"""-m gpu: BASELINE.json's full sizes (120k-point pair, 2M-point map).  The oracle needs seconds-to-minutes there, so
parity is checked (a) index-exact on a query SHARD (same shard rule on both sides, full target) and (b) through
size-independent properties: determinism, idempotence, shard additivity, agreement of two independent kernels
(per-lane reference kernel vs the LDS-staged shell walk), and recovery of the simulated motion."""
import os

import numpy as np
import pytest

import helpers as H
from velo_amd import api, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pair():
    return synth.scan_pair()


@pytest.fixture(scope="module")
def full_ctx(hip_lib, pair):
    c = api.Context(0, icp_skip=1)
    c.set_target(pair["tgt_xyz"], pair["tgt_off"])
    c.set_source(pair["src_xyz"], pair["src_off"])
    yield c
    c.close()


@pytest.mark.parametrize("iter_,shard", [(1, 3), (2, 11)])
def test_full_size_shard_is_index_exact(full_ctx, oracle, pair, iter_, shard):
    world = 16
    orc = oracle.Oracle(threads=8, icp_skip=1)
    orc.set_query_shard(shard, world)
    orc.set_target(pair["tgt_xyz"], pair["tgt_off"])
    orc.set_source(pair["src_xyz"], pair["src_off"])
    full_ctx.set_query_shard(shard, world)
    try:
        for x in (pair["x0"], pair["x_true"]):
            assert full_ctx.associate(x, iter_) == orc.associate(x, iter_)
            a, b = full_ctx.correspondences(), orc.correspondences()
            assert len(a) == 7500
            H.assert_corr_equal(a, b)
            c1, H1, g1 = full_ctx.evaluate(x)
            c2, H2, g2 = orc.evaluate(x)
            assert abs(c1 - c2) <= 1e-12 * c2 and H.rel_err(H1, H2) <= 1e-12 and H.rel_err(g1, g2) <= 1e-12
    finally:
        full_ctx.set_query_shard(0, 1)


def test_full_size_reference_kernel_agrees_with_shell_walk(hip_lib, diag_lib, pair):
    tables = []
    for variant in ("product", "0", "5", "4", "1"):   # the product library's kernel; then, on the diagnostics build: per-lane reference, tube, box walk with 4 waves / 1 wave
        if variant == "product":
            c = api.Context(0, icp_skip=1)
        else:
            os.environ["VELO_ASSOC_VARIANT"] = variant
            try:
                c = api.Context(0, lib=diag_lib, icp_skip=1)
            finally:
                os.environ.pop("VELO_ASSOC_VARIANT", None)
        c.set_target(pair["tgt_xyz"], pair["tgt_off"])
        c.set_source(pair["src_xyz"], pair["src_off"])
        t = []
        for it, x in ((1, pair["x0"]), (2, pair["x_true"])):
            n = c.associate(x, it)
            t.append((n, c.correspondences()))
        tables.append(t)
        c.close()
    for t in tables[1:]:
        for (n0, a), (n1, b) in zip(tables[0], t):
            assert n0 == n1 and n0 > 90000
            H.assert_corr_equal(a, b)


def test_warm_started_rounds_equal_cold_rounds(hip_lib, diag_lib, pair, monkeypatch):
    """Rounds after the first start from the previous round's winners (AssocOut::prev).  Seeds are ordinary candidates entered
    early, so every table must equal the one of a context that never has seeds -- also when the pose jumps, the gate shrinks
    (iter 2), grows again, or the source / target is replaced in between."""
    warm = api.Context(0, icp_skip=1)                        # the product library (warm start is what it does)
    monkeypatch.setenv("VELO_WARM_START", "0")                # the switch exists in the diagnostics build only
    cold = api.Context(0, lib=diag_lib, icp_skip=1)
    for c in (warm, cold):
        c.set_target(pair["tgt_xyz"], pair["tgt_off"])
        c.set_source(pair["src_xyz"], pair["src_off"])
    x0, x1 = pair["x0"], pair["x_true"]
    seq = [(1, x0), (1, x0 + 0.3 * (x1 - x0)), (1, x0 + 0.9 * (x1 - x0)), (2, x1), (2, x1 + 1e-4), (2, x1),
           (1, np.array([0.02, -0.01, 0.03, 0.4, -0.3, 1.5])), (2, x0), (1, x1)]
    for it, x in seq:
        assert warm.associate(x, it) == cold.associate(x, it)
        assert warm.correspondences().tobytes() == cold.correspondences().tobytes()
    # swap the roles of the two scans: stale seeds must not survive a new source / target
    for c in (warm, cold):
        c.set_target(pair["src_xyz"], pair["src_off"])
        c.set_source(pair["tgt_xyz"], pair["tgt_off"])
    for it, x in ((1, np.zeros(6)), (2, np.zeros(6))):
        assert warm.associate(x, it) == cold.associate(x, it)
        assert warm.correspondences().tobytes() == cold.correspondences().tobytes()
    xa, Ta, sa = warm.frame_to_frame(np.zeros(6))
    xb, Tb, sb = cold.frame_to_frame(np.zeros(6))
    assert np.array_equal(xa, xb) and np.array_equal(Ta, Tb)
    warm.close()
    cold.close()


def test_full_size_frame_to_frame_properties(full_ctx, pair):
    x1, T1, s1 = full_ctx.frame_to_frame(pair["x0"])
    x2, T2, s2 = full_ctx.frame_to_frame(pair["x0"])
    assert np.array_equal(x1, x2) and np.array_equal(T1, T2)                      # deterministic, bit for bit
    assert s1.n_solves == 6 and s1.n_queries == 120000 and s1.n_target == 120000
    assert all(s1.solves[k].termination == 0 for k in range(6))
    assert s1.solves[5].final_cost < s1.solves[0].initial_cost
    # noise-limited recovery of the simulated motion (sigma = 2 cm range noise)
    assert np.linalg.norm(x1[3:] - pair["x_true"][3:]) < 3e-3 and np.linalg.norm(x1[:3] - pair["x_true"][:3]) < 5e-4
    # T is the matrix of x (utility.h:67-82) and inverts back
    np.testing.assert_allclose(api.pose_mat_to_vec(T1), x1, atol=1e-12)
    # SURVEY.md 8(d) byte accounting
    want = 6 * (40 * 120000 + 12 * 120000) + sum(s1.solves[k].evaluations * (36 * s1.solves[k].n_icp_valid + 224) for k in range(6))
    assert s1.algorithmic_bytes == want
    # a second association at the solution is idempotent
    n1 = full_ctx.associate(x1, 2)
    a = full_ctx.correspondences()
    n2 = full_ctx.associate(x1, 2)
    assert n1 == n2 and np.array_equal(a, full_ctx.correspondences())


def test_full_size_device_resident_inputs_and_stride16(hip_lib, pair, full_ctx):
    """velo_set_target/source accept device pointers and pcl::PointXYZ's 16-byte stride (zero-copy contract)."""
    import torch
    dev = torch.device("cuda", 0)
    t4 = torch.zeros((120000, 4), dtype=torch.float32, device=dev)
    t4[:, :3] = torch.from_numpy(pair["tgt_xyz"]).to(dev)
    s3 = torch.from_numpy(pair["src_xyz"]).to(dev)
    torch.cuda.synchronize()
    c = api.Context(0, icp_skip=1)
    c.set_target(t4, pair["tgt_off"])
    c.set_source(s3, pair["src_off"])
    n = c.associate(pair["x_true"], 1)
    full_ctx.set_query_shard(0, 1)
    assert n == full_ctx.associate(pair["x_true"], 1)
    assert np.array_equal(c.correspondences(), full_ctx.correspondences())
    c.close()


def test_batch_of_contexts_matches_single(hip_lib, pair):
    ctxs = [api.Context(0, icp_skip=4) for _ in range(3)]
    for c in ctxs:
        c.set_target(pair["tgt_xyz"], pair["tgt_off"])
        c.set_source(pair["src_xyz"], pair["src_off"])
    xs, Ts, Ss = api.frame_to_frame_batch(ctxs, np.tile(pair["x0"], (3, 1)))
    x_single, _, _ = ctxs[0].frame_to_frame(pair["x0"])
    for k in range(3):
        assert np.array_equal(xs[k], x_single) and Ss[k].n_solves == 6
    for c in ctxs:
        c.close()


def test_scan_to_map_config4_shard_parity_and_solve(hip_lib, oracle):
    """BASELINE configs[3]: 120k-point scan against a 2M-point accumulated map (1067 rings)."""
    m = synth.scan_to_map(2_000_000)
    assert m["tgt_xyz"].shape[0] == 2_000_000 and len(m["tgt_off"]) == 1068
    c = api.Context(0, icp_skip=1)
    c.set_target(m["tgt_xyz"], m["tgt_off"])
    c.set_source(m["src_xyz"], m["src_off"])
    world, shard = 64, 17
    orc = oracle.Oracle(threads=8, icp_skip=1)
    orc.set_query_shard(shard, world)
    orc.set_target(m["tgt_xyz"], m["tgt_off"])
    orc.set_source(m["src_xyz"], m["src_off"])
    c.set_query_shard(shard, world)
    for it, x in ((1, m["x0"]), (2, m["x_true"])):
        assert c.associate(x, it) == orc.associate(x, it)
        H.assert_corr_equal(c.correspondences(), orc.correspondences())
    c.set_query_shard(0, 1)
    x, T, s = c.frame_to_frame(m["x0"])
    assert s.n_target == 2_000_000 and all(s.solves[k].termination == 0 for k in range(6))
    assert np.linalg.norm(x[3:] - m["x_true"][3:]) < 8e-3 and np.linalg.norm(x[:3] - m["x_true"][:3]) < 1e-3   # map noise
    c.close()


def test_batch_entry_point_equals_single_calls(hip_lib, monkeypatch):
    """velo_frame_to_frame_batch advances the contexts in lock-step with shared LM launches; every context must get exactly what
    a call of its own gives (same kernels' arithmetic, different launch structure), also with different pairs in one batch,
    and the thread-per-context fallback must agree too."""
    pairs = [H.small_pair(16, 128), H.small_pair(32, 200), H.small_pair(16, 96, scene_seed=3), H.small_pair(24, 160), H.small_pair(16, 128)]
    # two of the five contexts also carry stereo matches (the visual sweep joins the shared launches)
    visuals = [None, api.matches_from_dict(synth.stereo_matches(40, mix="all")), None, api.matches_from_dict(synth.stereo_matches(25, seed=8)), None]
    singles = []
    for d, v in zip(pairs, visuals):
        c = api.Context(0, icp_skip=1)
        c.set_target(d["tgt_xyz"], d["tgt_off"]); c.set_source(d["src_xyz"], d["src_off"])
        if v is not None:
            c.set_visual(v)
        singles.append(c.frame_to_frame(d["x0"]))
        c.close()
    for lockstep in ("1", "0"):
        monkeypatch.setenv("VELO_BATCH_LOCKSTEP", lockstep)
        ctxs = [api.Context(0, icp_skip=1) for _ in pairs]
        for c, d, v in zip(ctxs, pairs, visuals):
            c.set_target(d["tgt_xyz"], d["tgt_off"]); c.set_source(d["src_xyz"], d["src_off"])
            if v is not None:
                c.set_visual(v)
        for rep in range(2):                                  # second call: warm-start tables and chunk predictions carry over
            xs, Ts, Ss = api.frame_to_frame_batch(ctxs, [d["x0"] for d in pairs])
            for i, (x1, T1, s1) in enumerate(singles):
                assert np.array_equal(xs[i], x1), (lockstep, rep, i)
                assert np.array_equal(Ts[i], T1)
                assert Ss[i].n_solves == s1.n_solves
                for k in range(s1.n_solves):
                    a, b = Ss[i].solves[k], s1.solves[k]
                    assert (a.termination, a.lm_iterations, a.evaluations, a.n_icp_valid, a.n_visual_blocks, a.n_visual_residuals) == \
                        (b.termination, b.lm_iterations, b.evaluations, b.n_icp_valid, b.n_visual_blocks, b.n_visual_residuals)
                    assert a.final_cost == b.final_cost and a.initial_cost == b.initial_cost
                assert Ss[i].algorithmic_bytes == s1.algorithmic_bytes
        for c in ctxs:
            c.close()


def test_asker_list_in_single_calls_and_batches_equals_in_place_search(hip_lib, diag_lib, monkeypatch):
    """Density-shrunk grid (VELO_DENSE_REF forces it on small clouds): the cold round's asking queries are searched in place
    (VELO_ASKER_QUEUE=0), from the list by assoc_asker_kernel in single calls (default) or also in lock-step batches (=2): same poses,
    same solves, bit for bit."""
    pairs = [H.small_pair(16, 128), H.small_pair(32, 200), H.small_pair(16, 96, scene_seed=3)]
    monkeypatch.setenv("VELO_DENSE_REF", "300")
    res = {}
    for q in ("0", "1", "2"):
        monkeypatch.setenv("VELO_ASKER_QUEUE", q)
        ctxs = [api.Context(0, lib=diag_lib, icp_skip=1) for _ in pairs]
        single = []
        for c, d in zip(ctxs, pairs):
            c.set_target(d["tgt_xyz"], d["tgt_off"]); c.set_source(d["src_xyz"], d["src_off"])
            x, T, s = c.frame_to_frame(d["x0"])
            single.append((x.copy(), [(s.solves[k].evaluations, s.solves[k].n_icp_valid, s.solves[k].final_cost) for k in range(s.n_solves)]))
        for c, d in zip(ctxs, pairs):
            c.set_source(d["src_xyz"], d["src_off"])          # seeds cleared: the batch starts with a cold round again
        xs, Ts, Ss = api.frame_to_frame_batch(ctxs, [d["x0"] for d in pairs])
        batch = [(xs[i].copy(), [(Ss[i].solves[k].evaluations, Ss[i].solves[k].n_icp_valid, Ss[i].solves[k].final_cost) for k in range(Ss[i].n_solves)]) for i in range(len(pairs))]
        res[q] = (single, batch)
        for c in ctxs:
            c.close()
    for q in ("1", "2"):
        for part in (0, 1):
            for (x0, s0), (x1, s1) in zip(res["0"][part], res[q][part]):
                assert np.array_equal(x0, x1) and s0 == s1, (q, part)
    for (x0, s0), (x1, s1) in zip(res["0"][0], res["0"][1]):
        assert np.array_equal(x0, x1) and s0 == s1


def test_register_batch_equals_separate_uploads_and_batch(hip_lib):
    """velo_register_batch (scans handed over with the call, indexed on the group threads) against velo_set_target + velo_set_source +
    velo_frame_to_frame_batch: same poses, same summaries, bit for bit; host and device-resident inputs; targets only / sources only."""
    import torch
    d = synth.scan_pair(n_beams=32, n_azimuth=900)
    n = 5
    x0s = np.tile(d["x0"], (n, 1)) + 1e-3 * np.arange(n)[:, None]
    a = [api.Context(0, icp_skip=1) for _ in range(n)]
    b = [api.Context(0, icp_skip=1) for _ in range(n)]
    for c in a:
        c.set_target(d["tgt_xyz"], d["tgt_off"]); c.set_source(d["src_xyz"], d["src_off"])
    xa, Ta, Sa = api.frame_to_frame_batch(a, x0s)
    xb, Tb, Sb = api.register_batch(b, [(d["tgt_xyz"], d["tgt_off"])] * n, [(d["src_xyz"], d["src_off"])] * n, x0s)
    assert np.array_equal(xa, xb) and np.array_equal(Ta, Tb)
    for s1, s2 in zip(Sa, Sb):
        assert [s1.solves[k].lm_iterations for k in range(s1.n_solves)] == [s2.solves[k].lm_iterations for k in range(s2.n_solves)]
    # device-resident scans, and only the sources replaced (the targets stay in the contexts)
    src_dev = torch.from_numpy(d["src_xyz"]).to("cuda:0")
    xc, _, _ = api.register_batch(b, None, [(src_dev, d["src_off"])] * n, x0s)
    assert np.array_equal(xc, xa)
    tgt_dev = torch.from_numpy(d["tgt_xyz"]).to("cuda:0")
    xd, _, _ = api.register_batch(b, [(tgt_dev, d["tgt_off"])] * n, None, x0s)
    assert np.array_equal(xd, xa)
    # ONE job: velo_register_batch takes the single-pair path (set_target + set_source + frame_to_frame in one call)
    x1, T1, S1 = api.register_batch(b[:1], [(d["tgt_xyz"], d["tgt_off"])], [(src_dev, d["src_off"])], x0s[:1])
    assert np.array_equal(x1[0], xa[0]) and np.array_equal(T1[0], Ta[0])
    assert [S1[0].solves[k].lm_iterations for k in range(S1[0].n_solves)] == [Sa[0].solves[k].lm_iterations for k in range(Sa[0].n_solves)]
    # a bad job is reported, not ignored
    bad_off = d["tgt_off"].copy(); bad_off[3] = bad_off[2]
    with pytest.raises(api.VeloError):
        api.register_batch(b, [(d["tgt_xyz"], bad_off)] * n, None, x0s)
    with pytest.raises(api.VeloError):
        api.register_batch(b[:1], [(d["tgt_xyz"], bad_off)], None, x0s[:1])
    for c in a + b:
        c.close()


def test_batch_of_unequal_scans_equals_single_calls(hip_lib):
    """One lock-step batch over scans of different sizes (the merged association launch takes its grid from the largest, the sweeps
    from the largest block count) and five contexts (launches of 4 + 1): every pose, table and iteration count as in single calls."""
    shapes = [(16, 300), (64, 700), (8, 200), (32, 400), (48, 512)]
    data = [synth.scan_pair(n_beams=b, n_azimuth=a) for b, a in shapes]
    single, batch = [], []
    for d in data:
        c = api.Context(0, icp_skip=1)
        c.set_target(d["tgt_xyz"], d["tgt_off"]); c.set_source(d["src_xyz"], d["src_off"])
        x, T, S = c.frame_to_frame(d["x0"])
        single.append((x, [S.solves[k].lm_iterations for k in range(S.n_solves)], S.solves[S.n_solves - 1].n_icp_valid))
        c.close()
        batch.append(api.Context(0, icp_skip=1))
    xs, Ts, Ss = api.register_batch(batch, [(d["tgt_xyz"], d["tgt_off"]) for d in data], [(d["src_xyz"], d["src_off"]) for d in data],
                                    [d["x0"] for d in data])
    for i, (x, its, nv) in enumerate(single):
        assert np.array_equal(xs[i], x), i
        assert [Ss[i].solves[k].lm_iterations for k in range(Ss[i].n_solves)] == its
        assert Ss[i].solves[Ss[i].n_solves - 1].n_icp_valid == nv
    for c in batch:
        c.close()


# ---- BASELINE configs at full size against the oracle (the oracle needs ~1.5-3 s per pair on the GPU box's host cores) ----------
def _solve_counts(s):
    return [(s.solves[k].termination, s.solves[k].lm_iterations, s.solves[k].evaluations, s.solves[k].n_icp_valid,
             s.solves[k].n_visual_blocks, s.solves[k].n_visual_residuals) for k in range(s.n_solves)]


def test_config2_full_pair_pose_and_solves_match_oracle(full_ctx, oracle, pair):
    """configs[1]: the whole 120k x 120k registration (6 association rounds + 6 LM solves) against the CPU restatement: pose
    within the north_star tolerance (1e-4 m / 1e-5 rad), every solve with the same termination, iteration, evaluation and
    valid-correspondence counts, the same algorithmic byte count."""
    orc = oracle.Oracle(threads=oracle.max_threads(), icp_skip=1)
    orc.set_target(pair["tgt_xyz"], pair["tgt_off"])
    orc.set_source(pair["src_xyz"], pair["src_off"])
    xo, To, so = orc.frame_to_frame(pair["x0"])
    full_ctx.set_source(pair["src_xyz"], pair["src_off"])          # fresh seeds, like a new frame
    x, T, s = full_ctx.frame_to_frame(pair["x0"])
    assert H.pose_close(x, xo, 1e-4, 1e-5), (x, xo)
    assert s.n_solves == so.n_solves == 6 and _solve_counts(s) == _solve_counts(so)
    assert s.algorithmic_bytes == so.algorithmic_bytes and s.n_queries == 120000
    np.testing.assert_allclose(T, To, atol=1e-6)
    # the simulated motion is recovered too (not a parity statement: a sanity bound on the workload itself)
    assert np.linalg.norm(x[3:] - pair["x_true"][3:]) < 0.02 and np.linalg.norm(x[:3] - pair["x_true"][:3]) < 2e-3


def test_config4_full_pair_pose_and_solves_match_oracle(hip_lib, oracle):
    """configs[3]: the WHOLE scan-to-map call -- 120k queries against the 2M-point map's 1,067 rings (velo.h:800-903 on every ring), six
    association rounds + six LM solves -- against the CPU restatement: pose within the north_star tolerance (1e-4 m / 1e-5 rad), every
    solve with the same termination, iteration, evaluation and valid-correspondence counts, the same algorithmic byte count; the lock-step
    batch entry with the map SHARED by two contexts gives the same bits as the single call."""
    m = synth.scan_to_map(2_000_000)
    orc = oracle.Oracle(threads=oracle.max_threads(), icp_skip=1)
    orc.set_target(m["tgt_xyz"], m["tgt_off"])
    orc.set_source(m["src_xyz"], m["src_off"])
    xo, To, so = orc.frame_to_frame(m["x0"])
    ctxs = [api.Context(0, icp_skip=1) for _ in range(2)]
    try:
        c = ctxs[0]
        c.set_target(m["tgt_xyz"], m["tgt_off"]); c.set_source(m["src_xyz"], m["src_off"])
        x, T, s = c.frame_to_frame(m["x0"])
        assert H.pose_close(x, xo, 1e-4, 1e-5), (x, xo)
        assert s.n_solves == so.n_solves == 6 and _solve_counts(s) == _solve_counts(so)
        assert s.algorithmic_bytes == so.algorithmic_bytes and s.n_queries == 120000 and s.n_target == 2_000_000
        np.testing.assert_allclose(T, To, atol=1e-6)
        refs = (api.scan_refs([(m["tgt_xyz"], m["tgt_off"])] * 2, 0, shared=True), api.scan_refs([(m["src_xyz"], m["src_off"])] * 2, 0))
        xs, Ts, Ss = api.register_batch(ctxs, None, None, np.tile(m["x0"], (2, 1)), refs=refs)
        for i in range(2):
            assert np.array_equal(xs[i], x) and _solve_counts(Ss[i]) == _solve_counts(s)
    finally:
        for c in ctxs:
            c.close()


def test_config3_full_pair_with_2000_stereo_blocks_matches_oracle(hip_lib, oracle, pair):
    """configs[2]: the 120k pair + 1,000 matches x 2 cameras (2,000 reprojection blocks, 10 % gross outliers): pose, the outlier
    gate's good_matches list after the call (iter 2) and every solve's counts equal the oracle's; single call and lock-step
    batch entry agree bit for bit."""
    vis = synth.stereo_matches(1000)
    orc = oracle.Oracle(threads=oracle.max_threads(), icp_skip=1)
    orc.set_target(pair["tgt_xyz"], pair["tgt_off"])
    orc.set_source(pair["src_xyz"], pair["src_off"])
    orc.set_visual(vis)
    xo, To, so = orc.frame_to_frame(pair["x0"])
    go = orc.good_matches()
    ctxs = [api.Context(0, icp_skip=1) for _ in range(2)]
    try:
        for c in ctxs:
            c.set_target(pair["tgt_xyz"], pair["tgt_off"]); c.set_source(pair["src_xyz"], pair["src_off"]); c.set_visual(vis)
        x, T, s = ctxs[0].frame_to_frame(pair["x0"])
        assert H.pose_close(x, xo, 1e-4, 1e-5), (x, xo)
        assert _solve_counts(s) == _solve_counts(so)
        assert s.solves[0].n_visual_blocks == 2000 and s.solves[3].n_visual_blocks < 2000      # iter 2 gates the outliers out
        assert np.array_equal(ctxs[0].good_matches(), go)
        assert s.algorithmic_bytes == so.algorithmic_bytes
        for c in ctxs:
            c.set_source(pair["src_xyz"], pair["src_off"])
        xs, Ts, Ss = api.frame_to_frame_batch(ctxs, [pair["x0"], pair["x0"]])
        for i in range(2):
            assert np.array_equal(xs[i], x) and _solve_counts(Ss[i]) == _solve_counts(s)
    finally:
        for c in ctxs:
            c.close()
