"""-m gpu: target-sharded mode of SURVEY.md 8(e) / BASELINE config 5, verified on ONE GPU by running the shards one after
the other: W contexts each hold a block of whole target rings, search all queries against it, and the query's owner merges
the W per-query top-2 records.  The merged table must equal the oracle's association against the WHOLE target, and a
frame-to-frame loop driven through the pieces must reproduce the oracle's pose."""
import numpy as np
import pytest

import helpers as H
from velo_amd import api, shard, synth

pytestmark = pytest.mark.gpu


def make_shards(d, world, **params):
    ctxs = []
    for r in range(world):
        r0, r1, p0, local = shard.target_ring_block(d["tgt_off"], r, world)
        c = api.Context(0, **params)
        c.set_target_part(d["tgt_xyz"][p0:p0 + local[-1]], local, r0, p0)
        c.set_source(d["src_xyz"], d["src_off"])
        ctxs.append(c)
    return ctxs


@pytest.mark.parametrize("world", [2, 3, 5])
def test_merged_partials_equal_full_association(hip_lib, oracle, world):
    d = H.small_pair(16, 128)
    orc = oracle.Oracle(threads=4, icp_skip=1)
    orc.set_target(d["tgt_xyz"], d["tgt_off"])
    orc.set_source(d["src_xyz"], d["src_off"])
    ctxs = make_shards(d, world, icp_skip=1)
    try:
        for it, x in ((1, d["x0"]), (1, d["x_true"]), (2, d["x_true"]), (1, [0, 0, 0, 300.0, 0, 0])):
            tables = []
            for c in ctxs:
                c.associate_partial(x, it)
                tables.append(c.partials())
            assert all(len(t) == 16 * 128 for t in tables)
            n_cpu = orc.associate(x, it)
            want = orc.correspondences()
            for owner in (0, world - 1):
                n_gpu = ctxs[owner].merge_partials(tables)
                H.assert_corr_equal(ctxs[owner].correspondences(), want)
                assert n_gpu == n_cpu
                c1, H1, g1 = ctxs[owner].evaluate(x)
                c2, H2, g2 = orc.evaluate(x)
                assert abs(c1 - c2) <= 1e-12 * max(c2, 1e-300) and H.rel_err(H1, H2) <= 1e-12
    finally:
        for c in ctxs:
            c.close()


def test_owner_merges_only_its_query_share(hip_lib, oracle):
    """Config 5 data flow: rank r finishes the queries [Nq r/W, Nq (r+1)/W) from everybody's records; the shares tile the table."""
    d = H.small_pair(16, 96)
    world = 4
    orc = oracle.Oracle(threads=4, icp_skip=1)
    orc.set_target(d["tgt_xyz"], d["tgt_off"])
    orc.set_source(d["src_xyz"], d["src_off"])
    orc.associate(d["x_true"], 1)
    want = orc.correspondences()
    ctxs = make_shards(d, world, icp_skip=1)
    try:
        tables = []
        for c in ctxs:
            c.associate_partial(d["x_true"], 1)
            tables.append(c.partials())
        parts, acc = [], [0.0, np.zeros((6, 6)), np.zeros(6)]
        for r, c in enumerate(ctxs):
            c.set_query_shard(r, world)
            c.merge_partials(tables)
            parts.append(c.correspondences())
            cost, Hm, g = c.evaluate(d["x_true"])
            acc[0] += cost
            acc[1] += Hm
            acc[2] += g
        H.assert_corr_equal(np.concatenate(parts), want)
        c2, H2, g2 = orc.evaluate(d["x_true"])
        assert abs(acc[0] - c2) <= 1e-12 * c2 and H.rel_err(acc[1], H2) <= 1e-12 and H.rel_err(acc[2], g2) <= 1e-12
    finally:
        for c in ctxs:
            c.close()


def test_target_sharded_frame_to_frame_loop(hip_lib, oracle):
    """frameToFrame's loop (velo.h:616-910) with the association done shard by shard: same pose as the oracle."""
    d = H.small_pair(16, 128)
    world = 3
    orc = oracle.Oracle(threads=4, icp_skip=1)
    orc.set_target(d["tgt_xyz"], d["tgt_off"])
    orc.set_source(d["src_xyz"], d["src_off"])
    x_orc, _, s_orc = orc.frame_to_frame(d["x0"])
    ctxs = make_shards(d, world, icp_skip=1)
    try:
        x = d["x0"].copy()
        k = 0
        for it in (1, 2):
            for _ in range(3):
                tables = []
                for c in ctxs:
                    c.associate_partial(x, it)
                    tables.append(c.partials())
                nv = ctxs[0].merge_partials(tables)
                x, s = ctxs[0].solve(x)
                assert nv == s_orc.solves[k].n_icp_valid and s.evaluations == s_orc.solves[k].evaluations
                k += 1
        assert H.pose_close(x, x_orc), (x, x_orc)
    finally:
        for c in ctxs:
            c.close()


def test_single_rank_comm_target_sharded_path(hip_lib, oracle):
    """The RCCL code path of config 5 (partial search -> all-to-all via grouped send/recv -> merge) with a 1-rank
    communicator: executes the same calls as N ranks; must equal the plain path."""
    d = H.small_pair(16, 128)
    c = api.Context(0, icp_skip=1)
    c.set_target(d["tgt_xyz"], d["tgt_off"])
    c.set_source(d["src_xyz"], d["src_off"])
    x_plain, _, s_plain = c.frame_to_frame(d["x0"])
    c.comm_init(api.comm_unique_id(), 0, 1)
    c.comm_set_target_sharded(True)
    try:
        x_ts, _, s_ts = c.frame_to_frame(d["x0"])
        n = c.associate(d["x_true"], 1)
        tab = c.correspondences()
    finally:
        c.comm_set_target_sharded(False)
        c.comm_destroy()
    # same correspondences, same LM decisions; the plain path keeps its query list in patch order, the record-exchanging path in ring
    # order, so the 28 sums of an evaluation are added up in a different order: equal to rounding, not bit for bit
    assert np.abs(x_plain - x_ts).max() <= 1e-11
    assert [(s_ts.solves[k].lm_iterations, s_ts.solves[k].evaluations) for k in range(6)] == [(s_plain.solves[k].lm_iterations, s_plain.solves[k].evaluations) for k in range(6)]
    assert [s_ts.solves[k].n_icp_valid for k in range(6)] == [s_plain.solves[k].n_icp_valid for k in range(6)]
    assert n == c.associate(d["x_true"], 1)
    H.assert_corr_equal(tab, c.correspondences())
    c.close()


def test_map_config5_shape_ring_blocks(hip_lib, oracle):
    """2M-point map split into 8 ring blocks (config 5's shape), one query shard checked index-exact against the oracle."""
    m = synth.scan_to_map(400_000)        # 214 rings; enough to exercise uneven ring blocks quickly
    world = 8
    orc = oracle.Oracle(threads=8, icp_skip=1)
    orc.set_query_shard(5, 16)
    orc.set_target(m["tgt_xyz"], m["tgt_off"])
    orc.set_source(m["src_xyz"], m["src_off"])
    orc.associate(m["x_true"], 1)
    want = orc.correspondences()
    ctxs = make_shards(m, world, icp_skip=1)
    try:
        tables = []
        for c in ctxs:
            c.associate_partial(m["x_true"], 1)
            tables.append(c.partials())
        ctxs[3].set_query_shard(5, 16)
        ctxs[3].merge_partials(tables)
        H.assert_corr_equal(ctxs[3].correspondences(), want)
    finally:
        for c in ctxs:
            c.close()
