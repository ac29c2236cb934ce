"""N > 1 on CPU: two gloo ranks exercise the query-shard partition and the 28-double all-reduce layout that the
RCCL path of the library uses; the evaluator on each rank is the oracle (the checker)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import oracle_lib as ol
    import velo_amd  # noqa: F401
    from velo_amd import shard, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = synth.scan_pair(n_beams=8, n_azimuth=64)
    o = ol.Oracle(icp_skip=1)
    o.set_query_shard(rank, world)
    o.set_target(d["tgt_xyz"], d["tgt_off"])
    o.set_source(d["src_xyz"], d["src_off"])
    x = d["x0"].copy()
    nq_total = 8 * 64
    lo, hi = shard.query_shard_range(nq_total, rank, world)
    # three Gauss-Newton style rounds: associate shard, evaluate shard, all-reduce the 28 doubles, identical update
    for _ in range(3):
        o.associate(x, 1)
        assert len(o.correspondences()) == hi - lo
        cost, Hm, g = o.evaluate(x)
        block = torch.from_numpy(shard.pack_normal_equations(cost, Hm, g))
        dist.all_reduce(block, op=dist.ReduceOp.SUM)
        cost, Hm, g = shard.unpack_normal_equations(block.numpy())
        x = x - np.linalg.solve(Hm + 1e-9 * np.eye(6), g)
    gathered = [torch.zeros(6, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(gathered, torch.from_numpy(x))
    np.save(os.path.join(out_dir, f"x_{rank}.npy"), np.stack([t.numpy() for t in gathered]))
    np.save(os.path.join(out_dir, f"blk_{rank}.npy"), block.numpy())
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_query_sharding_matches_single_rank(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    xs = [np.load(tmp_path / f"x_{r}.npy") for r in range(world)]
    assert np.array_equal(xs[0], xs[1]) and np.array_equal(xs[0][0], xs[0][1])      # every rank took the same steps
    # single-process reference of the same three rounds
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    from velo_amd import shard, synth
    d = synth.scan_pair(n_beams=8, n_azimuth=64)
    o = ol.Oracle(icp_skip=1)
    o.set_target(d["tgt_xyz"], d["tgt_off"])
    o.set_source(d["src_xyz"], d["src_off"])
    x = d["x0"].copy()
    for _ in range(3):
        o.associate(x, 1)
        cost, Hm, g = o.evaluate(x)
        blk = shard.pack_normal_equations(cost, Hm, g)
        x = x - np.linalg.solve(Hm + 1e-9 * np.eye(6), g)
    np.testing.assert_allclose(xs[0][0], x, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(np.load(tmp_path / "blk_0.npy"), blk, rtol=1e-10)


def test_shard_ranges_tile_the_query_list():
    from velo_amd import shard
    for nq in (0, 1, 7, 640, 120000):
        for world in (1, 2, 3, 8):
            edges = [shard.query_shard_range(nq, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == nq
            assert all(edges[r][1] == edges[r + 1][0] for r in range(world - 1))
            assert max(b - a for a, b in edges) - min(b - a for a, b in edges) <= 1
    blk = shard.pack_normal_equations(2.5, np.arange(36.0).reshape(6, 6) + np.arange(36.0).reshape(6, 6).T, np.arange(6.0))
    c, Hm, g = shard.unpack_normal_equations(blk)
    assert c == 2.5 and np.array_equal(g, np.arange(6.0)) and np.array_equal(Hm, Hm.T)
    # wire layout = the device's: 21 upper-triangular entries row by row (tri(i, j) = 6 i - i (i - 1) / 2 + (j - i)), 6 x J^T r, cost
    tri = lambda i, j: i * 6 - (i * (i - 1)) // 2 + (j - i)   # noqa: E731
    Hfull = np.arange(36.0).reshape(6, 6) + np.arange(36.0).reshape(6, 6).T
    for i in range(6):
        for j in range(i, 6):
            assert blk[tri(i, j)] == Hfull[i, j] == Hm[i, j] == Hm[j, i]
    assert np.array_equal(blk[21:27], np.arange(6.0)) and blk[27] == 2.5 and len(blk) == 28


# ---- BASELINE config 5 on CPU: the per-round record exchange and the owner's merge rule, two gloo ranks --------------------------
def _partial_table(ol, d, x, it, r0, r1, first_point):
    """What one rank knows after searching ITS whole rings [r0, r1) for every query (the oracle as the per-shard searcher):
    key1 / key2 = (float bits of d^2) << 32 | global ring-major index, or all ones when absent; rings and the three plane points."""
    off = d["tgt_off"]
    local = (off[r0:r1 + 1] - off[r0]).astype(np.int32)
    o = ol.Oracle(icp_skip=1)
    o.set_target(d["tgt_xyz"][off[r0]:off[r1]], local)
    o.set_source(d["src_xyz"], d["src_off"])
    o.associate(x, it)
    c = o.correspondences()
    n = len(c)
    inf = np.uint64(0xFFFFFFFFFFFFFFFF)
    has1, has2 = c["ring_i"] >= 0, c["ring_j"] >= 0
    g1 = (first_point + local[np.maximum(c["ring_i"], 0)] + c["idx_i"]).astype(np.uint64)
    g2 = (first_point + local[np.maximum(c["ring_j"], 0)] + c["idx_j"]).astype(np.uint64)
    key1 = np.where(has1, (c["dist_i"].view(np.uint32).astype(np.uint64) << np.uint64(32)) | g1, inf)
    key2 = np.where(has2, (c["dist_j"].view(np.uint32).astype(np.uint64) << np.uint64(32)) | g2, inf)
    pts = d["tgt_xyz"][off[r0]:off[r1]]
    # the ring neighbour of the best point that is nearer to the transformed query (velo.h:852-863) -- the oracle reports it only
    # when the shard itself holds two rings in range, the record needs it whenever there is a best point: restated here in float32
    q = np.stack([ol.transform_point(p, x) for p in d["src_xyz"]]).astype(np.float32)
    ring_l = np.maximum(c["ring_i"], 0)
    n_ring = (local[ring_l + 1] - local[ring_l]).astype(np.int64)
    k1 = (c["idx_i"] + 1) % n_ring
    k2 = (c["idx_i"] - 1 + n_ring) % n_ring

    def dist2(a, b):
        dd = (a - b).astype(np.float32)
        r = (dd[:, 0] * dd[:, 0]).astype(np.float32)
        r = (r + (dd[:, 1] * dd[:, 1]).astype(np.float32)).astype(np.float32)
        return (r + (dd[:, 2] * dd[:, 2]).astype(np.float32)).astype(np.float32)
    d1 = dist2(pts[local[ring_l] + k1], q)
    d2 = dist2(pts[local[ring_l] + k2], q)
    idx_k = np.where(d1 < d2, k1, k2).astype(np.int32)
    v0 = pts[np.where(has1, local[ring_l] + c["idx_i"], 0)]
    v2 = pts[np.where(has1, local[ring_l] + idx_k, 0)]
    v1 = pts[np.where(has2, local[np.maximum(c["ring_j"], 0)] + c["idx_j"], 0)]
    out = {"key1": key1, "key2": key2, "ring1": np.where(has1, c["ring_i"] + r0, -1), "ring2": np.where(has2, c["ring_j"] + r0, -1),
           "idx1": c["idx_i"], "idx_k": idx_k, "idx2": c["idx_j"], "v0": v0, "v2": v2, "v1": v1}
    return out


def _merge_tables(tabs):
    """merge_partials_kernel restated in numpy: rings are disjoint across ranks, so best1 = min key1 over ranks (rank w*), best2 =
    min(key1 of the other ranks, key2 of rank w*); ties cannot happen (keys carry the global index)."""
    k1 = np.stack([t["key1"] for t in tabs])             # [W, n]
    k2 = np.stack([t["key2"] for t in tabs])
    w1 = np.argmin(k1, axis=0)
    n = k1.shape[1]
    cols = np.arange(n)
    best1 = k1[w1, cols]
    others = k1.copy()
    others[w1, cols] = np.uint64(0xFFFFFFFFFFFFFFFF)
    w2o = np.argmin(others, axis=0)
    cand_other = others[w2o, cols]
    cand_same = k2[w1, cols]
    second_is_key1 = cand_other <= cand_same              # equal only when both are absent
    best2 = np.where(second_is_key1, cand_other, cand_same)
    w2 = np.where(second_is_key1, w2o, w1)
    return best1, best2, w1, w2, second_is_key1


def _exchange_worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import oracle_lib as ol
    import velo_amd  # noqa: F401
    from velo_amd import shard, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = synth.scan_pair(n_beams=16, n_azimuth=96)
    nq = 16 * 96
    r0, r1, p0, _ = shard.target_ring_block(d["tgt_off"], rank, world)
    mine = _partial_table(ol, d, d["x_true"], 1, r0, r1, p0)
    # all-to-all of the record slices: rank r receives from everybody the records of ITS query share (velo_hip.hip
    # associate_target_sharded: grouped ncclSend / ncclRecv of velo_partial records)
    fields = ["key1", "key2", "ring1", "ring2", "idx1", "idx_k", "idx2", "v0", "v2", "v1"]
    lo, hi = shard.query_shard_range(nq, rank, world)
    got = [dict() for _ in range(world)]
    for f in fields:
        a = np.ascontiguousarray(mine[f])
        as_i64 = a.view(np.int64) if a.dtype == np.uint64 else a.astype(np.int64) if a.dtype.kind == "i" else None
        send = []
        for r in range(world):
            qlo, qhi = shard.query_shard_range(nq, r, world)
            blk = (as_i64[qlo:qhi] if as_i64 is not None else a[qlo:qhi].astype(np.float64))
            send.append(torch.from_numpy(np.ascontiguousarray(blk)))
        recv = [torch.empty((hi - lo,) + tuple(send[0].shape[1:]), dtype=send[0].dtype) for _ in range(world)]
        # gloo has no all_to_all for CPU tensors in every build: pairwise exchange with the same pattern
        for r in range(world):
            if r == rank:
                recv[r].copy_(send[r])
            elif rank < r:
                dist.send(send[r], dst=r); dist.recv(recv[r], src=r)
            else:
                dist.recv(recv[r], src=r); dist.send(send[r], dst=r)
        for r in range(world):
            v = recv[r].numpy()
            got[r][f] = v.view(np.uint64) if f.startswith("key") else v
    best1, best2, w1, w2, second_is_key1 = _merge_tables(got)
    cols = np.arange(hi - lo)
    pick = lambda f, w: np.stack([got[r][f] for r in range(world)])[w, cols]      # noqa: E731
    np.savez(os.path.join(out_dir, f"merged_{rank}.npz"), lo=lo, hi=hi, best1=best1, best2=best2,
             ring_i=pick("ring1", w1), idx_i=pick("idx1", w1), idx_k=pick("idx_k", w1), v0=pick("v0", w1), v2=pick("v2", w1),
             ring_j=np.where(second_is_key1, pick("ring1", w2), pick("ring2", w2)),
             idx_j=np.where(second_is_key1, pick("idx1", w2), pick("idx2", w2)),
             v1=np.where(second_is_key1[:, None], pick("v0", w2), pick("v1", w2)))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_target_sharded_exchange_and_merge_match_full_association(tmp_path):
    """Config 5's data path on two ranks: each searches its block of whole target rings for ALL queries, the per-query top-2
    records are exchanged so that rank r holds everybody's records of its query share, and the merge rule of
    merge_partials_kernel (restated in numpy) must give exactly the oracle's association against the whole target:
    winners, distances, ring neighbour and the three plane points."""
    world = 2
    mp.spawn(_exchange_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    import oracle_lib as ol
    from velo_amd import synth
    d = synth.scan_pair(n_beams=16, n_azimuth=96)
    o = ol.Oracle(icp_skip=1)
    o.set_target(d["tgt_xyz"], d["tgt_off"])
    o.set_source(d["src_xyz"], d["src_off"])
    o.associate(d["x_true"], 1)
    want = o.correspondences()
    off = d["tgt_off"]
    covered = 0
    for r in range(world):
        m = np.load(tmp_path / f"merged_{r}.npz")
        lo, hi = int(m["lo"]), int(m["hi"])
        w = want[lo:hi]
        covered += hi - lo
        has_i, has_j = w["ring_i"] >= 0, w["ring_j"] >= 0
        assert np.array_equal(m["ring_i"][has_i], w["ring_i"][has_i]) and np.array_equal(m["idx_i"][has_i], w["idx_i"][has_i])
        assert np.array_equal(m["ring_j"][has_j], w["ring_j"][has_j]) and np.array_equal(m["idx_j"][has_j], w["idx_j"][has_j])
        assert np.array_equal((m["best1"][has_i] >> np.uint64(32)).astype(np.uint32), w["dist_i"][has_i].view(np.uint32))
        assert np.array_equal((m["best2"][has_j] >> np.uint64(32)).astype(np.uint32), w["dist_j"][has_j].view(np.uint32))
        assert np.all(m["best1"][~has_i] == np.uint64(0xFFFFFFFFFFFFFFFF)) and np.all(m["best2"][~has_j] == np.uint64(0xFFFFFFFFFFFFFFFF))
        both = has_i & has_j
        assert np.array_equal(m["idx_k"][both], w["idx_k"][both])
        gi = off[w["ring_i"][both]] + w["idx_i"][both]
        gk = off[w["ring_i"][both]] + w["idx_k"][both]
        gj = off[w["ring_j"][both]] + w["idx_j"][both]
        assert np.array_equal(m["v0"][both].astype(np.float32), d["tgt_xyz"][gi])
        assert np.array_equal(m["v2"][both].astype(np.float32), d["tgt_xyz"][gk])
        assert np.array_equal(m["v1"][both].astype(np.float32), d["tgt_xyz"][gj])
        valid = w["valid"] == 1
        assert np.array_equal(w["v0"][valid], d["tgt_xyz"][off[w["ring_i"][valid]] + w["idx_i"][valid]])
    assert covered == 16 * 96
