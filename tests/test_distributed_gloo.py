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
