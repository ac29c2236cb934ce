"""-m gpu: the query-sharded mode of SURVEY.md 8(e) executed with MORE THAN ONE RANK on the one GPU of the test box: two
processes, one context each on device 0, the peer-slab all-reduce of the 28-double block inside every LM step
(velo_comm_peer_export / velo_comm_peer_attach: hipIpc-mapped slabs, system-scope stores, rank-order sum).  Every rank must
end with bit-identical poses and LM decisions, equal to the unsharded registration on the same GPU and to the oracle."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_main(rank, world, port, out_dir, shape, with_visual):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    import velo_amd  # noqa: F401
    from velo_amd import api, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)       # carries the 64-byte handles only
    d = synth.scan_pair(n_beams=shape[0], n_azimuth=shape[1])
    ctx = api.Context(0, icp_skip=1)
    handles = [None] * world
    dist.all_gather_object(handles, ctx.comm_peer_export())
    ctx.comm_peer_attach(handles, rank, world)
    assert ctx.comm_info() == (2, rank, world)
    ctx.set_target(d["tgt_xyz"], d["tgt_off"])
    ctx.set_source(d["src_xyz"], d["src_off"])
    if with_visual:
        ctx.set_visual(synth.stereo_matches(60, mix="all"))             # every rank sweeps its share of the visual blocks
    out = {}
    n = ctx.associate(d["x0"], 1)
    cost, Hm, g = ctx.evaluate(d["x0"])                                 # all-reduced sums: the same on every rank
    out.update(n_valid=n, cost=cost, H=Hm, g=g, corr=ctx.correspondences())
    for rep in range(2):                                                # second call: warm seeds, sequence numbers keep counting
        x, T, s = ctx.frame_to_frame(d["x0"])
    out.update(x=x, T=T, counts=np.array([[s.solves[k].termination, s.solves[k].lm_iterations, s.solves[k].evaluations, s.solves[k].n_icp_valid]
                                          for k in range(s.n_solves)]), costs=np.array([s.solves[k].final_cost for k in range(s.n_solves)]))
    np.savez(os.path.join(out_dir, f"rank_{rank}.npz"), **out)
    dist.barrier()
    ctx.close()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world,shape,with_visual", [(2, (32, 400), False), (3, (16, 128), True)])
def test_query_sharded_ranks_on_one_gpu_agree_with_single_rank_and_oracle(hip_lib, oracle, tmp_path, world, shape, with_visual):
    import torch.multiprocessing as mp
    import helpers as H
    from velo_amd import api, synth
    mp.spawn(_rank_main, args=(world, _free_port(), str(tmp_path), shape, with_visual), nprocs=world, join=True)
    ranks = [np.load(tmp_path / f"rank_{r}.npz") for r in range(world)]
    d = synth.scan_pair(n_beams=shape[0], n_azimuth=shape[1])
    vis = synth.stereo_matches(60, mix="all") if with_visual else None
    # every rank: the same sums, the same pose, the same LM decisions -- bit for bit
    for r in ranks[1:]:
        assert r["cost"] == ranks[0]["cost"] and np.array_equal(r["H"], ranks[0]["H"]) and np.array_equal(r["g"], ranks[0]["g"])
        assert np.array_equal(r["x"], ranks[0]["x"]) and np.array_equal(r["T"], ranks[0]["T"])
        assert np.array_equal(r["counts"][:, :3], ranks[0]["counts"][:, :3]) and np.array_equal(r["costs"], ranks[0]["costs"])
    # the shards tile the query list; the all-reduced sums are the oracle's
    orc = oracle.Oracle(threads=4, icp_skip=1)
    H.load_both(api.Context(0, icp_skip=1), orc, d, visual=vis)
    n_cpu = orc.associate(d["x0"], 1)
    assert sum(int(r["n_valid"]) for r in ranks) == n_cpu
    H.assert_corr_equal(np.concatenate([r["corr"] for r in ranks]), orc.correspondences())
    if vis is not None:
        orc.build_visual(d["x0"], 1)
    c2, H2, g2 = orc.evaluate(d["x0"])
    if vis is None:                                                      # (with visual blocks velo_evaluate needs build_visual first on rank 0)
        assert abs(float(ranks[0]["cost"]) - c2) <= 1e-12 * c2 and H.rel_err(ranks[0]["H"], H2) <= 1e-12 and H.rel_err(ranks[0]["g"], g2) <= 1e-12
    xo, To, so = orc.frame_to_frame(d["x0"])
    assert H.pose_close(ranks[0]["x"], xo, 1e-9, 1e-10)
    want = np.array([[so.solves[k].termination, so.solves[k].lm_iterations, so.solves[k].evaluations] for k in range(so.n_solves)])
    assert np.array_equal(ranks[0]["counts"][:, :3], want)
    assert sum(int(r["counts"][-1, 3]) for r in ranks) == so.solves[so.n_solves - 1].n_icp_valid
    # and the unsharded registration on the same GPU
    one = api.Context(0, icp_skip=1)
    one.set_target(d["tgt_xyz"], d["tgt_off"]); one.set_source(d["src_xyz"], d["src_off"])
    if vis is not None:
        one.set_visual(vis)
    x1, T1, s1 = one.frame_to_frame(d["x0"])
    one.close()
    assert H.pose_close(ranks[0]["x"], x1, 1e-11, 1e-12)


def test_peer_communicator_of_one_rank_changes_nothing(hip_lib):
    from velo_amd import api, synth
    d = synth.scan_pair(n_beams=16, n_azimuth=128)
    a, b = api.Context(0, icp_skip=1), api.Context(0, icp_skip=1)
    for c in (a, b):
        c.set_target(d["tgt_xyz"], d["tgt_off"]); c.set_source(d["src_xyz"], d["src_off"])
    b.comm_peer_attach([b.comm_peer_export()], 0, 1)
    xa, Ta, sa = a.frame_to_frame(d["x0"])
    xb, Tb, sb = b.frame_to_frame(d["x0"])
    assert np.array_equal(xa, xb) and sa.n_solves == sb.n_solves
    b.comm_destroy()
    assert b.comm_info()[0] == 0
    a.close(); b.close()


# ---- BASELINE config 5 with more than one rank: every rank holds a block of whole target rings -------------------------------------
def _target_sharded_main(rank, world, port, out_dir, n_scans):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    import velo_amd  # noqa: F401
    from velo_amd import api, shard, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = synth.scan_to_map(n_scans * 32 * 300, n_beams=32, n_azimuth=300)           # a small accumulated map: n_scans x 32 rings
    nq = 32 * 300
    ctx = api.Context(0, icp_skip=1)
    handles = [None] * world
    dist.all_gather_object(handles, ctx.comm_peer_export())
    ctx.comm_peer_attach(handles, rank, world)
    dist.all_gather_object(handles, ctx.comm_peer_export_records(nq))
    ctx.comm_peer_attach_records(handles, nq)
    ctx.comm_set_target_sharded(True)
    r0, r1, p0, local = shard.target_ring_block(d["tgt_off"], rank, world)
    ctx.set_target_part(d["tgt_xyz"][p0:p0 + int(local[-1])], local, r0, p0)
    ctx.set_source(d["src_xyz"], d["src_off"])
    n = ctx.associate(d["x0"], 1)                           # partial search, record exchange, merge of my query share
    corr = ctx.correspondences()
    for rep in range(2):
        x, T, s = ctx.frame_to_frame(d["x0"])
    np.savez(os.path.join(out_dir, f"ts_{rank}.npz"), n_valid=n, corr=corr, x=x,
             counts=np.array([[s.solves[k].termination, s.solves[k].lm_iterations, s.solves[k].evaluations, s.solves[k].n_icp_valid] for k in range(s.n_solves)]))
    dist.barrier()
    ctx.close()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 3])
def test_target_sharded_ranks_on_one_gpu_match_the_oracle(hip_lib, oracle, tmp_path, world):
    """Scan-to-map with the map's rings dealt over the ranks (processes on the one GPU): per-query top-2 records exchanged by
    direct stores into the owners' peer-mapped areas every association round, merged by the owner, then the query-sharded
    solve with the peer all-reduce.  Merged tables = the oracle's association against the WHOLE map; pose = the oracle's."""
    import torch.multiprocessing as mp
    import helpers as H
    from velo_amd import synth
    n_scans = 5
    mp.spawn(_target_sharded_main, args=(world, _free_port(), str(tmp_path), n_scans), nprocs=world, join=True)
    ranks = [np.load(tmp_path / f"ts_{r}.npz") for r in range(world)]
    d = synth.scan_to_map(n_scans * 32 * 300, n_beams=32, n_azimuth=300)
    orc = oracle.Oracle(threads=8, icp_skip=1)
    orc.set_target(d["tgt_xyz"], d["tgt_off"])
    orc.set_source(d["src_xyz"], d["src_off"])
    n_cpu = orc.associate(d["x0"], 1)
    assert sum(int(r["n_valid"]) for r in ranks) == n_cpu
    H.assert_corr_equal(np.concatenate([r["corr"] for r in ranks]), orc.correspondences())
    xo, To, so = orc.frame_to_frame(d["x0"])
    for r in ranks:
        assert np.array_equal(r["x"], ranks[0]["x"])                              # every rank took the same LM decisions
        assert np.array_equal(r["counts"][:, :3], ranks[0]["counts"][:, :3])
    assert H.pose_close(ranks[0]["x"], xo, 1e-9, 1e-10)
    want = np.array([[so.solves[k].termination, so.solves[k].lm_iterations, so.solves[k].evaluations] for k in range(so.n_solves)])
    assert np.array_equal(ranks[0]["counts"][:, :3], want)
    assert sum(int(r["counts"][-1, 3]) for r in ranks) == so.solves[so.n_solves - 1].n_icp_valid


# ---- chained calls over peers: the ranks agree on the launch counts, whatever their own predictions are -------------------------------
def _uneven_rank_main(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    # rank 1 predicts with a different (fixed) margin: without the agreement it would enqueue more LM launches per solve than rank 0,
    # its last all-reduces would wait for a partner that never comes, and the call would fail with VELO_ERR_COMM after 5 s
    if rank == 1:
        os.environ["VELO_CHAIN_MARGIN"] = "3"
    import torch.distributed as dist
    import velo_amd  # noqa: F401
    from velo_amd import api, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = synth.scan_pair(n_beams=32, n_azimuth=400)
    ctx = api.Context(0, icp_skip=1)
    out = {}
    for epoch in range(2):                                   # the second pass exports and attaches AGAIN: new slabs, same results
        handles = [None] * world
        dist.all_gather_object(handles, ctx.comm_peer_export())
        ctx.comm_peer_attach(handles, rank, world)
        ctx.set_target(d["tgt_xyz"], d["tgt_off"])
        ctx.set_source(d["src_xyz"], d["src_off"])
        for rep in range(3):
            x, T, s = ctx.frame_to_frame(d["x0"])
        out[f"x{epoch}"] = x
        out[f"counts{epoch}"] = np.array([[s.solves[k].termination, s.solves[k].lm_iterations, s.solves[k].evaluations] for k in range(s.n_solves)])
    out["chain"] = np.array(ctx.chain_stats())
    np.savez(os.path.join(out_dir, f"uneven_{rank}.npz"), **out)
    dist.barrier()
    ctx.close()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_chained_peer_calls_agree_on_launch_counts_when_the_ranks_predict_differently(hip_lib, tmp_path):
    import torch.multiprocessing as mp
    from velo_amd import api, synth
    mp.spawn(_uneven_rank_main, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    a, b = (np.load(tmp_path / f"uneven_{r}.npz") for r in range(2))
    for e in range(2):
        assert np.array_equal(a[f"x{e}"], b[f"x{e}"]) and np.array_equal(a[f"counts{e}"], b[f"counts{e}"])
    assert np.array_equal(a["x0"], a["x1"])
    assert a["chain"][0] >= 6 and b["chain"][0] >= 6                       # the calls really were chained ...
    assert a["chain"][1] == b["chain"][1]                                   # ... and a miss, if any, is every rank's miss
    d = synth.scan_pair(n_beams=32, n_azimuth=400)
    one = api.Context(0, icp_skip=1)
    one.set_target(d["tgt_xyz"], d["tgt_off"]); one.set_source(d["src_xyz"], d["src_off"])
    x1, _T, s1 = one.frame_to_frame(d["x0"])
    one.close()
    import helpers as H
    assert H.pose_close(a["x0"], x1, 1e-11, 1e-12)
    assert np.array_equal(a["counts0"], np.array([[s1.solves[k].termination, s1.solves[k].lm_iterations, s1.solves[k].evaluations] for k in range(s1.n_solves)]))
