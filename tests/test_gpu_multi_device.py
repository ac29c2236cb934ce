"""-m gpu: the multi-GPU modes of SURVEY.md 8(e) with ONE RANK PER DEVICE -- the first cross-device execution of the peer slabs
(hipIpc handles opened on another device, system-scope stores over xGMI) and of the RCCL path with more than one rank.

The test box of a round has one GPU, where these tests SKIP (loudly: the reason says what was not exercised); on a box with
>= 2 devices they run by themselves, so pytest -- not bench.py -- is the first thing that crosses devices.  The rank function is
shared with two cases that DO run on one GPU (a one-rank RCCL communicator; two peer-slab ranks on device 0), so the harness itself
is exercised every round.

  query-sharded  (north_star; BASELINE configs[1] at full size): queries split 1/W, the 28-double block all-reduced every LM
                 evaluation (velo.h:806-807 iterations are independent; velo.h:897-902 one solve for all)
  target-sharded (BASELINE configs[4]): whole target rings dealt over the ranks, per-query top-2 records exchanged every round
each over peer slabs AND over RCCL; every rank's pose must equal the single-rank registration (same LM decisions: <= 1e-9).
"""
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


def _workload(mode, small):
    from velo_amd import synth
    if mode == "query":
        return synth.scan_pair(n_beams=32, n_azimuth=400) if small else synth.scan_pair()
    return synth.scan_to_map(5 * 32 * 300, n_beams=32, n_azimuth=300) if small else synth.scan_to_map(2_000_000)


def _rank_main(rank, world, port, out_dir, mode, comm, devices, small):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    import velo_amd  # noqa: F401
    from velo_amd import api, shard
    dev = devices[rank]
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)       # carries handles / the RCCL id only
    d = _workload(mode, small)
    nq = int(d["src_xyz"].shape[0])
    ctx = api.Context(dev, icp_skip=1)
    if comm == "peer":
        handles = [None] * world
        dist.all_gather_object(handles, ctx.comm_peer_export())
        ctx.comm_peer_attach(handles, rank, world)                      # (no barrier needed behind it: the slab was cleared at export)
        if mode == "target":
            dist.all_gather_object(handles, ctx.comm_peer_export_records(nq))
            ctx.comm_peer_attach_records(handles, nq)
    else:
        uid = [api.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        ctx.comm_init(uid[0], rank, world)
    kind, r, w = ctx.comm_info()
    assert (kind, r, w) == (2 if comm == "peer" else 1, rank, world)
    if mode == "target":
        ctx.comm_set_target_sharded(True)
        r0, r1, p0, local = shard.target_ring_block(d["tgt_off"], rank, world)
        ctx.set_target_part(d["tgt_xyz"][p0:p0 + int(local[-1])], local, r0, p0)
    else:
        ctx.set_target(d["tgt_xyz"], d["tgt_off"])
    ctx.set_source(d["src_xyz"], d["src_off"])
    for rep in range(2):                                                # second call: warm seeds, chain predictions, sequence numbers keep counting
        x, T, s = ctx.frame_to_frame(d["x0"])
    counts = np.array([[s.solves[k].termination, s.solves[k].lm_iterations, s.solves[k].evaluations, s.solves[k].n_icp_valid] for k in range(s.n_solves)])
    np.savez(os.path.join(out_dir, f"md_{rank}.npz"), x=x, T=T, counts=counts, device=dev)
    dist.barrier()
    if comm != "peer":
        ctx.comm_destroy()
    ctx.close()
    dist.destroy_process_group()


def _run_and_check(tmp_path, mode, comm, devices, small):
    import torch.multiprocessing as mp
    from velo_amd import api
    world = len(devices)
    mp.spawn(_rank_main, args=(world, _free_port(), str(tmp_path), mode, comm, devices, small), nprocs=world, join=True)
    ranks = [np.load(tmp_path / f"md_{r}.npz") for r in range(world)]
    d = _workload(mode, small)
    one = api.Context(0, icp_skip=1)
    one.set_target(d["tgt_xyz"], d["tgt_off"]); one.set_source(d["src_xyz"], d["src_off"])
    for rep in range(2):
        x1, T1, s1 = one.frame_to_frame(d["x0"])
    one.close()
    want = np.array([[s1.solves[k].termination, s1.solves[k].lm_iterations, s1.solves[k].evaluations] for k in range(s1.n_solves)])
    for r in ranks:
        assert np.array_equal(r["x"], ranks[0]["x"]) and np.array_equal(r["T"], ranks[0]["T"])     # every rank: the same bits
        assert np.array_equal(r["counts"][:, :3], want), (r["counts"], want)                        # the single-rank call's LM decisions
    assert np.abs(ranks[0]["x"] - x1).max() <= 1e-9, (ranks[0]["x"], x1)
    assert sum(int(r["counts"][-1, 3]) for r in ranks) == s1.solves[s1.n_solves - 1].n_icp_valid   # the query shares tile the list
    return ranks


# ---- the harness on ONE device (runs every round) -----------------------------------------------------------------------------------
@pytest.mark.timeout(600)
@pytest.mark.parametrize("mode", ["query", "target"])
def test_harness_two_peer_ranks_on_one_device(hip_lib, tmp_path, mode):
    _run_and_check(tmp_path, mode, "peer", [0, 0], small=True)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("mode", ["query", "target"])
def test_harness_one_rank_rccl_communicator(hip_lib, tmp_path, mode):
    _run_and_check(tmp_path, mode, "rccl", [0], small=True)


# ---- one rank per device (self-activating on a multi-GPU box) ------------------------------------------------------------------------
def _devices():
    import torch
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip(f"NOT EXERCISED: {n} GPU visible -- the cross-device peer slabs (hipIpc + system-scope atomics over xGMI) and RCCL with "
                    f"more than one rank need >= 2 devices; this test runs by itself on such a box")
    return list(range(min(n, 8)))


@pytest.mark.timeout(1800)
@pytest.mark.parametrize("comm", ["peer", "rccl"])
def test_query_sharded_full_size_one_rank_per_device(hip_lib, tmp_path, comm):
    """BASELINE configs[1] (120k x 120k) split by queries over all devices of the node, and over two of them."""
    devs = _devices()
    _run_and_check(tmp_path, "query", comm, devs[:2], small=False)
    if len(devs) > 2:
        _run_and_check(tmp_path, "query", comm, devs, small=False)


@pytest.mark.timeout(1800)
@pytest.mark.parametrize("comm", ["peer", "rccl"])
def test_target_sharded_2m_map_one_rank_per_device(hip_lib, tmp_path, comm):
    """BASELINE configs[4]: the 2M-point map's rings dealt over the devices, records exchanged every association round."""
    devs = _devices()
    _run_and_check(tmp_path, "target", comm, devs, small=False)
