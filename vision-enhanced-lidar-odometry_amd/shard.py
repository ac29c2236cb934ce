"""Host-side logic of the multi-GPU modes; no arithmetic on the data path happens here.

Query-sharded mode (north_star): every rank keeps the whole target and the contiguous share
[nq*rank/world, nq*(rank+1)/world) of the query list; each LM evaluation all-reduces the 28-double block
(21 upper-triangular JtJ + 6 Jtr + cost).  The C library does this with RCCL (velo_comm_init); this module holds the
partition rule and the (un)packing of that block so that CPU tests can exercise the same layout over gloo.
"""
from __future__ import annotations

import numpy as np

N_ACC = 28


def query_shard_range(n_queries: int, rank: int, world: int):
    """Same rule as q_range() in csrc/velo_hip.hip and query_list() in the oracle."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    return (n_queries * rank) // world, (n_queries * (rank + 1)) // world


def target_ring_block(ring_offsets, rank: int, world: int):
    """Target-sharded mode: rank owns the contiguous block of WHOLE rings [Rs*rank/world, Rs*(rank+1)/world) (keeps the
    +-1 ring neighbour of velo.h:852-856 local).  Returns (first_ring, last_ring_excl, first_point, local_offsets)."""
    off = np.asarray(ring_offsets, dtype=np.int64)
    rs = len(off) - 1
    r0, r1 = (rs * rank) // world, (rs * (rank + 1)) // world
    local = (off[r0:r1 + 1] - off[r0]).astype(np.int32)
    return r0, r1, int(off[r0]), local


def pack_normal_equations(cost: float, JtJ: np.ndarray, Jtr: np.ndarray) -> np.ndarray:
    """cost, 6x6, 6 -> the 28-double wire block (row-major upper triangle, then Jtr, then cost)."""
    JtJ = np.asarray(JtJ, dtype=np.float64).reshape(6, 6)
    out = np.empty(N_ACC)
    k = 0
    for i in range(6):
        for j in range(i, 6):
            out[k] = JtJ[i, j]
            k += 1
    out[21:27] = np.asarray(Jtr, dtype=np.float64)
    out[27] = cost
    return out


def unpack_normal_equations(block: np.ndarray):
    block = np.asarray(block, dtype=np.float64)
    H = np.zeros((6, 6))
    k = 0
    for i in range(6):
        for j in range(i, 6):
            H[i, j] = H[j, i] = block[k]
            k += 1
    return float(block[27]), H, block[21:27].copy()


def init_comm_from_torch(ctx, dist, rank: int, world: int):
    """Bootstraps the library's RCCL communicator: rank 0 makes the id, torch.distributed carries it."""
    from . import api
    uid = [api.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(uid, src=0)
    ctx.comm_init(uid[0], rank, world)
