#!/usr/bin/env python3
"""bench.py -- scan-pairs/s of the MI355X scan-matching core on BASELINE.json's workload.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one batch of B complete scan-pair registrations: from the two raw ring clouds already resident
in HBM to the solved pose -- target index build (velo_set_target), query list (velo_set_source) and
velo_frame_to_frame (6 association rounds + 6 Levenberg-Marquardt solves to Ceres-default tolerances).
The headline (`value`) is workload configs[1] of BASELINE.json: synthetic HDL-64E pair, 64 x 1875 = 120,000 points each,
icp_skip = 1; at N > 1 every rank registers its own pairs (replicas, no data-path collective -> "scaling": "weak").

The same JSON line also carries
  configs   short legs of the other single-GPU workloads (c1: reference constants, c3: + 2,000 stereo blocks, c4: 2M-point map),
            each with pairs/s, launches, algorithmic bytes and its own roofline object                       (N = 1)
  modes     N > 1 only: the north_star's multi-GPU modes next to the replicas --
            sharded          ONE pair per step, queries split 1/N per rank, the 28-double normal-equation block all-reduced
                             every LM evaluation (peer-mapped slabs inside the LM step; --comm rccl: ncclAllReduce)  -> strong scaling
            target_sharded   BASELINE config 5: ONE scan-to-map pair per step, each rank holds a block of whole target rings,
                             per-query top-2 records exchanged every association round (direct stores into the owners'
                             peer-mapped areas; --comm rccl: grouped ncclSend / ncclRecv), then as "sharded"
            each with the rank count READ BACK from the communicator.
`--mode sharded|target-sharded` makes one of those the timed `value` instead.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
METRIC = "scan-pairs/sec + achieved HBM GB/s, 120k-pt HDL-64E frame-to-frame ICP"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--mode", choices=["replicas", "sharded", "target-sharded"], default="replicas")
    ap.add_argument("--workload", choices=["c1", "c2", "c3", "c4"], default="c2",
                    help="c2: 120k pair; c3: + 2000 stereo blocks; c4: 120k scan vs 2M-point map; c1: reference constants (icp_skip=200)")
    ap.add_argument("--batch", type=int, default=8, help="independent pairs in flight per GPU (one context + stream each)")
    ap.add_argument("--threads-per-pair", dest="batch_api", action="store_false",
                    help="drive every pair from its own host thread (frame_to_frame) instead of velo_register_batch")
    ap.add_argument("--separate-loads", action="store_true",
                    help="A/B: velo_set_target/source from B host threads, then velo_frame_to_frame_batch (instead of velo_register_batch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="only the timed workload: no c1/c3/c4 legs, no multi-GPU mode legs")
    ap.add_argument("--comm", choices=["peer", "rccl"], default="peer", help="all-reduce of the sharded mode: peer-mapped slabs or RCCL")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend for the barrier / max-over-ranks (nccl = RCCL)")
    ap.add_argument("--force-device", type=int, default=None, help="testing only: every rank uses this device (with --dist-backend gloo)")
    ap.add_argument("--cpu-sample-skip", type=int, default=4)
    return ap.parse_args()


_cache = {}


def make_workload(name):
    """-> (scan pair, visual matches or None, label, icp_skip)"""
    from velo_amd import synth
    if name in _cache:
        return _cache[name]
    if name == "c1":
        # configs[0], the reference's own constants (kitti.h:8: icp_skip = 200 -> 640 queries per round).  KITTI seq 00 is not in
        # this image: the synthetic pair stands in, in the same ring layout the KITTI reader produces.
        out = (synth.scan_pair(), None, "configs[0] stand-in: synthetic 120k-pt pair, reference constants (icp_skip=200)", 200)
    elif name == "c4":
        out = (synth.scan_to_map(2_000_000), None, "synthetic HDL-64E 120k-pt scan vs 2M-pt accumulated map (configs[3]), icp_skip=1", 1)
    elif name == "c3":
        out = (synth.scan_pair(), synth.stereo_matches(1000), "configs[2]: 120k-pt pair + 2000 stereo reprojection blocks, icp_skip=1", 1)
    else:
        out = (synth.scan_pair(), None, "synthetic HDL-64E 64x1875=120k-pt scan pair (configs[1]), icp_skip=1, point-to-plane ICP", 1)
    _cache[name] = out
    return out


def cpu_baseline(d, vis, sample_skip, icp_skip=1):
    """The CPU restatement (oracle = 'port') timed on this host: (i) all cores on the full pair,
    (ii) one thread -- the reference's configuration (velo.h:900) -- on a 1/sample_skip query sample (the whole pair when the
    queries are sparse already, icp_skip > 1: nothing extrapolated there)."""
    import oracle_lib
    cores = oracle_lib.max_threads()
    o = oracle_lib.Oracle(threads=cores, icp_skip=icp_skip)
    if icp_skip > 1:
        sample_skip = 1
    t0 = time.perf_counter()
    o.set_target(d["tgt_xyz"], d["tgt_off"])
    o.set_source(d["src_xyz"], d["src_off"])
    if vis is not None:
        o.set_visual(vis)
    x, _, s = o.frame_to_frame(d["x0"])
    t_all = time.perf_counter() - t0
    o1 = oracle_lib.Oracle(threads=1, icp_skip=icp_skip * sample_skip)
    t0 = time.perf_counter()
    o1.set_target(d["tgt_xyz"], d["tgt_off"])
    o1.set_source(d["src_xyz"], d["src_off"])
    if vis is not None:
        o1.set_visual(vis)
    o1.frame_to_frame(d["x0"])
    t_meas = time.perf_counter() - t0
    t_one = t_meas * sample_skip
    single = (f"single thread (the reference's configuration) on the whole pair = {t_one:.2f} s" if sample_skip == 1 else
              f"single thread (the reference's configuration) measured on a 1/{sample_skip} query sample = {t_meas:.1f} s, x {sample_skip} = {t_one:.1f} s/pair")
    return {
        "value": 1.0 / t_all, "unit": "scan-pairs/s", "cores": cores, "kind": "port",
        "sample": f"1 full pair (icp_skip={icp_skip}) on {cores} OpenMP threads = {t_all:.2f} s; {single}",
        "single_thread_pairs_per_s": 1.0 / t_one, "single_thread_extrapolated": sample_skip != 1,
        "x": [float(v) for v in x],
    }


class Rig:
    """torch.distributed plumbing of one rank (torch is only the allocator and the process group here)."""

    def __init__(self, a):
        import torch
        self.torch = torch
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist = None
        self.backend = a.dist_backend
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if a.force_device is not None:
                self.local_rank = a.force_device
            torch.cuda.set_device(self.local_rank)
            if a.dist_backend == "nccl":
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", self.local_rank))
            else:
                dist.init_process_group(backend=a.dist_backend)
            self.dist = dist
        self.dev = torch.device("cuda", self.local_rank)
        torch.cuda.set_device(self.dev)

    def barrier(self, ctxs=()):
        if self.dist is not None:
            self.dist.barrier()
        self.torch.cuda.synchronize()
        for c in ctxs:
            c.synchronize()

    def max_over_ranks(self, dt):
        if self.dist is None:
            return dt
        t = self.torch.tensor([dt], device=self.dev if self.backend == "nccl" else "cpu", dtype=self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())


def run_leg(rig, a, workload, mode, B, steps, warmup, single_leg=True, comm="peer"):
    """One timed leg: `steps` steps of B registrations (B = 1 in the sharded modes) bracketed by barrier + synchronize, max over ranks."""
    import velo_amd  # noqa: F401
    from velo_amd import api
    torch = rig.torch
    d, vis, label, icp_skip = make_workload(workload)
    world, rank = rig.world, rig.rank
    B = 1 if mode != "replicas" else max(1, B)
    tgt_off, tgt_first_ring, tgt_first_point = d["tgt_off"], 0, 0
    tgt_np = d["tgt_xyz"]
    if mode == "target-sharded" and world > 1:
        from velo_amd import shard
        r0, r1, p0, tgt_off = shard.target_ring_block(d["tgt_off"], rank, world)
        tgt_first_ring, tgt_first_point = r0, p0
        tgt_np = d["tgt_xyz"][p0:p0 + int(tgt_off[-1])]
    # inputs resident in HBM before the timed region
    tgt = torch.from_numpy(np.ascontiguousarray(tgt_np)).to(rig.dev)
    src = torch.from_numpy(d["src_xyz"]).to(rig.dev)
    torch.cuda.synchronize()
    ctxs = [api.Context(rig.local_rank, icp_skip=icp_skip) for _ in range(B)]
    comm_info = None
    try:
        for c in ctxs:
            c.set_timing(True)
            if vis is not None:
                c.set_visual(vis)
        if mode != "replicas" and world > 1:
            ok = comm == "peer"
            if ok:
                # peer-mapped slabs.  Every rank must end up on the same path: each local step is followed by an all-gather of its
                # outcome, so a rank that cannot export or map a handle makes everybody fall back to RCCL together.
                def agreed(fn):
                    try:
                        val = fn()
                    except Exception as e:       # noqa: BLE001
                        print(f"[bench] rank {rank}: peer slabs unavailable ({e})", file=sys.stderr, flush=True)
                        val = None
                    got = [None] * world
                    rig.dist.all_gather_object(got, val)
                    return got if all(g is not None for g in got) else None
                c0 = ctxs[0]
                handles = agreed(c0.comm_peer_export)
                ok = handles is not None and agreed(lambda: (c0.comm_peer_attach(handles, rank, world), 1)[1]) is not None
                if ok and mode == "target-sharded":
                    nq_max = int(d["src_xyz"].shape[0])
                    rh = agreed(lambda: c0.comm_peer_export_records(nq_max))
                    ok = rh is not None and agreed(lambda: (c0.comm_peer_attach_records(rh, nq_max), 1)[1]) is not None
                    if ok:
                        c0.comm_set_target_sharded(True)
                if not ok:
                    c0.comm_destroy()
            if not ok:
                uid = [api.comm_unique_id() if rank == 0 else None]
                rig.dist.broadcast_object_list(uid, src=0)
                ctxs[0].comm_init(uid[0], rank, world)
                if mode == "target-sharded":
                    ctxs[0].comm_set_target_sharded(True)
            kind, _, n_ranks = ctxs[0].comm_info()
            comm_info = {"kind": {1: "rccl", 2: "peer slabs (hipIpc)"}.get(kind, "none"), "ranks": n_ranks}
        results = [None] * B

        def load_pair(i):
            ctxs[i].set_target_part(tgt, tgt_off, tgt_first_ring, tgt_first_point)
            ctxs[i].set_source(src, d["src_off"])

        def one_pair(i):
            load_pair(i)
            results[i] = ctxs[i].frame_to_frame(d["x0"])

        pool = ThreadPoolExecutor(max_workers=B) if B > 1 else None
        # (B == 1, whole target: the same single library call with one job -- velo_register_batch routes it to the single-pair path)
        one_call = B == 1 and a.batch_api and tgt_first_ring == 0 and tgt_first_point == 0 and mode == "replicas"
        batch_refs = (api.scan_refs([(tgt, tgt_off)] * B, rig.local_rank), api.scan_refs([(src, d["src_off"])] * B, rig.local_rank)) if (B > 1 or one_call) else None
        x0s = np.tile(np.asarray(d["x0"], dtype=np.float64), (B, 1))

        def step():
            if pool is None and one_call:
                xs, Ts, Ss = api.register_batch(ctxs, None, None, x0s, refs=batch_refs)
                results[0] = (xs[0], Ts[0], Ss[0])
            elif pool is None:
                one_pair(0)
            elif a.batch_api and a.separate_loads:
                list(pool.map(load_pair, range(B)))
                xs, Ts, Ss = api.frame_to_frame_batch(ctxs, x0s)
                for i in range(B):
                    results[i] = (xs[i], Ts[i], Ss[i])
            elif a.batch_api:
                # the B pairs' scans (device pointers) and the B registrations in ONE library call (velo_register_batch)
                xs, Ts, Ss = api.register_batch(ctxs, None, None, x0s, refs=batch_refs)
                for i in range(B):
                    results[i] = (xs[i], Ts[i], Ss[i])
            else:
                list(pool.map(one_pair, range(B)))

        for _ in range(warmup):
            step()
        rig.barrier(ctxs)
        t0 = time.perf_counter()
        assoc_ms, assoc_n, alg_bytes, assoc_bytes, evals = 0.0, 0, 0, 0, 0
        for _ in range(steps):
            step()
            for r in results:
                s = r[2]
                assoc_ms += s.assoc_kernel_ms
                assoc_n += s.assoc_kernel_launches
                alg_bytes += s.algorithmic_bytes
                assoc_bytes += s.assoc_bytes
                evals += sum(s.solves[k].evaluations for k in range(s.n_solves))
        rig.barrier(ctxs)
        dt = rig.max_over_ranks(time.perf_counter() - t0)

        single = None
        if single_leg and world == 1 and mode == "replicas" and B > 1:
            # SURVEY 8(d) asks for the single-pair latency next to the throughput: one pair in flight, after the timed region
            for _ in range(3):
                one_pair(0)
            ctxs[0].synchronize()
            t1 = time.perf_counter()
            n1, a_ms, a_n = 20, 0.0, 0
            for _ in range(n1):
                one_pair(0)
                a_ms += results[0][2].assoc_kernel_ms
                a_n += results[0][2].assoc_kernel_launches
            ctxs[0].synchronize()
            lat = (time.perf_counter() - t1) / n1
            single = {"pairs_in_flight": 1, "ms_per_pair": 1e3 * lat, "pairs_per_s": 1.0 / lat, "assoc_avg_launch_us": 1e3 * a_ms / max(a_n, 1)}

        shared = None
        if workload == "c4" and mode == "replicas" and B > 1 and a.batch_api:
            # scan-to-map as it is used: B scans against ONE map -- the jobs name the same target with VELO_SCAN_SHARED, the library
            # indexes it once per step and the B contexts hold it by reference (one 110 MB map in HBM instead of B)
            refs_sh = (api.scan_refs([(tgt, tgt_off)] * B, rig.local_rank, shared=True), batch_refs[1])
            for _ in range(2):
                api.register_batch(ctxs, None, None, x0s, refs=refs_sh)
            rig.barrier(ctxs)
            t1 = time.perf_counter()
            n_sh = max(4, steps // 2)
            for _ in range(n_sh):
                xs_sh, _, _ = api.register_batch(ctxs, None, None, x0s, refs=refs_sh)
            rig.barrier(ctxs)
            dt_sh = rig.max_over_ranks(time.perf_counter() - t1)
            shared = {"pairs_per_s": n_sh * B * world / dt_sh, "ms_per_step": 1e3 * dt_sh / n_sh, "steps": n_sh,
                      "pose_equal_to_unshared": bool(np.array_equal(xs_sh[0], results[0][0])),
                      "note": "the B jobs share one target (VELO_SCAN_SHARED): one upload + index build per step instead of B"}

        n_pairs_rank = steps * B
        total_pairs = n_pairs_rank * (world if mode == "replicas" else 1)
        s0 = results[0][2]
        b_launch = assoc_bytes / max(assoc_n, 1)        # the batch driver serves the same round of several contexts with one launch
        avg_ms = assoc_ms / max(assoc_n, 1)
        achieved = (b_launch / 1e9) / (avg_ms / 1e3) if avg_ms > 0 else 0.0
        per_pair_bytes = alg_bytes / max(n_pairs_rank, 1)
        batched = B > 1 and a.batch_api
        leg = {
            "workload": label, "mode": mode, "pairs_in_flight_per_gpu": B, "steps": steps, "warmup": warmup,
            "pairs_per_s": total_pairs / dt, "ms_per_step": 1e3 * dt / steps,
            "Nq": int(s0.n_queries), "Nt": int(s0.n_target),
            "lm_evaluations_per_pair": evals / max(n_pairs_rank, 1),
            "association_launches_per_pair": assoc_n / max(n_pairs_rank, 1),
            "valid_correspondences_last_round": int(s0.solves[s0.n_solves - 1].n_icp_valid),
            "algorithmic_bytes_per_pair": per_pair_bytes,
            "achieved_hbm_GBs_whole_path": per_pair_bytes * (total_pairs / dt) / 1e9,
            # sparse rounds (icp_skip >= 4 and <= 12,288 queries: the c1 leg) are searched one wave per query, the others by the tube kernel
            "roofline": {"bound": "hbm", "kernel": (("assoc_direct_batch_kernel" if batched else "assoc_direct_kernel") if (icp_skip >= 4 and int(s0.n_queries) <= 12288)
                                                    else ("assoc_search_v5_batch_kernel" if batched else "assoc_search_v5_kernel")),
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "avg_launch_us": avg_ms * 1e3, "algorithmic_bytes_per_launch": b_launch},
            "solution_x": [float(v) for v in results[0][0]],
        }
        if single is not None:
            leg["single_pair"] = single
        if shared is not None:
            leg["shared_target"] = shared
        if comm_info is not None:
            leg["communicator"] = comm_info
        return leg
    finally:
        if rig.dist is not None:
            rig.dist.barrier()
        for c in ctxs:
            c.close()
        del tgt, src
        torch.cuda.empty_cache()


def main():
    a = parse()
    rig = Rig(a)
    world, rank = rig.world, rig.rank
    main_leg = run_leg(rig, a, a.workload, a.mode, a.batch, a.steps, a.warmup, comm=a.comm)
    legs, modes = {}, {}
    if not a.no_legs:
        if world == 1 and a.mode == "replicas":
            # short legs of the other single-GPU configs (each a complete bench of its own: warm-up, barrier-bracketed timed region)
            for name, steps in (("c1", 100), ("c3", 40), ("c4", 12)):
                if name != a.workload:
                    legs[name] = run_leg(rig, a, name, "replicas", a.batch, steps, 3)
        if world > 1 and a.mode == "replicas" and os.environ.get("VELO_BENCH_MODES", "1") != "0":
            # the north_star's multi-GPU modes, next to the replicas: one pair per step, strong scaling.  Never fatal for the headline:
            # a leg that fails on any rank is reported as an error by all of them (the ranks agree after every attempt), and a failed
            # peer-slab attempt is repeated over RCCL.
            def mode_leg(workload, mode, steps, warmup):
                out = None
                for comm in ([a.comm, "rccl"] if a.comm == "peer" else [a.comm]):
                    try:
                        res, ok = run_leg(rig, a, workload, mode, 1, steps, warmup, single_leg=False, comm=comm), True
                    except Exception as e:       # noqa: BLE001
                        res, ok = {"error": f"{comm}: {str(e)[:300]}"}, False
                    got = [None] * world
                    rig.dist.all_gather_object(got, ok)
                    if all(got):
                        if out is not None:
                            res["first_attempt"] = out
                        return res
                    out = res if not ok else {"error": f"{comm}: another rank failed"}
                return out
            modes["sharded"] = mode_leg(a.workload, "sharded", max(20, a.steps // 4), 5)
            modes["target_sharded"] = mode_leg("c4", "target-sharded", 10, 3)

    if rank == 0:
        rf = dict(main_leg["roofline"])
        # the committed PMC passes: profiles/rNN_traffic.json (config 2), profiles/rNN_c4_traffic.json (config 4)
        pat = {"c2": "r[0-9][0-9]_traffic.json", "c4": "r[0-9][0-9]_c4_traffic.json"}.get(a.workload)
        tf = sorted(glob.glob(os.path.join(ROOT, "profiles", pat))) if pat else []
        sq = None
        if tf:
            try:
                tj = json.load(open(tf[-1]))
                rf["traffic"] = tj["traffic_bytes_per_launch"]
                sq = tj.get("sq_per_launch")
            except Exception:       # noqa: BLE001
                pass
        rf["note"] = ("with several pairs in flight one association launch serves the same round of up to 4 contexts "
                      "(algorithmic_bytes_per_launch says how many); launch duration from HIP events (hipExtLaunchKernelGGL start/stop) on the "
                      "context stream over the timed region; with several pairs in flight a launch shares the chip with other streams' kernels "
                      "and its start marker waits for the command processor, so this reads higher than a kernel trace of the same run "
                      "(profiles/*_summary.txt splits the trace by phase); single_pair.assoc_avg_launch_us is the kernel alone; traffic = HBM-side "
                      "bytes per launch of the kernel alone from the committed PMC passes (profiles/*_traffic.json)")
        single = main_leg.get("single_pair")
        if sq and sq.get("SQ_INSTS_VALU"):
            t_alone = (single or {}).get("assoc_avg_launch_us", rf["avg_launch_us"]) * 1e-6
            rf["valu"] = {"wave_insts_per_launch": sq["SQ_INSTS_VALU"], "salu": sq.get("SQ_INSTS_SALU"), "lds": sq.get("SQ_INSTS_LDS"),
                          "issue_slots_per_launch_at_2p4GHz": 1024 * 2.4e9 / 4 * t_alone,
                          "note": "from the committed PMC pass (profiles/*_traffic.json); launch time = the kernel alone"}
        line = {
            "metric": METRIC, "value": main_leg["pairs_per_s"], "unit": "scan-pairs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": main_leg["ms_per_step"], "higher_is_better": True,
            "scaling": "weak" if a.mode == "replicas" else "strong", "vs_baseline": None,
            "dtype": "f32 association / f64 residuals+solve", "data": "synthetic",
            "config": {"workload": main_leg["workload"], "pairs_in_flight_per_gpu": main_leg["pairs_in_flight_per_gpu"], "mode": a.mode,
                       "Nq": main_leg["Nq"], "Nt": main_leg["Nt"], "lm_evaluations_per_pair": main_leg["lm_evaluations_per_pair"],
                       "valid_correspondences_last_round": main_leg["valid_correspondences_last_round"],
                       "algorithmic_bytes_per_pair": main_leg["algorithmic_bytes_per_pair"]},
            "achieved_hbm_GBs_whole_path": main_leg["achieved_hbm_GBs_whole_path"],
            "roofline": rf,
            "solution_x": main_leg["solution_x"],
        }
        if "communicator" in main_leg:
            line["config"]["communicator"] = main_leg["communicator"]
        if single is not None:
            line["single_pair"] = single
        if "shared_target" in main_leg:
            line["shared_target"] = main_leg["shared_target"]
        if legs:
            line["configs"] = {k: {kk: vv for kk, vv in v.items() if kk not in ("solution_x",)} for k, v in legs.items()}
            # the 2M-point map leg carries its own committed PMC pass (profiles/rNN_c4_traffic.json)
            tf4 = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_c4_traffic.json")))
            if tf4 and "c4" in line["configs"] and isinstance(line["configs"]["c4"].get("roofline"), dict):
                try:
                    line["configs"]["c4"]["roofline"]["traffic"] = json.load(open(tf4[-1]))["traffic_bytes_per_launch"]
                    line["configs"]["c4"]["roofline"]["traffic_note"] = "HBM-side bytes per launch of ONE context's round alone (committed PMC passes)"
                except Exception:       # noqa: BLE001
                    pass
        if modes:
            line["modes"] = {k: {kk: vv for kk, vv in v.items() if kk not in ("roofline",)} for k, v in modes.items()}   # solution_x stays: tests compare it with a single-rank call
        if not a.no_cpu_baseline and world == 1:             # rank 0 at N = 1 only: the other runs just report the GPU side
            d, vis, _, icp_skip = make_workload(a.workload)
            cb = cpu_baseline(d, vis, a.cpu_sample_skip, icp_skip)
            xo = np.array(cb.pop("x"))
            xg = np.array(main_leg["solution_x"])
            cb["pose_diff_vs_gpu"] = {"dt_m": float(np.linalg.norm(xo[3:] - xg[3:])), "dw_rad": float(np.linalg.norm(xo[:3] - xg[:3]))}
            line["cpu_baseline"] = cb
        print(json.dumps(line), flush=True)
    if rig.dist is not None:
        rig.dist.barrier()
        rig.dist.destroy_process_group()


if __name__ == "__main__":
    main()
