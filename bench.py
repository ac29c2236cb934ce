#!/usr/bin/env python3
"""bench.py -- scan-pairs/s of the MI355X scan-matching core on BASELINE.json's workload.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one batch of B complete scan-pair registrations: from the two raw ring clouds already resident
in HBM to the solved pose -- target index build (velo_set_target), query list (velo_set_source) and
velo_frame_to_frame (6 association rounds + 6 Levenberg-Marquardt solves to Ceres-default tolerances).
Workload (configs[1] of BASELINE.json): synthetic HDL-64E pair, 64 x 1875 = 120,000 points each, icp_skip = 1.

Multi-GPU (--mode):
  replicas (default)  every rank registers its own pairs, no data-path collective           -> "scaling": "weak"
  sharded             ONE pair per step, queries split 1/N per rank, RCCL all-reduce of the
                      28-double normal-equation block every LM evaluation (north_star)      -> "scaling": "strong"
  target-sharded      BASELINE config 5: ONE pair per step, each rank holds a block of whole target rings,
                      per-query top-2 records all-to-all each association round, then as "sharded"  -> "strong"
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--mode", choices=["replicas", "sharded", "target-sharded"], default="replicas")
    ap.add_argument("--workload", choices=["c1", "c2", "c3", "c4"], default="c2",
                    help="c2: 120k pair; c3: + 2000 stereo blocks; c4: 120k scan vs 2M-point map")
    ap.add_argument("--batch", type=int, default=8, help="independent pairs in flight per GPU (one context + stream each)")
    ap.add_argument("--threads-per-pair", dest="batch_api", action="store_false",
                    help="drive every pair from its own host thread (frame_to_frame) instead of velo_frame_to_frame_batch")
    ap.add_argument("--separate-loads", action="store_true",
                    help="A/B: velo_set_target/source from B host threads, then velo_frame_to_frame_batch (instead of velo_register_batch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend for the barrier / max-over-ranks (nccl = RCCL)")
    ap.add_argument("--force-device", type=int, default=None, help="testing only: every rank uses this device (with --dist-backend gloo)")
    ap.add_argument("--cpu-sample-skip", type=int, default=4)
    return ap.parse_args()


def make_workload(name):
    """-> (scan pair, visual matches or None, label, icp_skip)"""
    from velo_amd import synth
    if name == "c1":
        # configs[0], the reference's own constants (kitti.h:8: icp_skip = 200 -> 640 queries per round).  KITTI seq 00 is not in
        # this image: the synthetic pair stands in, in the same ring layout the KITTI reader produces.
        return synth.scan_pair(), None, "configs[0] stand-in: synthetic 120k-pt pair, reference constants (icp_skip=200)", 200
    if name == "c4":
        d = synth.scan_to_map(2_000_000)
        label = "synthetic HDL-64E 120k-pt scan vs 2M-pt accumulated map (configs[3]), icp_skip=1"
    else:
        d = synth.scan_pair()
        label = "synthetic HDL-64E 64x1875=120k-pt scan pair (configs[1]), icp_skip=1, point-to-plane ICP"
    vis = None
    if name == "c3":
        vis = synth.stereo_matches(1000)
        label = "configs[2]: 120k-pt pair + 2000 stereo reprojection blocks, icp_skip=1"
    return d, vis, label, 1


def cpu_baseline(d, vis, sample_skip, icp_skip=1):
    """The CPU restatement (oracle = 'port') timed on this host: (i) all cores on the full pair,
    (ii) one thread -- the reference's configuration (velo.h:900) -- on a 1/sample_skip query sample."""
    import oracle_lib
    cores = oracle_lib.max_threads()
    out = {}
    o = oracle_lib.Oracle(threads=cores, icp_skip=icp_skip)
    if icp_skip > 1:
        sample_skip = 1                      # sparse queries already: the single-thread leg runs the whole pair
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
    t_one = (time.perf_counter() - t0) * sample_skip
    out = {
        "value": 1.0 / t_all, "unit": "scan-pairs/s", "cores": cores, "kind": "port",
        "sample": f"1 full pair (icp_skip={icp_skip}) on {cores} OpenMP threads = {t_all:.2f} s; single-thread "
                  f"(reference configuration) extrapolated from a 1/{sample_skip} query sample = {t_one:.1f} s/pair",
        "single_thread_pairs_per_s": 1.0 / t_one,
        "x": [float(v) for v in x],
    }
    return out


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import velo_amd  # noqa: F401
    from velo_amd import api
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.force_device is not None:
            local_rank = a.force_device
        torch.cuda.set_device(local_rank)
        if a.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=a.dist_backend)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    d, vis, label, icp_skip = make_workload(a.workload)
    B = 1 if a.mode != "replicas" else max(1, a.batch)
    # inputs resident in HBM before the timed region (torch is only the allocator here)
    tgt_off, tgt_first_ring, tgt_first_point = d["tgt_off"], 0, 0
    tgt_np = d["tgt_xyz"]
    if a.mode == "target-sharded" and world > 1:
        from velo_amd import shard
        r0, r1, p0, tgt_off = shard.target_ring_block(d["tgt_off"], rank, world)
        tgt_first_ring, tgt_first_point = r0, p0
        tgt_np = d["tgt_xyz"][p0:p0 + int(tgt_off[-1])]
    tgt = torch.from_numpy(np.ascontiguousarray(tgt_np)).to(dev)
    src = torch.from_numpy(d["src_xyz"]).to(dev)
    torch.cuda.synchronize()
    ctxs = [api.Context(local_rank, icp_skip=icp_skip) for _ in range(B)]
    for c in ctxs:
        c.set_timing(True)
        if vis is not None:
            c.set_visual(vis)
    if a.mode != "replicas" and world > 1:
        uid = [api.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        ctxs[0].comm_init(uid[0], rank, world)
        if a.mode == "target-sharded":
            ctxs[0].comm_set_target_sharded(True)

    results = [None] * B

    def load_pair(i):
        c = ctxs[i]
        c.set_target_part(tgt, tgt_off, tgt_first_ring, tgt_first_point)
        c.set_source(src, d["src_off"])

    def one_pair(i):
        load_pair(i)
        results[i] = ctxs[i].frame_to_frame(d["x0"])

    pool = ThreadPoolExecutor(max_workers=B) if B > 1 else None
    batch_refs = (api.scan_refs([(tgt, tgt_off)] * B, local_rank), api.scan_refs([(src, d["src_off"])] * B, local_rank)) if B > 1 else None
    x0s = np.tile(np.asarray(d["x0"], dtype=np.float64), (B, 1))

    def step():
        if pool is None:
            one_pair(0)
        elif a.batch_api and a.separate_loads:
            # A/B: index builds from B host threads (one library call each), then the registrations through the batch entry point
            list(pool.map(load_pair, range(B)))
            xs, Ts, Ss = api.frame_to_frame_batch(ctxs, x0s)
            for i in range(B):
                results[i] = (xs[i], Ts[i], Ss[i])
        elif a.batch_api and a.mode == "replicas":
            # the B pairs' scans (device pointers) and the B registrations in ONE library call: every group of contexts builds its
            # indices and starts registering on its own thread (velo_register_batch)
            xs, Ts, Ss = api.register_batch(ctxs, None, None, x0s, refs=batch_refs)
            for i in range(B):
                results[i] = (xs[i], Ts[i], Ss[i])
        else:
            list(pool.map(one_pair, range(B)))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        for c in ctxs:
            c.synchronize()

    for _ in range(a.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    assoc_ms, assoc_n, alg_bytes, assoc_bytes, evals = 0.0, 0, 0, 0, 0
    for _ in range(a.steps):
        step()
        for r in results:
            s = r[2]
            assoc_ms += s.assoc_kernel_ms
            assoc_n += s.assoc_kernel_launches
            alg_bytes += s.algorithmic_bytes
            assoc_bytes += s.assoc_bytes
            evals += sum(s.solves[k].evaluations for k in range(s.n_solves))
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev if a.dist_backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # SURVEY 8(d) asks for the single-pair latency next to the throughput: a short untimed-by-the-contract leg, one pair in flight
    single = None
    if world == 1 and a.mode == "replicas" and B > 1:
        for _ in range(3):
            one_pair(0)
        ctxs[0].synchronize()
        t1 = time.perf_counter()
        n1 = 20
        a_ms, a_n = 0.0, 0
        for _ in range(n1):
            one_pair(0)
            a_ms += results[0][2].assoc_kernel_ms
            a_n += results[0][2].assoc_kernel_launches
        ctxs[0].synchronize()
        lat = (time.perf_counter() - t1) / n1
        single = {"pairs_in_flight": 1, "ms_per_pair": 1e3 * lat, "pairs_per_s": 1.0 / lat,
                  "assoc_avg_launch_us": 1e3 * a_ms / max(a_n, 1)}

    pairs_per_rank = a.steps * B
    total_pairs = pairs_per_rank * (world if a.mode == "replicas" else 1)
    value = total_pairs / dt
    x_gpu = results[0][0]
    s0 = results[0][2]

    if rank == 0:
        n_pairs_timed = a.steps * B
        per_pair_bytes = alg_bytes / max(n_pairs_timed, 1)
        # dominant kernel: the association search.  Algorithmic bytes per launch = 12 Nq + 12 Nt + 28 Nq
        b_launch = assoc_bytes / max(assoc_n, 1)            # the batch driver serves the same round of several contexts with one launch
        avg_ms = assoc_ms / max(assoc_n, 1)
        achieved = (b_launch / 1e9) / (avg_ms / 1e3) if avg_ms > 0 else 0.0
        traffic = None          # HBM-side bytes per launch from the committed PMC passes (tools/summarize_traffic.py)
        import glob
        tf = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")))
        sq = None
        if tf and a.workload == "c2":
            try:
                tj = json.load(open(tf[-1]))
                traffic = tj["traffic_bytes_per_launch"]
                sq = tj.get("sq_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "scan-pairs/sec + achieved HBM GB/s, 120k-pt HDL-64E frame-to-frame ICP",
            "value": value, "unit": "scan-pairs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True,
            "scaling": "weak" if a.mode == "replicas" else "strong", "vs_baseline": None,
            "dtype": "f32 association / f64 residuals+solve", "data": "synthetic",
            "config": {"workload": label, "pairs_in_flight_per_gpu": B, "mode": a.mode,
                       "Nq": int(s0.n_queries), "Nt": int(s0.n_target),
                       "lm_evaluations_per_pair": evals / max(n_pairs_timed, 1),
                       "valid_correspondences_last_round": int(s0.solves[s0.n_solves - 1].n_icp_valid),
                       "algorithmic_bytes_per_pair": per_pair_bytes},
            "achieved_hbm_GBs_whole_path": per_pair_bytes * value / 1e9,
            "roofline": {"bound": "hbm", "kernel": "assoc_search_v5_batch_kernel" if (B > 1 and a.batch_api and a.mode == "replicas") else "assoc_search_v5_kernel",
                         "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "avg_launch_us": avg_ms * 1e3, "algorithmic_bytes_per_launch": b_launch,
                         "note": "with several pairs in flight one association launch serves the same round of up to 4 contexts "
                                 "(algorithmic_bytes_per_launch says how many); "
                                 "launch duration from HIP events (hipExtLaunchKernelGGL start/stop) on the context stream over the timed "
                                 "region; with several pairs in flight a launch shares the chip with other streams' kernels and its start "
                                 "marker waits for the command processor, so this reads higher than a kernel trace of the same run "
                                 "(profiles/*_summary.txt splits the trace by phase); single_pair.assoc_avg_launch_us is the kernel alone"},
            "solution_x": [float(v) for v in x_gpu],
        }
        if single is not None:
            line["single_pair"] = single
        if sq and sq.get("SQ_INSTS_VALU"):
            # the bound that binds: VALU issue.  1024 SIMDs, one wave-instruction per 4 cycles each (packed-f32 and f64 ops take 8)
            t_alone = (single or {}).get("assoc_avg_launch_us", avg_ms * 1e3) * 1e-6
            line["roofline"]["valu"] = {"wave_insts_per_launch": sq["SQ_INSTS_VALU"], "salu": sq.get("SQ_INSTS_SALU"), "lds": sq.get("SQ_INSTS_LDS"),
                                        "issue_slots_per_launch_at_2p4GHz": 1024 * 2.4e9 / 4 * t_alone,
                                        "note": "from the committed PMC pass (profiles/*_traffic.json); launch time = the kernel alone"}
        if not a.no_cpu_baseline and world == 1:             # rank 0 at N = 1 only: the other runs just report the GPU side
            cb = cpu_baseline(d, vis, a.cpu_sample_skip, icp_skip)
            xo = np.array(cb.pop("x"))
            cb["pose_diff_vs_gpu"] = {"dt_m": float(np.linalg.norm(xo[3:] - x_gpu[3:])),
                                      "dw_rad": float(np.linalg.norm(xo[:3] - x_gpu[:3]))}
            line["cpu_baseline"] = cb
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        for c in ctxs:
            c.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
