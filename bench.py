#!/usr/bin/env python3
"""bench.py -- scan-pairs/s of the MI355X scan-matching core on BASELINE.json's workload.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: under python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ..., or plainly: without a launcher the
     script starts its N ranks itself, as a child process; a launcher whose WORLD_SIZE is not N is an error)

One "step" = B DRIVES advancing one frame each (the reference's loop, main.cpp:305-413, B sequences at a time): context i registers
frame k+1 of drive i against frame k -- the previous frame, registered as source in the last step, is promoted to target ON THE DEVICE
(velo_source_to_target through VELO_SCAN_PROMOTE: sd_prev, main.cpp:233,380 -- buffer swap + index build), the new frame comes from
the ring clouds already resident in HBM, the initial guess is the constant-velocity prediction from the drive's last two poses
(main.cpp:311-331; the start-up guess {0,0,0,0,0,1} for a drive's first pair, main.cpp:170) -- target index build, query list and
frame_to_frame (6 association rounds + 6 Levenberg-Marquardt solves to Ceres-default tolerances) of every pair, through ONE library
call (velo_register_batch).  Every timed registration is a pair its context has never seen (steps x B different pairs; synth.drive:
own scene, noise and speed / yaw profile per drive), so the lock-step groups diverge and the chain's launch prediction is a real
prediction; `chain` reports its calls and misses.  `--same-pairs` re-registers B fixed pairs every step (what round 3 measured).
The headline (`value`) is workload configs[1] of BASELINE.json: synthetic HDL-64E pairs, 64 x 1875 = 120,000 points each,
icp_skip = 1; at N > 1 every rank registers its own pairs (replicas, no data-path collective -> "scaling": "weak").

The same JSON line also carries
  roofline  the kernel with the largest share of the timed region's kernel time (HIP events on the launching stream, per kernel
            name, velo_set_timing(ctx, 2)) with its algorithmic bytes, achieved GB/s and fraction of the HBM peak; `kernels` = the
            same for the top three
  configs   short legs of the other single-GPU workloads (c1: reference constants -- on KITTI frames when VELO_KITTI_ROOT names a
            dataset root --, c3: + 2,000 stereo blocks, c4: 2M-point map shared by the B contexts, canonical_pair: the start-up-guess pair of
            rounds 1-3), each a complete bench of its own IN A CHILD PROCESS of its own, each with its cpu_baseline                (N = 1)
  host_inputs  the headline's drives with their frames in pageable host memory, uploaded inside the step (PCIe-inclusive; never `value`)
  against_simulated_motion  every timed pair's pose against the drive's simulated motion
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
import gc
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
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mode", choices=["replicas", "sharded", "target-sharded"], default="replicas")
    ap.add_argument("--workload", choices=["c1", "c2", "c3", "c4"], default="c2",
                    help="c2: 120k pairs; c3: + 2000 stereo blocks; c4: 120k scans vs 2M-point map; c1: reference constants (icp_skip=200)")
    ap.add_argument("--batch", type=int, default=8, help="independent pairs in flight per GPU (one context + stream each)")
    ap.add_argument("--same-pairs", action="store_true", help="A/B: B fixed, different pairs re-registered every step instead of B drives (what round 3 measured)")
    ap.add_argument("--same-pair", action="store_true", help="A/B: every context registers the canonical pair (what rounds 1-2 measured)")
    ap.add_argument("--sequences", action="store_true", help="A/B: ONE velo_register_sequences call for the timed frames (the lock-step groups walk their drives independently, no barrier "
                    "between frames) instead of one velo_register_batch call per step.  Measured: the groups drift apart, every kernel runs faster in the mix "
                    "(association 134 vs 149 us, LM launch 22.1 vs 23.7 us) but the call ends with its slowest group (3,297 vs 3,486 pairs/s)")
    ap.add_argument("--sequences-lockstep", action="store_true", help="A/B: ONE velo_register_sequences call with VELO_SEQ_LOCKSTEP (a barrier between frames inside the library): "
                    "the per-step form without the harness's interpreter in the loop")
    ap.add_argument("--own-map-copies", action="store_true", help="c4 A/B: every context holds its own copy of the 2M-point map and builds its own index every step (rounds 1-4)")
    ap.add_argument("--host-inputs", action="store_true", help="the drives' frames stay in host memory (numpy): every step uploads its B frames -- the PCIe-inclusive rate (never the headline)")
    ap.add_argument("--gen-procs", type=int, default=0, help="worker processes that synthesise the drives' frames (0: min(16, host cores))")
    ap.add_argument("--threads-per-pair", dest="batch_api", action="store_false",
                    help="drive every pair from its own host thread (frame_to_frame) instead of velo_register_batch")
    ap.add_argument("--separate-loads", action="store_true",
                    help="A/B: velo_set_target/source from B host threads, then velo_frame_to_frame_batch (instead of velo_register_batch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-oracle-check", action="store_true", help="skip the off-the-clock replay of two timed pairs through the CPU oracle (A/B tools)")
    ap.add_argument("--legs-in-process", action="store_true", help="A/B: the c1 / c3 / c4 / host-inputs / canonical-pair legs inside this process, behind the main leg, instead of one child process each")
    ap.add_argument("--no-legs", action="store_true", help="only the timed workload: no c1/c3/c4 legs, no multi-GPU mode legs")
    ap.add_argument("--comm", choices=["peer", "rccl"], default="peer", help="all-reduce of the sharded mode: peer-mapped slabs or RCCL")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend for the barrier / max-over-ranks (nccl = RCCL)")
    ap.add_argument("--force-device", type=int, default=None, help="testing only: every rank uses this device (with --dist-backend gloo)")
    ap.add_argument("--detail-out", default=None, help="where the FULL record of the run goes (default: bench_detail.json next to this script); stdout carries the compact line only")
    ap.add_argument("--timing", type=int, default=2, help="velo_set_timing level inside the timed region (2: per-kernel brackets; 1: association only)")
    return ap.parse_args()


_cache = {}


def kitti_pairs(root, n):
    """BASELINE configs[0] on real data: consecutive frame pairs k -> k+1 of KITTI odometry sequence 00 under
    VELO_KITTI_ROOT (the dataset root with sequences/00/velodyne/*.bin, or a directory of .bin files), segmented ON THE DEVICE by
    velo_set_scan_velodyne (kitti.h:121-185) with the sequence's calib.txt `Tr` when present.  None when the files are not there."""
    from velo_amd import api, synth
    for vd in (os.path.join(root, "sequences", "00", "velodyne"), os.path.join(root, "velodyne"), root):
        files = sorted(glob.glob(os.path.join(vd, "*.bin")))
        if len(files) >= 2:
            break
    else:
        return None
    M = synth.VELO_TO_CAM.astype(np.float32)
    calib = os.path.join(os.path.dirname(vd.rstrip("/")), "calib.txt")
    if os.path.exists(calib):
        for ln in open(calib):
            if ln.startswith("Tr:") or ln.startswith("Tr "):
                v = np.array(ln.split()[1:13], dtype=np.float64).reshape(3, 4)
                M = np.vstack([v, [0, 0, 0, 1]]).astype(np.float32)
    ctx = api.Context(int(os.environ.get("LOCAL_RANK", "0")))
    scans = []
    for f in files[:n + 1]:                                  # the device segmenter gives the ring clouds the path takes
        rec = np.fromfile(f, dtype=np.float32).reshape(-1, 4)
        ctx.set_scan_velodyne(False, rec, M)
        scans.append((ctx.cloud(False), ctx.ring_offsets(False)))
    ctx.close()
    x0 = synth.INITIAL_GUESS.copy()                          # main.cpp:170; later frames would use the previous motion
    return [dict(tgt_xyz=scans[k][0], tgt_off=scans[k][1], src_xyz=scans[k + 1][0], src_off=scans[k + 1][1], x0=x0, x_true=None)
            for k in range(min(n, len(scans) - 1))]


_drives = {}


def _drive_frame_job(job):
    from velo_amd import synth
    plan, k = job
    return synth.drive_frame(plan, k)


def make_drives(B, n_frames, procs=0):
    """B planned drives of n_frames frames each (synth.drive_plan / drive_frame), the frames synthesised by a pool of worker PROCESSES
    (0.3 s of numpy ray casting per 120k-point frame).  Called BEFORE anything touches the GPU (fork).  Cached: the c1 / c3 legs run on
    the same drives as the headline."""
    key = (B, n_frames)
    if key in _drives:
        return _drives[key]
    from velo_amd import synth
    plans = [synth.drive_plan(n_frames, seed) for seed in range(B)]
    # VELO_DRIVE_CACHE=<dir>: frames synthesised by an earlier run are read back (profiling passes: rocprofv3's preloaded library has
    # initialised the GPU before this program starts, and a process in that state should not fork workers).  Drives of up to 42 frames
    # share their scene and their first poses, so a shorter run takes a prefix of a longer one's frames.
    cache = os.environ.get("VELO_DRIVE_CACHE")
    cache_file = os.path.join(cache, f"drives_b{B}.npz") if cache else None
    # what the cached frames were made from: generator revision, every drive's scene and poses -- a cache written by another synth.drive_plan
    # (or another seed / scan shape) is regenerated, never silently reused
    import hashlib
    dg = hashlib.sha256()
    dg.update(repr((getattr(synth, "DRIVE_REVISION", 1), synth.N_BEAMS, synth.N_AZIMUTH, B)).encode())
    for p in plans:
        dg.update(repr(sorted(p["scene_kw"].items())).encode())
        dg.update(np.ascontiguousarray(np.asarray(p["poses_velo"], dtype=np.float64)[:min(n_frames, 42)]).tobytes() if n_frames <= 42 else b"")
    digest = dg.hexdigest()
    if cache_file and os.path.exists(cache_file) and n_frames <= 42:
        try:
            z = np.load(cache_file)
            nf_c = int(z["n_frames"])
            dg_c = hashlib.sha256()
            dg_c.update(repr((getattr(synth, "DRIVE_REVISION", 1), synth.N_BEAMS, synth.N_AZIMUTH, B)).encode())
            plans_c = [synth.drive_plan(nf_c, seed) for seed in range(B)] if nf_c != n_frames else plans
            for p in plans_c:
                dg_c.update(repr(sorted(p["scene_kw"].items())).encode())
                dg_c.update(np.ascontiguousarray(np.asarray(p["poses_velo"], dtype=np.float64)[:min(nf_c, 42)]).tobytes())
            same = str(z["digest"]) == dg_c.hexdigest() if "digest" in z.files else False
            prefix_ok = all(np.array_equal(np.asarray(pc["poses_velo"])[:n_frames], np.asarray(p["poses_velo"])[:n_frames]) for pc, p in zip(plans_c, plans))
            if same and prefix_ok and nf_c >= n_frames and nf_c <= 42 and int(z["B"]) == B:
                for i in range(B):
                    plans[i]["frames"] = [(z[f"xyz_{i}_{k}"], z[f"off_{i}_{k}"]) for k in range(n_frames)]
                print(f"[bench] {B} drives x {n_frames} frames read from {cache_file}", file=sys.stderr, flush=True)
                _drives[key] = plans
                return plans
            print(f"[bench] drive cache {cache_file} was made from other drives; synthesising", file=sys.stderr, flush=True)
        except Exception as e:       # noqa: BLE001
            print(f"[bench] drive cache unreadable ({e}); synthesising", file=sys.stderr, flush=True)
    # a profiler's preloaded library may have initialised the GPU before this program started: no forked workers then
    profiled = any("rocprof" in str(os.environ.get(k, "")).lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_LIBRARY")) \
        or any(k.startswith("ROCPROF") for k in os.environ)
    if profiled:
        procs = 1
    jobs = [(plans[i], k) for i in range(B) for k in range(n_frames)]
    procs = procs or max(1, min(16, (os.cpu_count() or 1) // max(1, int(os.environ.get("WORLD_SIZE", "1")))))
    t0 = time.perf_counter()
    frames = None
    if procs > 1:
        try:
            import multiprocessing as mp
            with mp.get_context("fork").Pool(procs) as pool:
                frames = pool.map(_drive_frame_job, jobs, chunksize=1)
        except Exception as e:       # noqa: BLE001
            print(f"[bench] frame pool unavailable ({e}); generating in-process", file=sys.stderr, flush=True)
    if frames is None:
        frames = [_drive_frame_job(j) for j in jobs]
    for i in range(B):
        plans[i]["frames"] = frames[i * n_frames:(i + 1) * n_frames]
    print(f"[bench] {B} drives x {n_frames} frames synthesised in {time.perf_counter() - t0:.1f} s ({procs} processes)", file=sys.stderr, flush=True)
    if cache_file:
        try:
            os.makedirs(cache, exist_ok=True)
            arrs = {"n_frames": np.int64(n_frames), "B": np.int64(B), "digest": np.str_(digest)}
            for i in range(B):
                for k in range(n_frames):
                    arrs[f"xyz_{i}_{k}"], arrs[f"off_{i}_{k}"] = plans[i]["frames"][k]
            np.savez(cache_file, **arrs)
        except Exception as e:       # noqa: BLE001
            print(f"[bench] drive cache not written ({e})", file=sys.stderr, flush=True)
    _drives[key] = plans
    return plans


def make_workload(name, B, same_pair=False):
    """-> dict(label, icp_skip, pairs: list of B dicts (tgt_xyz, tgt_off, src_xyz, src_off, x0, vis), shared_map: bool, distinct: int)"""
    from velo_amd import synth
    key = (name, B, same_pair)
    if key in _cache:
        return _cache[key]
    n = 1 if same_pair else B
    if name == "c4":
        m = synth.scan_to_map(2_000_000, n_queries=n)
        qs = m.get("queries") or [m]
        pairs = [dict(tgt_xyz=m["tgt_xyz"], tgt_off=m["tgt_off"], src_xyz=q["src_xyz"], src_off=q["src_off"], x0=q["x0"], vis=None) for q in qs]
        label, skip = "synthetic HDL-64E 120k-pt scans vs 2M-pt accumulated map (configs[3]), icp_skip=1", 1
    else:
        src = None
        if name == "c1" and os.environ.get("VELO_KITTI_ROOT"):
            src = kitti_pairs(os.environ["VELO_KITTI_ROOT"], n)
        kitti = src is not None
        if src is None:
            src = synth.distinct_pairs(n)
        pairs = [dict(d, vis=None) for d in src]
        if name == "c3":
            for k, d in enumerate(pairs):
                d["vis"] = synth.stereo_matches(1000, seed=3 + k, x_true=d["x_true"])
        label, skip = {
            "c1": (("configs[0]: KITTI seq 00 consecutive frames under VELO_KITTI_ROOT" if kitti else
                    "configs[0] stand-in: synthetic 120k-pt pairs in the KITTI ring layout") + ", reference constants (icp_skip=200)", 200),
            "c3": ("configs[2]: 120k-pt pairs + 2000 stereo reprojection blocks each, icp_skip=1", 1),
            "c2": ("synthetic HDL-64E 64x1875=120k-pt scan pairs (configs[1]), icp_skip=1, point-to-plane ICP", 1),
        }[name]
    distinct = len(pairs)
    pairs = [pairs[i % len(pairs)] for i in range(B)]
    out = dict(label=label, icp_skip=skip, pairs=pairs, shared_map=(name == "c4"), distinct=distinct)
    _cache[key] = out
    return out


def cpu_baseline(d, vis, icp_skip=1, single_thread=True):
    """The CPU restatement (oracle = 'port') timed on this host on the canonical pair of the workload: (i) all cores, (ii) ONE thread --
    the reference's configuration (velo.h:900) -- on the WHOLE pair, nothing extrapolated (about 15 s at icp_skip = 1).
    single_thread=False (the 2M-point map: 1,067 searches per query, minutes on one thread): the all-cores run only."""
    import oracle_lib
    cores = oracle_lib.max_threads()

    def one(threads):
        o = oracle_lib.Oracle(threads=threads, icp_skip=icp_skip)
        t0 = time.perf_counter()
        o.set_target(d["tgt_xyz"], d["tgt_off"])
        o.set_source(d["src_xyz"], d["src_off"])
        if vis is not None:
            o.set_visual(vis)
        x, _, _ = o.frame_to_frame(d["x0"])
        return time.perf_counter() - t0, x

    t_all, x = one(cores)
    out = {"value": 1.0 / t_all, "unit": "scan-pairs/s", "cores": cores, "kind": "port",
           "sample": f"the first pair of the workload (icp_skip={icp_skip}), whole: {cores} OpenMP threads = {t_all:.2f} s", "x": [float(v) for v in x]}
    if single_thread:
        t_one, _ = one(1)
        out["sample"] += f"; 1 thread (the reference's configuration, velo.h:900) = {t_one:.2f} s"
        out.update(single_thread_pairs_per_s=1.0 / t_one, single_thread_s_per_pair=t_one, single_thread_extrapolated=False)
    else:
        out["sample"] += "; the one-thread run (the reference's configuration, velo.h:900) is not taken on this workload: minutes per pair"
    return out


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


def kernel_table(acc):
    """{name: [ms, launches, bytes]} -> rows sorted by share of the summed kernel time, each with its roofline figures"""
    total = sum(v[0] for v in acc.values()) or 1.0
    rows = []
    for name, (ms, n, b) in acc.items():
        if n <= 0:
            continue
        ach = (b / 1e9) / (ms / 1e3) if ms > 0 and b > 0 else 0.0
        rows.append({"kernel": name, "share": ms / total, "launches": int(n), "avg_launch_us": 1e3 * ms / n,
                     "algorithmic_bytes_per_launch": b / n, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": ach / HBM_PEAK_GBS})
    rows.sort(key=lambda r: -r["share"])
    return rows


LINE_LIMIT = 8192          # bytes: the driver's parser gave up on a 20 KB line (BENCH_r05.json: parsed null); 13.7 KB still parsed.  Keep well below.


def _sig(v, n=6):
    """floats to n significant digits (the line is a report; full precision lives in the detail file)"""
    if isinstance(v, bool) or v is None:
        return v
    if isinstance(v, float):
        return float(f"{v:.{n}g}") if np.isfinite(v) else None
    if isinstance(v, dict):
        return {k: _sig(x, n) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_sig(x, n) for x in v]
    return v


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def compact_leg(v):
    """what the line keeps of a config leg (the full leg is in the detail file)"""
    if not isinstance(v, dict):
        return v
    if "error" in v and v["error"]:
        return {"error": str(v["error"])[-160:]}
    out = _pick(v, ("pairs_per_s", "ms_per_step", "steps", "lm_evaluations_per_pair", "of_resident_rate", "solution_equal_to_resident"))
    rf = v.get("roofline") or {}
    if rf:
        out.update(_pick(rf, ("kernel", "avg_launch_us", "frac")))
        out["traffic"] = rf.get("traffic")
    if isinstance(v.get("chain"), dict):
        out["chain_misses"] = v["chain"].get("misses")
    cb = v.get("cpu_baseline") or {}
    if cb:
        out["cpu_baseline"] = _pick(cb, ("value", "cores", "single_thread_pairs_per_s"))
        if "pose_diff_vs_gpu" in cb:
            out["pose_diff_vs_gpu"] = cb["pose_diff_vs_gpu"]
    if isinstance(v.get("single_pair"), dict):
        out["single_pair_ms"] = v["single_pair"].get("ms_per_pair")
    for kk in ("timed_pairs_vs_oracle", "against_simulated_motion"):
        if isinstance(v.get(kk), dict):
            out[kk] = _pick(v[kk], ("dt_m", "dw_rad", "max_dt_m", "max_dw_rad", "pairs", "counts_equal", "ok"))
    for kk in ("own_map_copies", "shared_target"):
        if isinstance(v.get(kk), dict):
            out[kk + "_pairs_per_s"] = v[kk].get("pairs_per_s")
    return out


def compact_mode(v):
    """what the line keeps of a multi-GPU mode leg: the rate, the communicator read back, the pose (tests compare it), a one-line error"""
    if not isinstance(v, dict):
        return v
    out = _pick(v, ("pairs_per_s", "ms_per_step", "steps", "communicator", "lm_evaluations_per_pair"))
    if v.get("solution_x") is not None:
        out["solution_x"] = v["solution_x"]              # full precision: tests hold it to a single-rank call
    if v.get("error"):
        out["error"] = str(v["error"])[-200:]
    if isinstance(v.get("first_attempt"), dict) and v["first_attempt"].get("error"):
        out["first_attempt_error"] = str(v["first_attempt"]["error"])[-200:]
    return out


def compact_line(full, detail_name=None):
    """The ONE JSON line rank 0 prints: the contract's keys, the roofline of the dominant kernel, the CPU baseline, a few figures per leg --
    never more than LINE_LIMIT bytes.  Everything else (kernel tables, notes, poses of every context, the legs in full) is `full`, which goes
    to the detail file and to stderr."""
    line = _pick(full, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling"))
    line["vs_baseline"] = full.get("vs_baseline")
    line.update(_pick(full, ("dtype", "data")))
    cfg = full.get("config") or {}
    line["config"] = _pick(cfg, ("workload", "pairs_in_flight_per_gpu", "distinct_pairs", "frames_per_drive", "mode", "Nq", "Nt", "lm_evaluations_per_pair",
                                  "algorithmic_bytes_per_pair", "communicator"))
    if len(line["config"].get("workload", "")) > 200:
        line["config"]["workload"] = line["config"]["workload"][:200]
    line.update(_pick(full, ("achieved_hbm_GBs_whole_path",)))
    if isinstance(full.get("chain"), dict):
        line["chain"] = _pick(full["chain"], ("calls", "misses"))
    rf = full.get("roofline") or {}
    crf = _pick(rf, ("kernel", "bound", "achieved", "peak", "unit", "frac"))
    crf["traffic"] = rf.get("traffic")
    crf.update(_pick(rf, ("share", "launches", "avg_launch_us", "algorithmic_bytes_per_launch", "traffic_source")))
    crf["note"] = "dominant kernel by share of the timed region's kernel time; HIP events on the launching stream; traffic = fabric-side bytes per launch from the committed PMC pass, or null"
    line["roofline"] = crf
    line["kernels"] = [dict(_pick(k, ("kernel", "share", "avg_launch_us", "algorithmic_bytes_per_launch", "frac")), traffic=k.get("traffic")) for k in (full.get("kernels") or [])[:3]]
    cb = full.get("cpu_baseline")
    if isinstance(cb, dict):
        c = _pick(cb, ("value", "unit", "cores", "kind", "sample", "single_thread_pairs_per_s", "pose_diff_vs_gpu"))
        if len(c.get("sample", "")) > 220:
            c["sample"] = c["sample"][:220]
        line["cpu_baseline"] = c
    if full.get("solution_x") is not None:
        line["solution_x"] = full["solution_x"]
    if isinstance(full.get("single_pair"), dict):
        line["single_pair"] = _pick(full["single_pair"], ("ms_per_pair", "pairs_per_s", "assoc_avg_launch_us"))
    for kk in ("timed_pairs_vs_oracle", "against_simulated_motion"):
        if isinstance(full.get(kk), dict):
            line[kk] = _pick(full[kk], ("dt_m", "dw_rad", "max_dt_m", "max_dw_rad", "pairs", "counts_equal", "which", "ok"))
    if isinstance(full.get("host_inputs"), dict):
        line["host_inputs"] = compact_leg(full["host_inputs"])
    for kk in ("own_map_copies", "shared_target"):
        if isinstance(full.get(kk), dict):
            line[kk] = _pick(full[kk], ("pairs_per_s", "ms_per_step"))
    if isinstance(full.get("configs"), dict):
        line["configs"] = {k: compact_leg(v) for k, v in full["configs"].items()}
    if isinstance(full.get("modes"), dict):
        line["modes"] = {k: compact_mode(v) for k, v in full["modes"].items()}
    if detail_name:
        line["detail"] = detail_name
    keep_precise = {"solution_x"}

    def rounded(d):
        if isinstance(d, dict):
            return {k: (v if k in keep_precise else rounded(v)) for k, v in d.items()}
        return _sig(d)
    line = rounded(line)
    # the guard: whatever a future leg adds, the line stays readable -- optional blocks go first, the contract's keys never
    for drop in ("kernels", "against_simulated_motion", "single_pair", "host_inputs", "configs", "modes"):
        if len(json.dumps(line)) <= LINE_LIMIT:
            break
        if drop in line:
            line[drop] = {"dropped": "line limit; see detail"} if drop in ("configs", "modes") else None
    assert len(json.dumps(line)) <= LINE_LIMIT, "bench line over the limit even without its optional blocks"
    return line


class DriveWalker:
    """B drives advancing in step (main.cpp:305-413 for B sequences at a time): context i holds frame k of drive i as its source; a step
    promotes it to target on the device, registers frame k+1 against it and hands the pose over -- T[k+1] = T[k] * dpose (main.cpp:408),
    next guess = pose_vec2mat(T[k]^-1 T[k+1]) (main.cpp:311-331).  One library call per step (velo_register_batch)."""

    def __init__(self, api, ctxs, frames, local_rank, vis=None, ahead=None, one_call=None):
        self.api, self.ctxs, self.frames, self.vis, self.local_rank = api, ctxs, frames, vis, local_rank
        self.B = len(ctxs)
        self.host_frames = isinstance(frames[0][0][0], np.ndarray)
        # velo_hint_next_frame before every step (A/B: VELO_BENCH_NO_AHEAD=1 -- every step loads its own frame, as in rounds 4 and before)
        self.ahead = (not os.environ.get("VELO_BENCH_NO_AHEAD")) if ahead is None else bool(ahead)
        self.n_frames = min(len(f) for f in frames)
        self.promote = api.promote_refs(self.B)
        self.src_refs = [api.scan_refs([frames[i][k] for i in range(self.B)], local_rank) for k in range(self.n_frames)]
        # the front-end's matches of every frame pair, handed over with the scans in the same call (velo_register_batch_visual)
        self.vis_refs = [api.visual_refs([vis[i][k] for i in range(self.B)]) for k in range(self.n_frames - 1)] if vis is not None else None
        self.guess_log = np.zeros((self.n_frames, self.B, 6))   # the guess every step started from (48 doubles per step: what the oracle check replays)
        self.restart()
        # A step as ONE library call (api.DriveStep: velo_register_sequences for one frame, the next one announced) with every argument
        # prepared here, off the clock -- what the loop body of a compiled caller costs.  VELO_BENCH_TWO_CALLS=1: velo_register_batch +
        # velo_pose_handoff through the general wrappers (two calls and a dozen numpy / ctypes objects per step), as rounds 4 and before.
        self.one_call = self.ahead and ((not os.environ.get("VELO_BENCH_TWO_CALLS")) if one_call is None else bool(one_call))
        if self.one_call:
            import ctypes as C
            self._call = api.DriveStep(ctxs, self.P_prev, self.x0)
            self._xs, self._Ts, self._S, self._out = self._call.outputs(self.n_frames - 1)
            self._pairs = []                                 # frames k and k + 1 of every drive, side by side
            for k in range(self.n_frames):
                arr = (api.VeloScanRef * (2 * self.B))()
                for i in range(self.B):
                    arr[i] = self.src_refs[k][0][i]
                    if k + 1 < self.n_frames:
                        arr[self.B + i] = self.src_refs[k + 1][0][i]
                self._pairs.append((arr, C.cast(arr, C.c_void_p)))
            self._vis = [(C.cast(v[0], C.c_void_p), C.cast(v[1], C.c_void_p)) for v in self.vis_refs] if self.vis_refs is not None else None

    def restart(self):
        from velo_amd import synth
        for i, c in enumerate(self.ctxs):
            c.set_source(*self.frames[i][0])                 # frame 0 waits on the device as "source"
        self.k = 0
        if getattr(self, "P_prev", None) is None:
            self.P_prev = np.ascontiguousarray(np.tile(np.eye(4), (self.B, 1, 1)))
            self.x0 = np.tile(synth.INITIAL_GUESS, (self.B, 1))  # main.cpp:170
        else:                                                # (in place: a prepared step holds their addresses)
            self.P_prev[...] = np.eye(4)
            self.x0[...] = synth.INITIAL_GUESS
        self.first = None

    def step(self):
        k = self.k + 1
        self.guess_log[k] = self.x0
        if self.one_call:
            j = k - 1
            self._call(self._pairs[k][1], self._out[j], self._S[j], self._vis[j] if self._vis is not None else None, announce=k + 1 < self.n_frames)
            self.k = k
            if k == 1:
                self.first = self._xs[j].copy()
            return self._xs[j], self._Ts[j], self._S[j]
        if k + 1 < self.n_frames and self.ahead:             # the next step announced: its promotion, ingest and index build (and, for frames in host
            self.api.hint_next_frames(self.ctxs, self.src_refs[k + 1][0])     # memory, the upload) are enqueued behind this step's launches
        elif self.host_frames and k + 1 < self.n_frames:     # frames in host memory: the NEXT frames' uploads run under this step's launches
            self.api.hint_next_sources(self.ctxs, self.src_refs[k + 1][0])
        xs, Ts, Ss = self.api.register_batch(self.ctxs, None, None, self.x0, refs=(self.promote, self.src_refs[k]),
                                             visual=self.vis_refs[k - 1] if self.vis_refs is not None else None)
        # T[k] = T[k-1] dpose (main.cpp:408); next guess = pose_vec2mat(T[k-1]^-1 T[k]) (main.cpp:315-317,331) -- one native call for the B drives
        self.x0 = self.api.pose_handoff(self.P_prev, Ts)
        self.k = k
        if k == 1:
            self.first = xs.copy()
        return xs, Ts, Ss

    def prepare(self, K):
        """descriptors of the next K frames of every drive (and of their matches), laid out as velo_register_sequences takes them"""
        k0 = self.k
        refs, keep, _ = self.api.sequence_refs(self.frames, self.local_rank, first=k0 + 1, count=K)
        vis = self.api.sequence_visual_refs(self.vis, first=k0, count=K) if self.vis is not None else None
        return (k0, K, refs, keep, vis)

    def walk(self, prep, lockstep=False):
        """K frames of every drive in ONE library call (velo_register_sequences): the lock-step groups walk their drives independently, or
        (lockstep) start every frame together"""
        k0, K, refs, _keep, vis = prep
        assert k0 == self.k
        xs, Ts, Ss = self.api.register_sequences(self.ctxs, refs, K, self.P_prev, self.x0, visual=vis, lockstep=lockstep)
        if k0 == 0:
            self.first = xs[0].copy()
        self.k = k0 + K
        return xs, Ts, Ss


def run_leg(rig, a, workload, mode, B, steps, warmup, single_leg=True, comm="peer", drives=None):
    """One timed leg: `steps` steps of B registrations (B = 1 in the sharded modes) bracketed by barrier + synchronize, max over ranks.
    drives (replicas, c1 / c2 / c3): B planned drives with their frames -- every step registers the next frame of every drive."""
    import velo_amd  # noqa: F401
    from velo_amd import api
    torch = rig.torch
    world, rank = rig.world, rig.rank
    B = 1 if mode != "replicas" else max(1, B)
    drive = drives is not None and mode == "replicas" and workload in ("c1", "c2", "c3") and len(drives) >= B
    if drive:
        from velo_amd import synth
        n_frames = min(len(p["frames"]) for p in drives[:B])
        steps = max(1, min(steps, n_frames - 2 - warmup))      # (the last timed step still announces a frame: main.cpp:216 loads one per step)
        label, icp_skip = {
            "c1": ("configs[0] stand-in: synthetic 120k-pt drives in the KITTI ring layout, reference constants (icp_skip=200)", 200),
            "c2": ("synthetic HDL-64E 64x1875=120k-pt drives (configs[1]): frame k+1 -> frame k, icp_skip=1, point-to-plane ICP", 1),
            "c3": ("configs[2]: 120k-pt drives + 2000 stereo reprojection blocks per pair, icp_skip=1", 1)}[workload]
        label += f"; {B} drives advancing one frame per step, the previous frame promoted to target on the device, constant-velocity initial guess"
        vis_all = None
        if workload == "c3":
            vis_all = [[synth.stereo_matches(1000, seed=3 + 1000 * i + k, x_true=drives[i]["x_true"][k]) for k in range(n_frames - 1)] for i in range(B)]
        first_pairs = [dict(tgt_xyz=p["frames"][0][0], tgt_off=p["frames"][0][1], src_xyz=p["frames"][1][0], src_off=p["frames"][1][1],
                            x0=synth.INITIAL_GUESS.copy(), x_true=p["x_true"][0], vis=(vis_all[i][0] if vis_all else None)) for i, p in enumerate(drives[:B])]
        W = dict(label=label, icp_skip=icp_skip, pairs=first_pairs, shared_map=False, distinct=steps * B)
    else:
        W = make_workload(workload, B, a.same_pair or mode != "replicas")
    pairs, icp_skip, label = W["pairs"], W["icp_skip"], W["label"]
    d0 = pairs[0]
    tgt_off0, tgt_first_ring, tgt_first_point = d0["tgt_off"], 0, 0
    sharded_part = mode == "target-sharded" and world > 1
    if sharded_part:
        from velo_amd import shard
        r0, r1, p0, tgt_off0 = shard.target_ring_block(d0["tgt_off"], rank, world)
        tgt_first_ring, tgt_first_point = r0, p0
    # inputs resident in HBM before the timed region: one tensor per DIFFERENT cloud (contexts that register the same cloud share it)
    dev_of = {}

    def resident(arr):
        k = id(arr)
        if k not in dev_of:
            dev_of[k] = torch.from_numpy(np.ascontiguousarray(arr)).to(rig.dev)
        return dev_of[k]

    if sharded_part:
        tgts = [(resident(d0["tgt_xyz"][p0:p0 + int(tgt_off0[-1])]), tgt_off0)]
    else:
        tgts = [(resident(d["tgt_xyz"]), d["tgt_off"]) for d in pairs]
    srcs = [(resident(d["src_xyz"]), d["src_off"]) for d in pairs]
    keep = (lambda arr: np.ascontiguousarray(arr)) if getattr(a, "host_inputs", False) else resident
    frames_dev = [[(keep(f[0]), f[1]) for f in p["frames"]] for p in drives[:B]] if drive else None
    torch.cuda.synchronize()
    ctxs = [api.Context(rig.local_rank, icp_skip=icp_skip) for _ in range(B)]
    comm_info = None
    try:
        for c, d in zip(ctxs, pairs):
            c.set_timing(a.timing)
            if d["vis"] is not None:
                c.set_visual(d["vis"])
        if mode != "replicas" and world > 1:
            ok = comm == "peer"
            peer_errors = []                                 # why the peer slabs were given up, for the JSON line (not stderr only)
            if ok:
                # peer-mapped slabs.  Every rank must end up on the same path: each local step is followed by an all-gather of its
                # outcome, so a rank that cannot export or map a handle makes everybody fall back to RCCL together.
                def agreed(fn):
                    try:
                        val = fn()
                    except Exception as e:       # noqa: BLE001
                        print(f"[bench] rank {rank}: peer slabs unavailable ({e})", file=sys.stderr, flush=True)
                        peer_errors.append(f"rank {rank}: {type(e).__name__}: {str(e)[:120]}")
                        val = None
                    got = [None] * world
                    rig.dist.all_gather_object(got, val)
                    if not all(g is not None for g in got) and not peer_errors:
                        peer_errors.append("ranks " + ",".join(str(r) for r, g in enumerate(got) if g is None) + " could not export / map a peer slab")
                    return got if all(g is not None for g in got) else None
                c0 = ctxs[0]
                handles = agreed(c0.comm_peer_export)
                ok = handles is not None and agreed(lambda: (c0.comm_peer_attach(handles, rank, world), 1)[1]) is not None
                if ok and mode == "target-sharded":
                    nq_max = int(d0["src_xyz"].shape[0])
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
            if peer_errors:
                comm_info["peer_fallback_reason"] = peer_errors[0][:160]
        results = [None] * B

        def load_pair(i, k=None):
            k = i if k is None else k                        # context i takes pair k
            ctxs[i].set_target_part(tgts[k % len(tgts)][0], tgts[k % len(tgts)][1], tgt_first_ring, tgt_first_point)
            ctxs[i].set_source(srcs[k][0], srcs[k][1])

        def one_pair(i, k=None):
            load_pair(i, k)
            results[i] = ctxs[i].frame_to_frame(pairs[i if k is None else k]["x0"])

        pool = ThreadPoolExecutor(max_workers=B) if B > 1 else None
        # (B == 1, whole target: the same single library call with one job -- velo_register_batch routes it to the single-pair path)
        one_call = B == 1 and a.batch_api and not sharded_part and mode == "replicas"
        # scan-to-map as it is used (BASELINE configs[3]): B scans against ONE accumulated map -- the jobs name the same target with
        # VELO_SCAN_SHARED, the library uploads and indexes it once per step and the B contexts hold it by reference (one 110 MB map in HBM
        # instead of B); --own-map-copies gives every context a copy and an index build of its own (what rounds 1-4 timed)
        use_shared = bool(W["shared_map"]) and mode == "replicas" and B > 1 and a.batch_api and not getattr(a, "own_map_copies", False)
        batch_refs = (api.scan_refs(tgts, rig.local_rank, shared=use_shared), api.scan_refs(srcs, rig.local_rank)) if (B > 1 or one_call) else None
        x0s = np.stack([np.asarray(d["x0"], dtype=np.float64) for d in pairs])

        walker = DriveWalker(api, ctxs, frames_dev, rig.local_rank, vis_all) if drive else None

        def step():
            if walker is not None:
                xs, Ts, Ss = walker.step()
                for i in range(B):
                    results[i] = (xs[i], Ts[i], Ss[i])
            elif pool is None and one_call:
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

        seq = walker is not None and (getattr(a, "sequences", False) or getattr(a, "sequences_lockstep", False))                 # the drives' timed frames through ONE velo_register_sequences call
        if seq:
            if warmup > 0:
                walker.walk(walker.prepare(warmup))
            prep = walker.prepare(steps)                         # (descriptor arrays only: built off the clock)
        else:
            for _ in range(warmup):
                step()
        # The harness's own interpreter must not stall the timed region: with torch imported a full pass of Python's cyclic garbage
        # collector takes ~55 ms, and when its allocation counter happened to trip inside a 12-step leg the leg read 860 instead of 1,450
        # pairs/s (kernel trace: all four queues idle for 56 ms in the middle of the region).  Collect now, keep it off until the leg ends
        # (what timeit does); nothing the library does is affected.
        gc.collect()
        gc.disable()
        for c in ctxs:
            c.kernel_times(reset=True)                       # the per-kernel log starts with the timed region
        chain0 = [c.chain_stats() for c in ctxs]
        rig.barrier(ctxs)
        k_timed_first = walker.k if walker is not None else 0     # the drives' frame index the timed region starts from
        t0 = time.perf_counter()
        assoc_ms, assoc_n, alg_bytes, assoc_bytes, evals = 0.0, 0, 0, 0, 0
        kept = []                                            # the steps' summaries: added up behind the timed region (150 ctypes reads per step)
        xs_seq = None
        if seq:
            xs_seq, Ts_seq, kept = walker.walk(prep, lockstep=getattr(a, "sequences_lockstep", False))
            for i in range(B):
                results[i] = (xs_seq[-1][i], Ts_seq[-1][i], kept[-1][i])
        else:
            xs_steps = []
            for _ in range(steps):
                step()
                kept.append([r[2] for r in results])
                xs_steps.append([r[0] for r in results])
            if walker is not None:
                xs_seq = xs_steps
        rig.barrier(ctxs)
        dt = rig.max_over_ranks(time.perf_counter() - t0)
        kept_pick = {(i, f): kept[f][i] for f in range(len(kept)) for i in range(len(kept[f]))} if (walker is not None and not seq) else {}
        if os.environ.get("VELO_BENCH_DUMP_EVALS"):              # dev aid: evaluations per step, context and solve (what the chain's launch prediction has to guess)
            np.save(os.environ["VELO_BENCH_DUMP_EVALS"], np.array([[[s.solves[k].evaluations for k in range(s.n_solves)] for s in ss] for ss in kept], dtype=np.int32))
        for step_summaries in kept:
            for s in step_summaries:
                assoc_ms += s.assoc_kernel_ms
                assoc_n += s.assoc_kernel_launches
                alg_bytes += s.algorithmic_bytes
                assoc_bytes += s.assoc_bytes
                evals += sum(s.solves[k].evaluations for k in range(s.n_solves))
        # every timed registration against the drive's simulated motion (off the clock): a drive that diverged -- a bad registration feeding a
        # bad constant-velocity guess -- would change the iteration counts and still report a valid-looking rate
        truth = None
        if walker is not None and xs_seq is not None:
            k_first = k_timed_first
            et = max(float(np.linalg.norm(xs_seq[f][i][3:] - np.asarray(drives[i]["x_true"][k_first + f])[3:])) for f in range(steps) for i in range(B))
            er = max(float(np.linalg.norm(xs_seq[f][i][:3] - np.asarray(drives[i]["x_true"][k_first + f])[:3])) for f in range(steps) for i in range(B))
            truth = {"max_dt_m": et, "max_dw_rad": er, "pairs": steps * B,
                     "note": "largest difference between a timed pair's solved pose and the drive's SIMULATED relative pose: the registration's own noise "
                             "(range noise sigma = 0.02 m; 640 queries at icp_skip = 200), reported, not a pass/fail bound -- the check on the timed pairs is "
                             "timed_pairs_vs_oracle"}
        # TIMED pairs against the CPU oracle (off the clock): the frames a timed step registered -- loaded ahead behind the previous chain, the
        # target promoted by buffer rotation -- and the guess it started from, replayed through the oracle.  north_star: 1e-4 m / 1e-5 rad.
        vs_oracle = None
        if walker is not None and xs_seq is not None and not seq and not getattr(a, "no_oracle_check", False) and rank == 0:
            import oracle_lib
            k_first = k_timed_first
            picks = sorted({(min(3, B - 1), min(7, steps - 1)), (min(6, B - 1), min(15, steps - 1))})
            worst_t = worst_r = 0.0
            counts_equal, which = True, []
            for (i, f) in picks:
                kk = k_first + f + 1                             # the step that registered frame kk against frame kk - 1
                o = oracle_lib.Oracle(threads=oracle_lib.max_threads(), icp_skip=icp_skip)
                o.set_target(*drives[i]["frames"][kk - 1])
                o.set_source(*drives[i]["frames"][kk])
                if vis_all is not None:
                    o.set_visual(vis_all[i][kk - 1])
                xo, _To, so = o.frame_to_frame(walker.guess_log[kk][i])
                xg, sg = np.asarray(xs_seq[f][i]), kept_pick[(i, f)]
                worst_t = max(worst_t, float(np.linalg.norm(xo[3:] - xg[3:])))
                worst_r = max(worst_r, float(np.linalg.norm(xo[:3] - xg[:3])))
                counts_equal = counts_equal and [sg.solves[j].evaluations for j in range(sg.n_solves)] == [so.solves[j].evaluations for j in range(so.n_solves)]
                which.append(f"drive {i} timed step {f}")
            vs_oracle = {"dt_m": worst_t, "dw_rad": worst_r, "counts_equal": bool(counts_equal), "pairs": len(picks), "which": which,
                         "ok": bool(worst_t <= 1e-4 and worst_r <= 1e-5),
                         "note": "timed pairs replayed through the CPU oracle on the same two frames and the same guess; bound = north_star's 1e-4 m / 1e-5 rad"}
            if not vs_oracle["ok"]:
                raise RuntimeError(f"timed pairs differ from the oracle beyond the north_star tolerance: {vs_oracle}")
        del kept
        chain1 = [c.chain_stats() for c in ctxs]
        kacc = {}
        for c in ctxs:
            for name, (ms, n, b) in c.kernel_times(reset=True).items():
                e = kacc.setdefault(name, [0.0, 0, 0])
                e[0] += ms; e[1] += n; e[2] += b
        solutions = [[float(v) for v in r[0]] for r in results]
        first_pair_solutions = [[float(v) for v in x] for x in walker.first] if (walker is not None and walker.first is not None) else None

        if walker is not None and not seq and walker.k + 1 < walker.n_frames:
            # off the clock and behind the kernel log's read-out: the frame the last timed step announced and loaded ahead is registered by one more step (the drive's last frame:
            # nothing announced behind it), so that the contexts hold no frame loaded ahead when the single-pair walk restarts them
            walker.step()
        single = None
        if single_leg and world == 1 and walker is not None and B > 1 and walker.n_frames >= 8:
            # the single-pair latency next to the throughput -- ONE drive in flight (context 0 walks drive 0 again from its
            # first frame: a pair per call through the single-pair path, promotion and hand-off included), after the timed region
            w1 = DriveWalker(api, ctxs[:1], frames_dev[:1], rig.local_rank, vis_all[:1] if vis_all else None)
            n_warm = 3
            n1 = min(24, w1.n_frames - 2 - n_warm)
            for _ in range(n_warm):
                w1.step()
            ctxs[0].synchronize()
            ctxs[0].kernel_times(reset=True)
            t1 = time.perf_counter()
            a_ms, a_n = 0.0, 0
            for _ in range(n1):
                _xs, _Ts, Ss1 = w1.step()
                a_ms += Ss1[0].assoc_kernel_ms
                a_n += Ss1[0].assoc_kernel_launches
            ctxs[0].synchronize()
            lat = (time.perf_counter() - t1) / n1
            ktab1 = kernel_table(ctxs[0].kernel_times(reset=True))
            a_us = next((r["avg_launch_us"] for r in ktab1 if r["kernel"].startswith("assoc")), 1e3 * a_ms / max(a_n, 1))
            single = {"pairs_in_flight": 1, "pairs_walked": n1, "ms_per_pair": 1e3 * lat, "pairs_per_s": 1.0 / lat,
                      "assoc_avg_launch_us": a_us,
                      "kernels": [{k: r[k] for k in ("kernel", "share", "avg_launch_us", "frac")} for r in ktab1[:3]]}
        elif single_leg and world == 1 and mode == "replicas" and B > 1:
            # SURVEY 8(d) asks for the single-pair latency next to the throughput: ONE pair in flight, a sequence that walks through
            # the step's different pairs (a drive is a sequence, main.cpp:305-413), after the timed region
            for k in range(3):
                one_pair(0, k % B)
            ctxs[0].synchronize()
            ctxs[0].kernel_times(reset=True)
            t1 = time.perf_counter()
            n1, a_ms, a_n = 24, 0.0, 0
            for k in range(n1):
                one_pair(0, k % B)
                a_ms += results[0][2].assoc_kernel_ms
                a_n += results[0][2].assoc_kernel_launches
            ctxs[0].synchronize()
            lat = (time.perf_counter() - t1) / n1
            ktab1 = kernel_table(ctxs[0].kernel_times(reset=True))
            a_us = next((r["avg_launch_us"] for r in ktab1 if r["kernel"].startswith("assoc")), 1e3 * a_ms / max(a_n, 1))
            single = {"pairs_in_flight": 1, "pairs_walked": min(B, n1), "ms_per_pair": 1e3 * lat, "pairs_per_s": 1.0 / lat,
                      "assoc_avg_launch_us": a_us,
                      "kernels": [{k: r[k] for k in ("kernel", "share", "avg_launch_us", "frac")} for r in ktab1[:3]]}
            one_pair(0, 0)                                   # context 0 holds its own pair again

        shared = None
        if W["shared_map"] and mode == "replicas" and B > 1 and a.batch_api:
            # the other way of holding the map, next to the timed one (shared by default; see use_shared)
            refs_sh = (api.scan_refs([tgts[0]] * B, rig.local_rank, shared=not use_shared), batch_refs[1])
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
                      "pose_equal_to_timed_leg": bool(np.array_equal(xs_sh[0], results[0][0])),
                      "note": ("every context holds its OWN copy of the map: B uploads + index builds per step" if use_shared else
                               "the B jobs share one target (VELO_SCAN_SHARED): one upload + index build per step instead of B")}

        n_pairs_rank = steps * B
        total_pairs = n_pairs_rank * (world if mode == "replicas" else 1)
        s0 = results[0][2]
        per_pair_bytes = alg_bytes / max(n_pairs_rank, 1)
        ktab = kernel_table(kacc)
        if ktab:
            rf = dict(ktab[0])
        else:      # --timing 1: only the association launches were bracketed
            b_launch, avg_ms = assoc_bytes / max(assoc_n, 1), assoc_ms / max(assoc_n, 1)
            ach = (b_launch / 1e9) / (avg_ms / 1e3) if avg_ms > 0 else 0.0
            rf = {"kernel": "association search", "share": None, "launches": assoc_n, "avg_launch_us": avg_ms * 1e3, "algorithmic_bytes_per_launch": b_launch,
                  "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS}
        rf["traffic"] = None
        leg = {
            "workload": label + ("; inputs in HOST memory, uploaded every step (PCIe-inclusive)" if (drive and getattr(a, "host_inputs", False)) else ""),
            "mode": mode, "pairs_in_flight_per_gpu": B, "distinct_pairs": W["distinct"] if mode == "replicas" else 1,
            "steps": steps, "warmup": warmup,
            "pairs_per_s": total_pairs / dt, "ms_per_step": 1e3 * dt / steps,
            "Nq": int(s0.n_queries), "Nt": int(s0.n_target),
            "lm_evaluations_per_pair": evals / max(n_pairs_rank, 1),
            "association_launches_per_pair": assoc_n / max(n_pairs_rank, 1),
            "valid_correspondences_last_round": int(s0.solves[s0.n_solves - 1].n_icp_valid),
            "algorithmic_bytes_per_pair": per_pair_bytes,
            "achieved_hbm_GBs_whole_path": per_pair_bytes * (total_pairs / dt) / 1e9,
            "chain": {"calls": sum(b[0] - a_[0] for a_, b in zip(chain0, chain1)), "misses": sum(b[1] - a_[1] for a_, b in zip(chain0, chain1)),
                      "note": "calls enqueued as one chain of launches / calls whose predicted launch count was too short and were repeated host-driven"},
            "roofline": rf,
            "kernels": ktab[:3],
            "solution_x": solutions[0], "solutions": solutions,
        }
        if truth is not None:
            leg["against_simulated_motion"] = truth
        if vs_oracle is not None:
            leg["timed_pairs_vs_oracle"] = vs_oracle
        leg["call_shape"] = ("one velo_register_sequences call for the timed frames (groups walk their drives independently)" if seq else
                             "one library call per step") if walker is not None else "one library call per step"
        if first_pair_solutions is not None:                 # a drive's last pair is not its first: the pose the CPU baseline is compared with
            leg["first_pair_solution_x"] = first_pair_solutions[0]
            leg["frames_per_drive"] = walker.n_frames
        if single is not None:
            leg["single_pair"] = single
        if shared is not None:
            leg["own_map_copies" if use_shared else "shared_target"] = shared
        if W["shared_map"]:
            leg["map"] = "one map shared by the B contexts (VELO_SCAN_SHARED): one index build per step" if use_shared else "one copy and one index build per context and step"
        if comm_info is not None:
            leg["communicator"] = comm_info
        return leg
    finally:
        gc.enable()
        if rig.dist is not None:
            rig.dist.barrier()
        for c in ctxs:
            c.close()
        dev_of.clear()
        torch.cuda.empty_cache()


def self_launch(a):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves -- as a CHILD process (torch.distributed.run), before
    this process has imported torch or touched the GPU; relay its output and exit with its code.  Under a launcher (WORLD_SIZE set) the
    world size must be the one --gpus names: a mismatch is an error, never a silent one-GPU measurement."""
    ws = os.environ.get("WORLD_SIZE")
    if ws is not None:
        if int(ws) != a.gpus:
            sys.exit(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={ws} ranks")
        return
    if a.gpus <= 1:
        return
    import socket
    import subprocess
    with socket.socket() as s:                               # a free rendezvous port on the loop-back address
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    print(f"[bench] --gpus {a.gpus} without a launcher: starting {a.gpus} ranks ({' '.join(cmd[1:8])} ...)", file=sys.stderr, flush=True)
    sys.exit(subprocess.run(cmd, env=env).returncode)


def main():
    a = parse()
    self_launch(a)
    # the drives' frames are synthesised by worker processes BEFORE this process initialises the GPU (fork)
    drives = None
    if a.mode == "replicas" and not (a.same_pairs or a.same_pair) and a.workload in ("c1", "c2", "c3") and a.batch_api and not a.separate_loads \
            and not (a.workload == "c1" and os.environ.get("VELO_KITTI_ROOT")):
        import velo_amd  # noqa: F401
        if not a.no_legs and not os.environ.get("VELO_DRIVE_CACHE") and int(os.environ.get("WORLD_SIZE", "1")) == 1:
            import tempfile                                  # the legs run as child processes and read these frames back instead of synthesising them again
            os.environ["VELO_DRIVE_CACHE"] = tempfile.mkdtemp(prefix="velo_drives_")
            a._own_cache = os.environ["VELO_DRIVE_CACHE"]
        # frames per drive: 1 held at the start + one per warm-up and timed step + ONE MORE, so that the last timed step announces and loads its
        # next frame like every other (round 5 timed 19 frame loads for 20 steps)
        drives = make_drives(max(1, a.batch), a.warmup + a.steps + 2, a.gen_procs)
    rig = Rig(a)
    world, rank = rig.world, rig.rank
    main_leg = run_leg(rig, a, a.workload, a.mode, a.batch, a.steps, a.warmup, comm=a.comm, drives=drives)
    legs, modes = {}, {}
    if not a.no_legs:
        if world == 1 and a.mode == "replicas":
            # Short legs of the other single-GPU configs, EACH IN A PROCESS OF ITS OWN (`python bench.py --workload <leg> --no-legs`, started as a
            # child once this process has released its contexts and buffers): a leg is a complete bench -- warm-up, barrier-bracketed timed
            # region, single-pair latency -- of that config, measured the way the driver measures the headline.  (Run inside this process behind
            # the main leg, the host-bound legs read 15-25 % low -- c3 2.4-2.65 k against 3.1-3.25 k pairs/s alone, c1 6.9-7.9 k against 8.0-8.5 k --
            # with the same kernels, the same four evenly busy queues and more idle time between their operations; `--legs-in-process` keeps
            # that form for A/B.)  c1 / c3 walk the SAME drives as the headline (as many steps as their frames allow: the children read the
            # parent's frames back from VELO_DRIVE_CACHE), c4 registers B scans against the 2M-point map.
            import subprocess

            def child_leg(workload, steps, warmup, extra=()):
                import tempfile
                fd, dpath = tempfile.mkstemp(prefix=f"velo_leg_{workload}_", suffix=".json")
                os.close(fd)
                cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--workload", workload, "--steps", str(steps), "--warmup", str(warmup), "--batch", str(a.batch),
                       "--timing", str(a.timing), "--no-legs", "--no-cpu-baseline", "--detail-out", dpath, *extra]
                try:
                    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
                    rows = [ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")]
                    if out.returncode != 0 or not rows:
                        return {"error": (out.stderr or "no output")[-400:]}
                    ln = json.load(open(dpath))                  # the child's FULL record (its stdout line is the compact one)
                except Exception as e:       # noqa: BLE001  (a leg never takes the headline down)
                    return {"error": f"{type(e).__name__}: {str(e)[:300]}"}
                finally:
                    try:
                        os.unlink(dpath)
                    except OSError:
                        pass
                leg = {"workload": ln["config"]["workload"], "mode": ln["config"]["mode"], "pairs_in_flight_per_gpu": ln["config"]["pairs_in_flight_per_gpu"],
                       "distinct_pairs": ln["config"]["distinct_pairs"], "steps": ln["steps"], "warmup": ln["warmup"], "pairs_per_s": ln["value"], "ms_per_step": ln["ms_per_step"],
                       "Nq": ln["config"]["Nq"], "Nt": ln["config"]["Nt"], "lm_evaluations_per_pair": ln["config"]["lm_evaluations_per_pair"],
                       "algorithmic_bytes_per_pair": ln["config"]["algorithmic_bytes_per_pair"], "achieved_hbm_GBs_whole_path": ln["achieved_hbm_GBs_whole_path"],
                       "chain": ln["chain"], "roofline": ln["roofline"], "kernels": ln["kernels"], "solution_x": ln["solution_x"], "process": "a child process of its own"}
                for kk in ("single_pair", "against_simulated_motion", "timed_pairs_vs_oracle", "own_map_copies", "shared_target", "map", "first_pair_solution_x"):
                    if kk in ln:
                        leg[kk] = ln[kk]
                return leg

            in_proc = getattr(a, "legs_in_process", False)
            leg_cap = int(os.environ.get("VELO_BENCH_LEG_STEPS", "0"))         # tests: the legs' steps capped (the line's shape is what they check)
            for name, steps in (("c1", 100), ("c3", 40), ("c4", 12)):
                if leg_cap > 0:
                    steps = min(steps, leg_cap)
                if name != a.workload:
                    kitti_c1 = name == "c1" and os.environ.get("VELO_KITTI_ROOT")
                    # (warm-up as the headline's: a context's buffers are allocated over its first three steps, and an allocation stalls the queues
                    #  for 6-7 ms one step later -- tools/step_times.py)
                    w_leg = 3 if name == "c4" else max(a.warmup, 5)
                    if in_proc:
                        legs[name] = run_leg(rig, a, name, "replicas", a.batch, steps, w_leg, drives=None if kitti_c1 else drives)
                    else:
                        n_leg = steps if (drives is None or kitti_c1 or name == "c4") else max(1, min(steps, len(drives[0]["frames"]) - 2 - w_leg))
                        legs[name] = child_leg(name, n_leg, w_leg)
                    if not a.no_cpu_baseline and "error" not in legs[name]:
                        # every leg next to the CPU restatement on its own first pair, with the pose difference.  c1: the reference's own constants on
                        # its own kind of host (icp_skip = 200, one thread, velo.h:900, and all cores); c3: with the pair's stereo matches; c4: the
                        # whole scan-to-map call on all cores (the one-thread run would take minutes)
                        from velo_amd import synth
                        if name == "c4" or drives is None or kitti_c1:
                            dleg = make_workload(name, a.batch, a.same_pair)["pairs"][0]
                        else:
                            f0, f1 = drives[0]["frames"][0], drives[0]["frames"][1]
                            dleg = dict(tgt_xyz=f0[0], tgt_off=f0[1], src_xyz=f1[0], src_off=f1[1], x0=synth.INITIAL_GUESS.copy(),
                                        vis=synth.stereo_matches(1000, seed=3, x_true=drives[0]["x_true"][0]) if name == "c3" else None)
                        cbl = cpu_baseline(dleg, dleg.get("vis"), 200 if name == "c1" else 1, single_thread=name != "c4")
                        xol = np.array(cbl.pop("x"))
                        xgl = np.array(legs[name].get("first_pair_solution_x", legs[name]["solution_x"]))
                        cbl["pose_diff_vs_gpu"] = {"dt_m": float(np.linalg.norm(xol[3:] - xgl[3:])), "dw_rad": float(np.linalg.norm(xol[:3] - xgl[:3]))}
                        legs[name]["cpu_baseline"] = cbl
            if drives is not None and not getattr(a, "host_inputs", False):
                # main.cpp:216,349 load a scan per frame: the same drives with their frames in pageable HOST memory -- every step uploads its B frames
                # (the library announces the next frames to itself, velo_hint_next_source: their copies run under the current step's launches)
                if in_proc:
                    import copy
                    a_h = copy.copy(a)
                    a_h.host_inputs = True
                    hl = run_leg(rig, a_h, a.workload, "replicas", a.batch, a.steps, a.warmup, single_leg=False, drives=drives)
                    hl_sol = hl["solutions"]
                else:
                    hl = child_leg(a.workload, a.steps, a.warmup, ("--host-inputs",))
                    hl_sol = None
                if "error" in hl:
                    legs["host_inputs"] = hl
                else:
                    legs["host_inputs"] = {"pairs_per_s": hl["pairs_per_s"], "ms_per_step": hl["ms_per_step"], "steps": hl["steps"], "chain": hl["chain"],
                                           "of_resident_rate": hl["pairs_per_s"] / main_leg["pairs_per_s"],
                                           "solution_equal_to_resident": (hl_sol == main_leg["solutions"]) if hl_sol is not None else (hl["solution_x"] == main_leg["solution_x"]),
                                           "note": "frames in pageable host memory, 1.44 MB uploaded per pair inside the step (PCIe-inclusive); never `value`"}
            if a.workload == "c2" and not a.same_pair:
                # SURVEY 8(d) "Motion": the canonical pair from the start-up guess {0,0,0,0,0,1} (main.cpp:170) in every context -- the unit of work the
                # rounds before the drive workload measured (~40 LM evaluations per pair), kept so that rounds stay comparable
                if in_proc:
                    import copy
                    a_cp = copy.copy(a)
                    a_cp.same_pair = True
                    cp = run_leg(rig, a_cp, "c2", "replicas", a.batch, 10, 3, drives=None)
                else:
                    cp = child_leg("c2", min(10, leg_cap) if leg_cap > 0 else 10, 5, ("--same-pair",))
                legs["canonical_pair"] = {k: cp[k] for k in ("workload", "pairs_per_s", "ms_per_step", "lm_evaluations_per_pair", "algorithmic_bytes_per_pair",
                                                             "achieved_hbm_GBs_whole_path", "chain", "single_pair", "solution_x", "error") if k in cp}
                legs["canonical_pair"]["initial_guess"] = "start-up guess {0,0,0,0,0,1} (main.cpp:170); true motion: yaw 0.02 rad, t = (1.00, 0.02, 0.01) m (SURVEY 8d)"
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
        # the committed PMC passes: profiles/rNN_traffic.json (config 2), profiles/rNN_c4_traffic.json (config 4): HBM-side bytes per launch
        # of the kernels measured alone, keyed by kernel name ("traffic_by_kernel"); older files carry the association kernel only
        pat = {"c2": "r[0-9][0-9]_traffic.json", "c4": "r[0-9][0-9]_c4_traffic.json"}.get(a.workload)
        tf = sorted(glob.glob(os.path.join(ROOT, "profiles", pat))) if pat else []
        sq = None
        if tf:
            try:
                tj = json.load(open(tf[-1]))
                by = tj.get("traffic_by_kernel") or {}

                by_load = tj.get("traffic_by_kernel_in_flight") or {}

                def traffic_of(name):      # by the name of the instantiation that RAN, from the pass with the bench's own pairs in flight; else from
                    if name in by_load:    # the kernel-alone pass; never another instantiation's bytes
                        return by_load[name]
                    return by.get(name)
                rf["traffic"] = traffic_of(rf["kernel"])
                for kr in main_leg["kernels"]:
                    kr["traffic"] = traffic_of(kr["kernel"])
                sq = tj.get("sq_per_launch")
                rf["traffic_source"] = os.path.basename(tf[-1])
            except Exception:       # noqa: BLE001
                pass
        rf["note"] = ("the kernel with the largest share of the timed region's summed kernel time; launch durations from HIP events "
                      "(hipExtLaunchKernelGGL start/stop on the launching stream), accumulated per kernel name inside the library; a lock-step "
                      "group's launch serves its two contexts (algorithmic_bytes_per_launch says how much); with several pairs in flight a launch "
                      "shares the chip with the other groups' kernels and its start marker waits for the command processor, so the brackets read "
                      "higher than the kernel alone (single_pair.kernels); traffic = HBM-side bytes per launch of THIS kernel instantiation from the "
                      "committed PMC passes of this same command (profiles/*_traffic.json: FETCH_SIZE x 2 + WRITE_SIZE, separate passes, the bench's "
                      "own pairs in flight; rocprofv3 serialises the dispatches while it counts), or null when that instantiation was not measured")
        single = main_leg.get("single_pair")
        if sq and sq.get("SQ_INSTS_VALU"):
            t_alone = (single or {}).get("assoc_avg_launch_us", 0.0) * 1e-6
            rf["valu"] = {"kernel": "assoc_search_v5_kernel", "wave_insts_per_launch": sq["SQ_INSTS_VALU"], "salu": sq.get("SQ_INSTS_SALU"), "lds": sq.get("SQ_INSTS_LDS"),
                          "issue_slots_per_launch_at_2p4GHz": 1024 * 2.4e9 / 2 * t_alone,
                          "note": "association kernel, from the committed PMC pass (profiles/*_traffic.json); launch time = the kernel alone; a wave64 VALU "
                                  "instruction occupies a SIMD-32 for 2 cycles (MI355X_MICROARCH.md), packed-f32 and f64 ones for more"}
        line = {
            "metric": METRIC, "value": main_leg["pairs_per_s"], "unit": "scan-pairs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": main_leg["ms_per_step"], "higher_is_better": True,
            "scaling": "weak" if a.mode == "replicas" else "strong", "vs_baseline": None,
            "dtype": "f32 association / f64 residuals+solve",
            "data": "kitti" if (a.workload == "c1" and "KITTI seq 00" in main_leg["workload"]) else "synthetic",
            "config": {"workload": main_leg["workload"], "pairs_in_flight_per_gpu": main_leg["pairs_in_flight_per_gpu"],
                       "distinct_pairs": main_leg["distinct_pairs"], "frames_per_drive": main_leg.get("frames_per_drive"), "mode": a.mode,
                       "Nq": main_leg["Nq"], "Nt": main_leg["Nt"], "lm_evaluations_per_pair": main_leg["lm_evaluations_per_pair"],
                       "valid_correspondences_last_round": main_leg["valid_correspondences_last_round"],
                       "algorithmic_bytes_per_pair": main_leg["algorithmic_bytes_per_pair"], "call_shape": main_leg.get("call_shape"),
                       "initial_guess": "constant-velocity prediction from the drive's last two poses (main.cpp:311-331); start-up guess {0,0,0,0,0,1} for a drive's first pair (main.cpp:170)"},
            "achieved_hbm_GBs_whole_path": main_leg["achieved_hbm_GBs_whole_path"],
            "chain": main_leg["chain"],
            "roofline": rf,
            "kernels": main_leg["kernels"],
            "solution_x": main_leg["solution_x"],
        }
        if "first_pair_solution_x" in main_leg:
            line["first_pair_solution_x"] = main_leg["first_pair_solution_x"]
        if "communicator" in main_leg:
            line["config"]["communicator"] = main_leg["communicator"]
        if single is not None:
            line["single_pair"] = single
        for kk in ("against_simulated_motion", "timed_pairs_vs_oracle"):
            if kk in main_leg:
                line[kk] = main_leg[kk]
        for kk in ("shared_target", "own_map_copies", "map"):
            if kk in main_leg:
                line[kk] = main_leg[kk]
        if "host_inputs" in legs:
            line["host_inputs"] = legs.pop("host_inputs")
        if legs:
            line["configs"] = {k: {kk: vv for kk, vv in v.items() if kk not in ("solution_x", "solutions")} for k, v in legs.items()}
            # the 2M-point map leg carries its own committed PMC pass (profiles/rNN_c4_traffic.json)
            tf4 = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_c4_traffic.json")))
            if tf4 and "c4" in line["configs"] and isinstance(line["configs"]["c4"].get("roofline"), dict):
                try:
                    t4 = json.load(open(tf4[-1]))
                    r4 = line["configs"]["c4"]["roofline"]
                    by4 = dict(t4.get("traffic_by_kernel") or {})
                    by4.update(t4.get("traffic_by_kernel_in_flight") or {})
                    r4["traffic"] = by4.get(r4["kernel"])
                    r4["traffic_note"] = "HBM-side bytes per launch of this kernel instantiation (committed PMC passes of the c4 workload), or null"
                except Exception:       # noqa: BLE001
                    pass
        if modes:
            line["modes"] = {k: {kk: vv for kk, vv in v.items() if kk not in ("roofline", "kernels", "solutions")} for k, v in modes.items()}   # solution_x stays: tests compare it with a single-rank call
        if not a.no_cpu_baseline and world == 1:             # rank 0 at N = 1 only: the other runs just report the GPU side
            if drives is not None:                           # the first pair of drive 0 (start-up guess), with the matches the c3 walk gave it
                from velo_amd import synth
                f0, f1 = drives[0]["frames"][0], drives[0]["frames"][1]
                d = dict(tgt_xyz=f0[0], tgt_off=f0[1], src_xyz=f1[0], src_off=f1[1], x0=synth.INITIAL_GUESS.copy(),
                         vis=synth.stereo_matches(1000, seed=3, x_true=drives[0]["x_true"][0]) if a.workload == "c3" else None)
                skip = 200 if a.workload == "c1" else 1
            else:
                W = make_workload(a.workload, a.batch, a.same_pair)
                d, skip = W["pairs"][0], W["icp_skip"]
            cb = cpu_baseline(d, d["vis"], skip)
            xo = np.array(cb.pop("x"))
            xg = np.array(main_leg.get("first_pair_solution_x", main_leg["solution_x"]))
            cb["pose_diff_vs_gpu"] = {"dt_m": float(np.linalg.norm(xo[3:] - xg[3:])), "dw_rad": float(np.linalg.norm(xo[:3] - xg[:3]))}
            line["cpu_baseline"] = cb
        # the FULL record: a side file next to this script (bench_detail.json; --detail-out names another place) and stderr.  stdout carries
        # ONE compact line (<= LINE_LIMIT bytes) -- the driver's parser gave up on round 5's 20 KB line.
        dpath = a.detail_out or os.path.join(ROOT, "bench_detail.json")
        try:
            with open(dpath, "w") as f:
                json.dump(line, f)
        except OSError as e:
            print(f"[bench] detail file not written ({e})", file=sys.stderr, flush=True)
        print("[bench] full record: " + json.dumps(line), file=sys.stderr, flush=True)
        sys.stdout.flush()
        print(json.dumps(compact_line(line, os.path.basename(dpath))), flush=True)
    if getattr(a, "_own_cache", None):
        import shutil
        shutil.rmtree(a._own_cache, ignore_errors=True)
    if rig.dist is not None:
        rig.dist.barrier()
        rig.dist.destroy_process_group()


if __name__ == "__main__":
    main()
