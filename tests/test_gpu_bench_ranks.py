"""-m gpu: bench.py as the driver launches it for N > 1 (one process per rank through torch.distributed.run), here with two
ranks forced onto the one GPU of the box and gloo for the barrier: the JSON line must carry the replicas headline AND the
north_star's two multi-GPU modes, each with the communicator kind and the ranks read back from it."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_bench_line_reports_both_sharded_modes(hip_lib):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
           "--dist-backend", "gloo", "--force-device", "0", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    text = out.stdout.strip().splitlines()[-1]
    assert len(text) <= 8192, len(text)                      # the N > 1 line adds `modes`: still small enough for the driver's parser
    assert [ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")] == [text]
    line = json.loads(text)
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    for mode in ("sharded", "target_sharded"):
        m = line["modes"][mode]
        assert "error" not in m or m["error"] is None, m
        assert m["communicator"]["ranks"] == 2 and m["communicator"]["kind"] in ("peer slabs (hipIpc)", "rccl")
        assert m["pairs_per_s"] > 0
    # what the sharded ranks computed is what ONE rank computes: the 120k-point pair split by queries, and the 120k scan against the
    # 2M-point map split by target rings (BASELINE configs[4]'s shape with two ranks; every rank searched its ring block of the full map)
    import numpy as np
    from velo_amd import api, synth
    for mode, d in (("sharded", synth.scan_pair()), ("target_sharded", synth.scan_to_map(2_000_000))):
        c = api.Context(0, icp_skip=1)
        c.set_target(d["tgt_xyz"], d["tgt_off"]); c.set_source(d["src_xyz"], d["src_off"])
        x, T, s = c.frame_to_frame(d["x0"])
        c.close()
        got = np.array(line["modes"][mode]["solution_x"])
        assert np.abs(got - x).max() <= 1e-9, (mode, got, x)


def test_bench_starts_its_own_ranks_when_no_launcher_wraps_it(hip_lib):
    """`python bench.py --gpus 2` WITHOUT torch.distributed.run: the script starts the two ranks itself (a plain --gpus N run must
    never measure one GPU and print n_gpus: 1)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--force-device", "0", "--dist-backend", "gloo",
           "--steps", "4", "--warmup", "1", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    text = out.stdout.strip().splitlines()[-1]
    assert len(text) <= 8192, len(text)
    line = json.loads(text)
    assert line["n_gpus"] == 2 and line["value"] > 0
    for mode in ("sharded", "target_sharded"):
        assert line["modes"][mode]["communicator"]["ranks"] == 2, line["modes"][mode]


@pytest.mark.timeout(900)
def test_eight_rank_bench_line_is_one_small_line(hip_lib):
    """The driver's 8-GPU command shape -- `python bench.py --gpus 8 --steps K --warmup W`, the script starting its own ranks -- with all eight
    ranks forced onto the one GPU of the box: the line adds `modes` for N > 1 and must still be ONE line of at most 8 KB that parses
    (BENCH_r05.json: parsed null was a line that had outgrown the driver), with both modes' communicators reporting eight ranks."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--force-device", "0", "--dist-backend", "gloo",
           "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-oracle-check"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=850, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = [ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(rows) == 1 and out.stdout.strip().splitlines()[-1] == rows[0]
    assert len(rows[0]) <= 8192, len(rows[0])
    line = json.loads(rows[0])
    assert line["n_gpus"] == 8 and line["scaling"] == "weak" and line["value"] > 0 and line["config"]["pairs_in_flight_per_gpu"] == 8
    for key in ("metric", "unit", "steps", "warmup", "ms_per_step", "roofline", "dtype", "data"):
        assert key in line, key
    for mode in ("sharded", "target_sharded"):
        m = line["modes"][mode]
        assert not m.get("error"), m
        assert m["communicator"]["ranks"] == 8 and m["pairs_per_s"] > 0
