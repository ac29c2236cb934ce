"""-m gpu: the driver's contract for bench.py at N = 1 -- ONE JSON line with the metric BASELINE.json names, the roofline object of the
dominant kernel and the CPU baseline -- checked on a short run of the default (drive) workload; the poses behind the line are the oracle's."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(900)
def test_single_gpu_bench_line_meets_the_contract(hip_lib, tmp_path):
    env = dict(os.environ, VELO_DRIVE_CACHE=str(tmp_path))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2", "--no-legs"],
                         capture_output=True, text=True, timeout=850, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 prints ONE JSON line"
    line = json.loads(lines[0])
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert line["metric"] == base["metric"] and line["unit"] == "scan-pairs/s"
    assert line["n_gpus"] == 1 and line["steps"] == 4 and line["warmup"] == 2 and line["higher_is_better"] is True
    assert line["scaling"] == "weak" and line["vs_baseline"] is None and line["data"] == "synthetic" and "f64" in line["dtype"]
    assert line["value"] > 0 and abs(line["value"] - 8 * 1e3 / line["ms_per_step"]) < 1e-6 * line["value"]
    cfg = line["config"]
    assert "workload" in cfg and "model" not in cfg and "configs[1]" in cfg["workload"] and cfg["Nq"] == 120000 and cfg["Nt"] == 120000
    assert cfg["distinct_pairs"] == 4 * 8 and cfg["pairs_in_flight_per_gpu"] == 8           # every timed registration a pair of its own
    assert line["chain"]["calls"] == 4 * 8 and line["chain"]["misses"] <= line["chain"]["calls"]
    rf = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_us"):
        assert key in rf, key
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert rf["kernel"].endswith("_kernel") and 0 < rf["frac"] < 1
    # traffic: the PMC figure of THIS kernel instantiation from the committed passes, or null -- never another instantiation's
    import glob
    tj = json.load(open(sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_traffic.json")))[-1]))      # the file bench.py reads: the newest round's
    known = dict(tj.get("traffic_by_kernel") or {})
    known.update(tj.get("traffic_by_kernel_in_flight") or {})
    assert rf["traffic"] == known.get(rf["kernel"])
    cb = line["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cb, key
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["unit"] == "scan-pairs/s" and cb["value"] > 0
    # the GPU's pose of drive 0's first pair is the oracle's (north_star tolerance; measured 1e-16)
    assert cb["pose_diff_vs_gpu"]["dt_m"] <= 1e-4 and cb["pose_diff_vs_gpu"]["dw_rad"] <= 1e-5
    assert np.all(np.isfinite(line["solution_x"]))
