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
    # the driver's command shape WITH the legs (their steps capped): round 5's line was only ever checked with --no-legs, grew to 20 KB with
    # them and the driver could not parse it (BENCH_r05.json: parsed null)
    detail = tmp_path / "detail.json"
    env = dict(os.environ, VELO_DRIVE_CACHE=str(tmp_path), VELO_BENCH_LEG_STEPS="3")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2", "--detail-out", str(detail)],
                         capture_output=True, text=True, timeout=850, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    all_lines = out.stdout.strip().splitlines()
    lines = [ln for ln in all_lines if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 prints ONE JSON line"
    assert all_lines[-1] == lines[0], "nothing behind the line"
    assert len(lines[0]) <= 8192, len(lines[0])
    line = json.loads(lines[0])
    # the legs' figures the line keeps, and the full record next to it
    for leg in ("c1", "c3", "c4"):
        got = line["configs"][leg]
        assert "error" not in got, got
        for key in ("pairs_per_s", "ms_per_step", "kernel", "frac", "traffic", "cpu_baseline", "pose_diff_vs_gpu"):
            assert key in got, (leg, key)
        assert got["pose_diff_vs_gpu"]["dt_m"] <= 1e-4 and got["pose_diff_vs_gpu"]["dw_rad"] <= 1e-5
    assert line["configs"]["c3"]["timed_pairs_vs_oracle"]["ok"] is True
    assert line["host_inputs"]["pairs_per_s"] > 0 and line["host_inputs"]["solution_equal_to_resident"] is True
    full = json.load(open(detail))
    assert full["value"] == pytest.approx(line["value"], rel=1e-5) and "kernels" in full["configs"]["c4"] and len(json.dumps(full)) > len(lines[0])
    # the timed pairs themselves -- frames loaded ahead behind the previous chain, targets promoted by rotation -- are the oracle's
    tp = line["timed_pairs_vs_oracle"]
    assert tp["ok"] is True and tp["counts_equal"] is True and tp["dt_m"] <= 1e-4 and tp["dw_rad"] <= 1e-5 and tp["pairs"] == 2
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert line["metric"] == base["metric"] and line["unit"] == "scan-pairs/s"
    assert line["n_gpus"] == 1 and line["steps"] == 4 and line["warmup"] == 2 and line["higher_is_better"] is True
    assert line["scaling"] == "weak" and line["vs_baseline"] is None and line["data"] == "synthetic" and "f64" in line["dtype"]
    assert line["value"] > 0 and abs(line["value"] - 8 * 1e3 / line["ms_per_step"]) < 1e-4 * line["value"]        # (the line's floats carry 6 significant digits)
    cfg = line["config"]
    assert "workload" in cfg and "model" not in cfg and "configs[1]" in cfg["workload"] and cfg["Nq"] == 120000 and cfg["Nt"] == 120000
    assert cfg["distinct_pairs"] == 4 * 8 and cfg["pairs_in_flight_per_gpu"] == 8           # every timed registration a pair of its own
    assert cfg["frames_per_drive"] == 2 + 4 + 2               # held + warm-up + timed + ONE MORE: the last timed step announces a frame too
    assert line["chain"]["calls"] == 4 * 8 and line["chain"]["misses"] <= line["chain"]["calls"]
    rf = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_us"):
        assert key in rf, key
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-5 * rf["frac"]      # (six significant digits in the line)
    assert rf["kernel"].endswith("_kernel") and 0 < rf["frac"] < 1
    # traffic: the PMC figure of THIS kernel instantiation from the committed passes, or null -- never another instantiation's
    import glob
    tj = json.load(open(sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_traffic.json")))[-1]))      # the file bench.py reads: the newest round's
    known = dict(tj.get("traffic_by_kernel") or {})
    known.update(tj.get("traffic_by_kernel_in_flight") or {})
    assert rf["traffic"] == pytest.approx(known.get(rf["kernel"]), rel=1e-5)            # (6 significant digits in the line)
    assert full["roofline"]["traffic"] == known.get(rf["kernel"])
    cb = line["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cb, key
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["unit"] == "scan-pairs/s" and cb["value"] > 0
    # the GPU's pose of drive 0's first pair is the oracle's (north_star tolerance; measured 1e-16)
    assert cb["pose_diff_vs_gpu"]["dt_m"] <= 1e-4 and cb["pose_diff_vs_gpu"]["dw_rad"] <= 1e-5
    assert np.all(np.isfinite(line["solution_x"]))
