"""CPU: bench.py's --gpus contract.  A plain `python bench.py --gpus N` starts N ranks itself (child process, before torch is
imported); under a launcher the world size must equal --gpus."""
import os
import subprocess
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    import importlib
    sys.path.insert(0, ROOT)
    return importlib.import_module("bench")


def test_world_size_must_match_gpus(monkeypatch):
    b = _bench()
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(SystemExit) as e:
        b.self_launch(types.SimpleNamespace(gpus=8))
    assert "WORLD_SIZE=2" in str(e.value)
    b.self_launch(types.SimpleNamespace(gpus=2))           # matching: falls through, this process is a rank


def test_single_gpu_run_is_not_relaunched(monkeypatch):
    b = _bench()
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    b.self_launch(types.SimpleNamespace(gpus=1))


def test_plain_gpus_n_starts_n_ranks(monkeypatch):
    b = _bench()
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return types.SimpleNamespace(returncode=7)
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    with pytest.raises(SystemExit) as e:
        b.self_launch(types.SimpleNamespace(gpus=4))
    assert e.value.code == 7                               # the child's return code is ours
    c = seen["cmd"]
    assert c[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in c and "127.0.0.1" in c
    assert c[-4:] == ["--gpus", "4", "--steps", "3"] and os.path.samefile(c[-5], os.path.join(ROOT, "bench.py"))
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" or "HSA_ENABLE_IPC_MODE_LEGACY" in os.environ


def test_bench_does_not_import_torch_before_launching():
    code = "import sys; sys.argv=['bench.py']; import bench; assert 'torch' not in sys.modules, 'torch imported at module load'"
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-500:]


def _fat_leg(name, with_cpu=True):
    """a leg as run_leg / child_leg return it, with everything that made round 5's line 20 KB: notes, kernel tables, poses of every context"""
    rf = {"kernel": f"{name}_dominant_kernel", "share": 0.5612345678, "launches": 3100, "avg_launch_us": 27.123456789, "algorithmic_bytes_per_launch": 5109440.0,
          "bound": "hbm", "achieved": 186.123456, "peak": 8000.0, "unit": "GB/s", "frac": 0.0233, "traffic": 8137730.0, "note": "n" * 1100}
    leg = {"workload": "w" * 300, "mode": "replicas", "pairs_in_flight_per_gpu": 8, "distinct_pairs": 160, "steps": 20, "warmup": 5, "pairs_per_s": 4151.123456789,
           "ms_per_step": 1.9268123456, "Nq": 120000, "Nt": 120000, "lm_evaluations_per_pair": 25.75, "algorithmic_bytes_per_pair": 136435000.0,
           "achieved_hbm_GBs_whole_path": 566.4, "chain": {"calls": 160, "misses": 0, "note": "c" * 200}, "roofline": rf,
           "kernels": [dict(rf, kernel=f"k{i}_kernel") for i in range(3)], "solution_x": [0.1234567890123] * 6, "solutions": [[0.1234567890123] * 6] * 8,
           "single_pair": {"pairs_in_flight": 1, "pairs_walked": 22, "ms_per_pair": 0.875, "pairs_per_s": 1142.0, "assoc_avg_launch_us": 81.0,
                           "kernels": [dict(rf, kernel=f"s{i}_kernel") for i in range(3)]},
           "against_simulated_motion": {"max_dt_m": 0.0494, "max_dw_rad": 0.0015, "pairs": 160, "note": "a" * 300},
           "timed_pairs_vs_oracle": {"dt_m": 1e-16, "dw_rad": 1e-18, "counts_equal": True, "pairs": 2, "which": ["drive 3 timed step 7", "drive 6 timed step 15"], "ok": True, "note": "t" * 200},
           "first_pair_solution_x": [0.1] * 6, "process": "a child process of its own"}
    if with_cpu:
        leg["cpu_baseline"] = {"value": 0.65, "unit": "scan-pairs/s", "cores": 128, "kind": "port", "sample": "s" * 400, "single_thread_pairs_per_s": 0.077,
                               "pose_diff_vs_gpu": {"dt_m": 1.5e-16, "dw_rad": 1.8e-18}}
    return leg


def _fat_full_line(n_gpus):
    import json
    main = _fat_leg("main")
    line = {"metric": json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"], "value": main["pairs_per_s"], "unit": "scan-pairs/s", "n_gpus": n_gpus,
            "steps": 20, "warmup": 5, "ms_per_step": main["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 association / f64 residuals+solve", "data": "synthetic",
            "config": {"workload": main["workload"], "pairs_in_flight_per_gpu": 8, "distinct_pairs": 160, "frames_per_drive": 27, "mode": "replicas", "Nq": 120000, "Nt": 120000,
                       "lm_evaluations_per_pair": 25.7, "valid_correspondences_last_round": 90000, "algorithmic_bytes_per_pair": 1.36e8, "call_shape": "x" * 100, "initial_guess": "g" * 200},
            "achieved_hbm_GBs_whole_path": 566.4, "chain": main["chain"], "roofline": dict(main["roofline"], valu={"note": "v" * 400}), "kernels": main["kernels"],
            "solution_x": main["solution_x"], "first_pair_solution_x": [0.1] * 6, "single_pair": main["single_pair"],
            "against_simulated_motion": main["against_simulated_motion"], "timed_pairs_vs_oracle": main["timed_pairs_vs_oracle"]}
    if n_gpus == 1:
        line["host_inputs"] = {"pairs_per_s": 3857.8, "ms_per_step": 2.07, "steps": 20, "chain": main["chain"], "of_resident_rate": 0.93, "solution_equal_to_resident": True, "note": "h" * 200}
        line["configs"] = {k: _fat_leg(k) for k in ("c1", "c3", "c4", "canonical_pair")}
        line["cpu_baseline"] = main["cpu_baseline"]
    else:
        m = _fat_leg("mode", with_cpu=False)
        m["communicator"] = {"kind": "peer slabs (hipIpc)", "ranks": n_gpus}
        line["modes"] = {"sharded": dict(m), "target_sharded": dict(m, error="rccl: " + "e" * 600, first_attempt={"error": "peer: " + "f" * 600})}
    return line


@pytest.mark.parametrize("n_gpus", [1, 8])
def test_the_printed_line_stays_small_enough_for_the_driver_to_parse(n_gpus):
    """BENCH_r05.json: parsed null -- the 20 KB line was unreadable.  Whatever the legs carry, the line rank 0 prints is <= 8 KB, keeps the
    contract's keys, the roofline and (N = 1) the CPU baseline, and the per-leg figures the review named."""
    import json
    b = _bench()
    full = _fat_full_line(n_gpus)
    assert len(json.dumps(full)) > 15000                     # the canned record is as fat as round 5's
    line = b.compact_line(full, "bench_detail.json")
    text = json.dumps(line)
    assert len(text) <= 8192 and len(text) <= b.LINE_LIMIT, len(text)
    assert json.loads(text) == line
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in line, key
    assert line["n_gpus"] == n_gpus and "workload" in line["config"] and "model" not in line["config"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_us"):
        assert key in line["roofline"], key
    assert len(line["roofline"]["note"]) < 200
    assert line["solution_x"] == full["solution_x"]          # full precision (tests hold poses to single-rank calls)
    assert line["timed_pairs_vs_oracle"]["ok"] is True and "note" not in line["timed_pairs_vs_oracle"]
    if n_gpus == 1:
        for key in ("value", "unit", "cores", "kind", "sample"):
            assert key in line["cpu_baseline"], key
        for leg in ("c1", "c3", "c4"):
            got = line["configs"][leg]
            for key in ("pairs_per_s", "ms_per_step", "kernel", "frac", "traffic", "cpu_baseline", "pose_diff_vs_gpu", "single_pair_ms"):
                assert key in got, (leg, key)
            assert "kernels" not in got and "roofline" not in got and "workload" not in got
    else:
        for mode in ("sharded", "target_sharded"):
            m = line["modes"][mode]
            assert m["communicator"]["ranks"] == 8 and m["pairs_per_s"] > 0 and m["solution_x"] == full["modes"][mode]["solution_x"]
        assert len(line["modes"]["target_sharded"]["error"]) <= 200 and "first_attempt_error" in line["modes"]["target_sharded"]


def test_the_line_guard_drops_optional_blocks_before_it_ever_exceeds_the_limit():
    import json
    b = _bench()
    full = _fat_full_line(1)
    full["configs"] = {f"leg{i}": _fat_leg(f"leg{i}") for i in range(40)}      # a future round adds legs without looking
    line = b.compact_line(full, "bench_detail.json")
    assert len(json.dumps(line)) <= b.LINE_LIMIT
    assert line["value"] == pytest.approx(full["value"], rel=1e-5) and "roofline" in line and "cpu_baseline" in line
