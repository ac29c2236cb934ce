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
