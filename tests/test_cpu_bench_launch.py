"""bench.py's launcher logic on CPU: `python bench.py --gpus N` without a rank environment re-executes itself under
torch.distributed.run with N ranks on 127.0.0.1 (the driver's SCALE command has exactly that bare form)."""
import importlib.util
import os
import sys
import types

import pytest

from common import ROOT


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_self_launch_builds_the_torchrun_command(monkeypatch):
    b = _bench()
    seen = {}

    def fake_exec(exe, argv, env):
        seen.update(exe=exe, argv=argv, env=env)
        raise SystemExit(0)

    monkeypatch.setattr(b.os, "execve", fake_exec)
    monkeypatch.setattr(b.torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "5", "--warmup", "2"])
    with pytest.raises(SystemExit):
        b.self_launch(types.SimpleNamespace(gpus=8))
    a = seen["argv"]
    assert seen["exe"] == sys.executable and a[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in a and a[a.index("--nproc-per-node") + 1] == "8" and a[a.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(a[a.index("--master-port") + 1]) < 65536
    i = a.index(os.path.join(ROOT, "bench.py"))
    assert a[i + 1:] == ["--gpus", "8", "--steps", "5", "--warmup", "2"]  # the original flags travel unchanged
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def test_self_launch_refuses_when_the_gpus_are_not_there(monkeypatch):
    b = _bench()
    monkeypatch.setattr(b.torch.cuda, "device_count", lambda: 1)
    monkeypatch.delenv("PQT_BENCH_SAME_DEVICE", raising=False)
    with pytest.raises(SystemExit) as e:
        b.self_launch(types.SimpleNamespace(gpus=4))
    assert "only 1 GPU" in str(e.value)
