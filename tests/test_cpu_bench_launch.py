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


def _roof_inputs(b, shared, sr):
    w = b.WORKLOADS["synth100m"]
    qn, k = w["qn"], 100
    cand = 229_900_000
    W = {"w": w, "qn": qn, "shard": None, "n": w["n_base"], "name": "synth100m"}
    R = {"k": k, "st": {"candidates": cand, "bins_visited": 500 * qn, "queries": qn}, "stage": {"tables": 0.0, "traverse": 0.12, "gap": 0.01, "rerank_select": 2.15, "select": 0.43},
         "path": "traverse=fused-shape2 rerank=mode2-nw12-runs" + ("-shared" if shared else ""), "bv": 20000, "bb": 500, "n_timed": 5, "steps": 10, "sr": sr}
    ctx = types.SimpleNamespace(world=1)
    args = types.SimpleNamespace(option=[], no_live_traffic=True)
    return ctx, args, W, R, cand


def test_roofline_of_the_shared_row_pass_prices_deduplicated_bytes_and_stays_below_one():
    """VERDICT r05 #2, checked without a device: with the statistics the pass counts on the device (here: round 5's measured figures at 100 M,
    (20000, 500): 64.0 M distinct rows, 229.9 M distances, 2.15 ms) roofline.frac is the read-once fraction (0.54), SURVEY 8(d)'s bytes are a
    speed-up, the selection has its own fraction, the whole-path bytes per query shrink accordingly; without statistics the old pricing is kept
    and says so; a run without the pass is untouched."""
    b = _bench()
    sr = {"bins": 10937, "pairs": 16753, "distinct_rows": 63_963_423, "rows_read": 66_000_000, "items": 40000, "uncovered_queries": 12, "distances_written": 229_000_000,
          "capacity_flag": 0, "candidates": 229_900_000}
    ctx, args, W, R, cand = _roof_inputs(b, True, sr)
    roof, ex = b.roofline_block(ctx, args, W, R, live=False)
    once = sr["distinct_rows"] * 132 + 4 * sr["distances_written"]
    assert roof["kernel"] == "pqt_k_sr_adc" and roof["algorithmic_bytes_per_launch"] == once
    assert abs(roof["frac"] - once / 2.15e-3 / 1e9 / b.HBM_PEAK_GBS) < 1e-9 and 0.5 < roof["frac"] < 0.6
    dd = roof["deduplicated"]
    assert dd["survey_8d_bytes_per_launch"] == cand * 132 + W["qn"] * 800 and dd["algorithmic_equivalent_GBps"] > b.HBM_PEAK_GBS and 3.0 < dd["algorithmic_equivalent_speedup"] < 3.5
    assert 0 < roof["selection_kernel"]["frac"] < 1 and roof["selection_kernel"]["bytes_per_launch"] > 4 * sr["distances_written"]
    assert ex["path_bytes_q"] * W["qn"] < 0.4 * (cand * 132)  # the step's necessary bytes: a third of the per-candidate formula's
    assert "SURVEY 8(d)'s per-candidate bytes are kept as" in roof["accounting"]
    # no statistics (the extra call failed): the per-candidate pricing stays, flagged
    ctx, args, W, R, cand = _roof_inputs(b, True, {"error": "x"})
    roof, ex = b.roofline_block(ctx, args, W, R, live=False)
    assert "deduplicated" not in roof and roof["frac"] > 1 and "WITHOUT device statistics" in roof["accounting"]
    # the pass off: nothing changes
    ctx, args, W, R, cand = _roof_inputs(b, False, None)
    R["stage"]["rerank_select"] = 4.9
    roof, ex = b.roofline_block(ctx, args, W, R, live=False)
    assert roof["kernel"] == "pqt_k_rerank_select" and "deduplicated" not in roof and 0.7 < roof["frac"] < 0.85
    assert abs(ex["path_bytes_q"] - (512 + 4000 + 132 * cand / W["qn"] + 800)) < 1e-6
