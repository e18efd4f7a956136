"""bench.py's multi-GPU path on the one GPU the test box has: two ranks (gloo, both on device 0) run the range-sharded
layout end to end -- every rank builds only its own id range of a chunk-built cfg3-shape database, the per-bin counts come
from the build-time all-gather, every batch goes through pqt_query_shard -> one all-gather -> pqt_merge_topk -- and the
line the driver would parse must say: ranks agree, and the merged result equals the single-GPU result of the same database.
(RCCL itself needs one device per rank; the collective is the only thing gloo replaces here.)"""
import json
import os
import subprocess
import sys

import pytest

from common import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("traversal", ["sharded", "replicated"])
def test_two_rank_range_sharded_bench_line(traversal):
    env = dict(os.environ, PQT_BENCH_BACKEND="gloo", PQT_BENCH_SAME_DEVICE="1", MASTER_ADDR="127.0.0.1")
    port = 29700 + os.getpid() % 200 + (0 if traversal == "sharded" else 211)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "synth1m", "--steps", "3", "--warmup", "1",
           "--traversal", traversal]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    c = d["config"]
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and c["collective_world_size"] == 2
    assert c["ranks_agree"] is True
    assert c["same_workload_1gpu"]["results_identical_to_sharded"] is True
    assert "range-sharded" in c["parallelism"] and c["traversal"] == traversal
    assert ("bins-resolved" in c["kernel_path"]) == (traversal == "sharded"), c["kernel_path"]
    # the line carries the ratio of its own strong scaling at the top level (the two ranks share one GPU here: no speed-up expected)
    assert d["scaling_vs_1gpu"] == c["same_workload_1gpu"]["speedup_of_this_run"] > 0
    assert d["roofline"]["frac"] > 0 and c["mean_candidates"] > 100
    # the shards partition the candidates: this rank reranked about half of them
    assert 0.2 < c["mean_candidates_this_rank"] / c["mean_candidates"] < 0.8


@pytest.mark.parametrize("exchange", ["alltoall", "allgather"])
def test_single_rank_rccl_drives_every_collective_of_the_sharded_path(exchange):
    """PQT_BENCH_FORCE_SHARD=1: the range-sharded layout with ONE rank over the real RCCL backend -- broadcast of the tree and
    the queries, the build-time padded all-gather of bin counts, the per-batch all-to-all / all-gather, the MIN/MAX
    all-reduces of the ground truth and the timing -- an API / dtype check of the N > 1 code on the 1-GPU box."""
    env = dict(os.environ, PQT_BENCH_FORCE_SHARD="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29900 + os.getpid() % 90))
    env.pop("PQT_BENCH_BACKEND", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "synth1m", "--steps", "3", "--warmup", "1", "--exchange", exchange],
                         capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    c = d["config"]
    assert c["collective_backend"] == "rccl" and c["exchange"] == exchange and d["scaling"] == "strong"
    assert c["ranks_agree"] is True and c["same_workload_1gpu"]["results_identical_to_sharded"] is True


def test_live_traffic_of_the_dominant_kernel_is_collected_by_the_bench_run_itself():
    """`roofline.traffic` of the N = 1 line is measured by the run (two child runs of the same command under `rocprofv3 --pmc FETCH_SIZE` /
    `--pmc WRITE_SIZE`), not replayed from the committed profile: within a factor of the algorithmic bytes, source says so, and the timed value is
    untouched by it (collected after the timed region)."""
    env = dict(os.environ, PQT_BENCH_NO_PIPELINE="1")
    env.pop("PQT_BENCH_NO_LIVE_TRAFFIC", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "synth1m", "--steps", "4", "--warmup", "1", "--no-cpu", "--no-hbm-leg", "--pipeline", "1"],
                         capture_output=True, text=True, cwd=ROOT, timeout=1200, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["config"]["pipeline"].startswith("one batch at a time") and "one_batch_at_a_time" not in d["config"]  # --pipeline 1: the line IS that figure
    r = d["roofline"]
    assert r["traffic_source"].startswith("measured in this run"), r["traffic_source"]
    assert r["traffic"] > 0 and 0.05 < r["traffic_ratio"] < 20, r


def test_default_line_carries_the_hbm_roofline_leg():
    """`python bench.py` (N = 1): `value` is the SIFT1M-shape number and config.hbm_roofline_leg holds the configs[2] workload at both
    knob sets with the dominant kernel's roofline fraction (here with a small stand-in workload so the test stays short)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1", "--no-cpu", "--hbm-workload", "synth1m"],
                         capture_output=True, text=True, cwd=ROOT, timeout=900, env={k_: v_ for k_, v_ in os.environ.items() if k_ != "PQT_BENCH_NO_PIPELINE"})
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["scaling"] == "strong" and d["config"]["workload_name"] == "sift1m"
    # default: two whole batches in flight on the one device; the same steps one batch at a time ride along, results identical
    assert "two whole batches in flight" in d["config"]["pipeline"] and "two batches in flight" in d["config"]["kernel_path"]
    one = d["config"]["one_batch_at_a_time"]
    assert "error" not in one and one["results_identical"] is True and one["queries_per_sec"] > 0
    # both figures at the top level, each labelled; the view slot's outputs were compared with slot 0
    assert d["value_one_batch_at_a_time"] == one["queries_per_sec"] and d["ms_per_step_one_batch_at_a_time"] == one["ms_per_step"] and "two whole" in d["value_is"]
    assert d["config"]["view_slot_results_identical"] is True
    assert d["roofline"]["bound"] == "issue"  # the 64 MB line store of configs[1] is cache resident: `frac` is the contract's figure, not the bound
    # `roofline` prices the kernel alone on the device (the one-batch-at-a-time steps); the overlapped launches of `value` ride along
    fl = d["roofline"]["in_flight"]
    assert fl["kernel"] == d["roofline"]["kernel"] and fl["avg_launch_ms"] > 0 and d["roofline"]["avg_launch_ms"] > 0 and len(fl["all_kernels"]) == 2
    assert d["roofline"]["avg_launch_ms"] in one["stage_ms"].values()
    assert d["roofline"]["traffic_source"] is None or "committed profile" in d["roofline"]["traffic_source"]
    assert d["roofline"]["measured_stream_GBps"] > 1000
    leg = d["config"]["hbm_roofline_leg"]
    assert "error" not in leg, leg
    for knobs in ("knobs_20000_500", "knobs_4096_4096"):
        e = leg[knobs]
        assert e["queries_per_sec"] > 0 and 0 < e["roofline"]["frac"] < 1 and e["roofline"]["kernel"].startswith("pqt_k_")
        assert e["batches"].startswith("a fresh batch") and e["same_batch_every_step"]["queries_per_sec"] > 0  # the leg's figures: a fresh batch every step
        assert "rerank=mode2" in e["kernel_path"] and e["filter_fallbacks"] == 0


def test_eight_gpu_default_code_path_with_two_ranks_and_small_stand_ins():
    """`--gpus 8` (no --workload): `value` is the big configuration and the sweep's workload rides beside it as config.strong_scaling_leg
    with its own one-GPU denominator.  The same code path with two gloo ranks on one device and small stand-in workloads."""
    env = dict(os.environ, PQT_BENCH_BACKEND="gloo", PQT_BENCH_SAME_DEVICE="1", MASTER_ADDR="127.0.0.1", PQT_BENCH_HEAD_WL="synth10m", PQT_BENCH_SWEEP_WL="synth1m")
    port = 30300 + os.getpid() % 200
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    c = d["config"]
    assert c["workload_name"] == "synth10m" and d["scaling"] == "strong" and c["ranks_agree"] is True
    leg = c["strong_scaling_leg"]
    assert "N=1000000" in leg["workload"] and leg["ranks_agree"] is True
    assert leg["same_workload_1gpu"]["results_identical_to_sharded"] is True
    assert d["scaling_vs_1gpu"] == leg["same_workload_1gpu"]["speedup_of_this_run"] > 0
    assert "strong_scaling_leg" in d["scaling_vs_1gpu_what"]


@pytest.mark.parametrize("pipeline", [2, 1, 3])
def test_bare_command_launches_its_own_ranks_and_reports_the_exchange(pipeline):
    """`python bench.py --gpus 2` with NO rank environment (the driver's own command form): bench.py re-executes itself under
    torch.distributed.run, rank 0 prints the one line, and the line carries the per-collective device times (exchange_ms), every
    rank's stage times, and the step time of the other schedule (two batches in flight vs one at a time) with identical results."""
    env = dict(os.environ, PQT_BENCH_BACKEND="gloo", PQT_BENCH_SAME_DEVICE="1")
    for k_ in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        env.pop(k_, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "synth1m", "--steps", "4", "--warmup", "1", "--timing-period", "2",
                          "--pipeline", str(pipeline)], capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    c = d["config"]
    assert d["n_gpus"] == 2 and c["collective_world_size"] == 2 and d["scaling"] == "strong"
    assert c["ranks_agree"] is True and c["same_workload_1gpu"]["results_identical_to_sharded"] is True
    ex = c["exchange_ms"]
    assert set(ex) == {"bins_allgather", "topk_alltoall", "merged_allgather", "calls_timed"}, ex
    assert all(ex[n_] > 0 for n_ in ("bins_allgather", "topk_alltoall", "merged_allgather"))
    assert ex["calls_timed"] == (4 if pipeline == 3 else 2)  # 2 of the 4 timed steps carry events; two half-batch calls each with --pipeline 3
    pr = c["per_rank_stage_ms"]
    assert [r_["rank"] for r_ in pr] == [0, 1] and all(r_["stage_ms"]["rerank_select"] > 0 and r_["candidates"] > 0 for r_ in pr)
    tag = {1: "one batch at a time", 2: "two whole batches in flight", 3: "two half batches in flight"}[pipeline]
    assert tag in c["pipeline"] and (pipeline == 1 or tag.replace("whole ", "") in c["kernel_path"])
    ab = c["pipeline_ab"]
    assert "error" not in ab, ab
    assert ab["results_identical"] is True and ab["other"] == "pipeline=%d" % (2 if pipeline == 1 else 1) and ab["other_ms_per_step"] > 0


def test_bare_command_with_eight_ranks_on_one_device():
    """BASELINE configs[3]'s world size through the driver's own command form: `python bench.py --gpus 8` with no rank
    environment, eight gloo ranks on the one device, a small stand-in database: 2000 queries = 250 per rank, every collective with eight
    participants, the line carries the three exchange times, every rank's stage times and the one-GPU denominator."""
    if not os.environ.get("PQT_TEST_EIGHT_RANKS"):
        # written while the GPU pool was closed for this repository: not yet run once on a GPU box, so it does not gate the suite.
        # PQT_TEST_EIGHT_RANKS=1 python -m pytest tests/test_gpu_bench_sharded.py -k eight_ranks   (scripts/r05_final.sh does)
        pytest.skip("set PQT_TEST_EIGHT_RANKS=1 (eight processes on one device: ~5 minutes)")
    env = dict(os.environ, PQT_BENCH_BACKEND="gloo", PQT_BENCH_SAME_DEVICE="1")
    for k_ in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        env.pop(k_, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--workload", "synth1m", "--steps", "4", "--warmup", "1", "--timing-period", "2"],
                         capture_output=True, text=True, env=env, cwd=ROOT, timeout=1500)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    c = d["config"]
    assert d["n_gpus"] == 8 and c["collective_world_size"] == 8 and d["scaling"] == "strong"
    assert c["ranks_agree"] is True and c["same_workload_1gpu"]["results_identical_to_sharded"] is True
    ex = c["exchange_ms"]
    assert set(ex) == {"bins_allgather", "topk_alltoall", "merged_allgather", "calls_timed"}, ex
    assert [r_["rank"] for r_ in c["per_rank_stage_ms"]] == list(range(8))
    assert d["scaling_vs_1gpu"] > 0 and c["planned_build_s_per_rank"] > 0
