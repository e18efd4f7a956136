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


def test_two_rank_range_sharded_bench_line():
    env = dict(os.environ, PQT_BENCH_BACKEND="gloo", PQT_BENCH_SAME_DEVICE="1", MASTER_ADDR="127.0.0.1")
    port = 29700 + os.getpid() % 200
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "synth1m", "--steps", "3", "--warmup", "1"]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    c = d["config"]
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and c["collective_world_size"] == 2
    assert c["ranks_agree"] is True
    assert c["same_workload_1gpu"]["results_identical_to_sharded"] is True
    assert "range-sharded" in c["parallelism"]
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
