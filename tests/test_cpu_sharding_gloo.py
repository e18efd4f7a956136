"""world_size-2 gloo test of the N>1 driver logic (product-quantization-tree_amd/sharding.py) on CPU.

The HIP engine is replaced by a stand-in that answers query_shard / merge_topk from the oracle, so this covers the
sharding arithmetic, buffer packing, the single all-gather and the (distance, position) merge protocol -- the parts
of the multi-GPU path that are host logic.  The kernels behind the same protocol are covered by the -m gpu test
test_sharded_two_way_equals_unsharded."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from common import ROOT, fixture, pqt_pkg


class OracleShardEngine:
    def __init__(self, fx, lo, hi):
        self.fx, self.lo, self.hi = fx, lo, hi

    def query_shard(self, q, bv, bb, k, out_idx, out_dist, out_pos, out_count):
        o = self.fx.oracle
        for qi in range(q.shape[0]):
            ids, d = o.query_unsorted(q[qi].numpy(), bv, bb)
            pos = np.arange(len(ids), dtype=np.uint32)
            m = (ids >= self.lo) & (ids < self.hi)
            ids, d, pos = ids[m], d[m], pos[m]
            order = np.lexsort((pos, d))[:k]
            n = len(order)
            out_idx[qi] = -1
            out_pos[qi] = -1
            out_dist[qi] = float("inf")
            out_idx[qi, :n] = torch.from_numpy(ids[order].astype(np.int64)).to(torch.int32)
            out_dist[qi, :n] = torch.from_numpy(d[order])
            out_pos[qi, :n] = torch.from_numpy(pos[order].astype(np.int64)).to(torch.int32)
            out_count[qi] = len(m)

    def merge_topk(self, world, qn, k, idx0, dist0, pos0, out_idx, out_dist, shard_stride):
        # idx0/dist0/pos0 are views of shard 0's [qn][k] block inside the gathered buffer; shard s sits shard_stride
        # 32-bit words further (the same addressing the HIP merge kernel uses)
        def blocks(t0):
            base = t0.untyped_storage()
            flat = torch.tensor([], dtype=t0.dtype).set_(base)
            o = t0.storage_offset()
            return torch.stack([flat[o + s * shard_stride:o + s * shard_stride + qn * k].view(qn, k) for s in range(world)])
        all_idx, all_dist, all_pos = blocks(idx0), blocks(dist0), blocks(pos0)
        for qi in range(qn):
            ids = all_idx[:, qi].reshape(-1).numpy()
            d = all_dist[:, qi].reshape(-1).numpy()
            pos = all_pos[:, qi].reshape(-1).numpy().view(np.uint32)
            order = np.lexsort((pos, d))[:k]
            out_idx[qi] = torch.from_numpy(ids[order])
            out_dist[qi] = torch.from_numpy(d[order])


def _worker(rank, world, port, q, exchange="alltoall"):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sh = __import__("importlib").import_module("product-quantization-tree_amd.sharding")
        fx = fixture("odd")
        n = fx.oracle.num_vectors
        lo, hi = sh.shard_range(rank, world, n)
        eng = OracleShardEngine(fx, lo, hi)
        queries = torch.from_numpy(fx.queries[:7])  # 7 % 2 != 0 and 7 % 3 != 0: the last query slice is padded
        k, bv, bb = 20, 300, 100
        buf = sh.ShardBuffers(world, queries.shape[0], k, "cpu")
        oi, od, cnt = sh.sharded_query(eng, dist, world, queries, bv, bb, k, buf, exchange=exchange)
        ok = True
        fx.oracle.set_sort_mode(1)
        for qi in range(queries.shape[0]):
            ids, d = fx.oracle.query(fx.queries[qi], bv, bb)
            kk = min(k, len(ids))
            ok &= bool(np.array_equal(oi[qi, :kk].numpy().view(np.uint32), ids[:kk]))
            ok &= bool(np.array_equal(od[qi, :kk].numpy().view(np.uint32), d[:kk].view(np.uint32)))
            ok &= int(cnt[qi]) == len(ids)
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def _worker_counts(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sh = __import__("importlib").import_module("product-quantization-tree_amd.sharding")
        # every rank derives the same global "bin of each vector" table from a seed, and keeps its own id range
        g = torch.Generator().manual_seed(1234)
        n = 5000
        bins_all = torch.randint(0, 2 ** 32, (300,), generator=g, dtype=torch.int64)[torch.randint(0, 300, (n,), generator=g)]
        lo, hi = sh.shard_range(rank, world, n)
        keys, counts, members = sh.local_bin_lists(bins_all[lo:hi].to(torch.int32), lo)
        uk, gs, low, ls = sh.global_bin_counts(dist, world, rank, keys, counts)
        # expectation from the unsharded table
        ek, ec = torch.unique(bins_all, return_counts=True)
        ok = bool(torch.equal(uk, ek) and torch.equal(gs, ec))
        below = torch.stack([(bins_all[:lo] == k).sum() for k in ek]) if lo else torch.zeros_like(ec)
        mine = torch.stack([(bins_all[lo:hi] == k).sum() for k in ek])
        ok &= bool(torch.equal(low, below) and torch.equal(ls, mine))
        # members: bin by bin in key order, ids ascending, all inside the shard's range
        off = 0
        for k, c in zip(keys.tolist(), counts.tolist()):
            seg = members[off:off + c]
            ok &= bool(torch.all(bins_all[seg] == k) and torch.all(seg[1:] > seg[:-1]) and seg.min() >= lo and seg.max() < hi)
            off += c
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_three_rank_gloo_build_time_bin_count_exchange():
    """Shard-by-shard build: per-bin global population / lower / local counts from ONE padded all-gather."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker_counts, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True), (2, True)]


def test_shard_ranges_partition():
    sh = __import__("importlib").import_module("product-quantization-tree_amd.sharding")
    for n in (0, 1, 7, 1000, 10 ** 9):
        for world in (1, 2, 3, 8):
            r = [sh.shard_range(i, world, n) for i in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))
            assert max(hi - lo for lo, hi in r) - min(hi - lo for lo, hi in r) <= 1


@pytest.mark.parametrize("world,exchange", [(2, "alltoall"), (3, "alltoall"), (2, "allgather")])
def test_gloo_sharded_query_equals_unsharded(world, exchange):
    """6 queries over 2 or 3 ranks (3: the query slices are padded, qn % world != 0 for the rows of the last slice): exchange
    by query slice (all-to-all + all-gather of the merged slices) and the single all-gather give the unsharded result."""
    fixture("odd")  # build once before forking
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + 7 * world + (3 if exchange == "allgather" else 0)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, exchange)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, True) for r in range(world)]
