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

    # ---- query-sharded traversal (include/pqt_hip.h: pqt_traverse_bins / pqt_query_shard_bins) restated over the oracle ----
    def _bin_table(self):
        if not hasattr(self, "_bins"):
            fx = self.fx
            starts = np.concatenate([[0], np.cumsum(fx.bin_sizes.astype(np.int64))])
            self._bins = {int(b): fx.members[starts[i]:starts[i + 1]] for i, b in enumerate(fx.bin_ids)}
        return self._bins

    def traverse_bins(self, q, bv, bb, cap, out_bins):
        """Row i: the included populated bins of query i in visiting order as bin id | global start << 32, trailer = count | nCand << 32
        (count 0xffffffff when the list does not fit cap)."""
        o, tab = self.fx.oracle, self._bin_table()
        for qi in range(q.shape[0]):
            ids, _, seq = o.stage_bins(q[qi].numpy(), bb)
            run, ent = 0, []
            for b in ids[seq].tolist():  # visiting order; the reference's cut: a bin is taken while the count before it is <= Bv
                if run > bv:
                    break
                size = len(tab.get(b, ()))
                if size:
                    ent.append(b | (run << 32))
                run += size
            row = np.zeros(cap + 1, np.uint64)
            if len(ent) <= cap:
                row[:len(ent)] = np.array(ent, np.uint64)
                row[cap] = len(ent) | (run << 32)
            else:
                row[cap] = 0xffffffff
            out_bins[qi] = torch.from_numpy(row.view(np.int64))

    def query_shard_bins(self, q, bv, bb, k, bins, cap, out_idx, out_dist, out_pos, out_count):
        o, tab = self.fx.oracle, self._bin_table()
        for qi in range(q.shape[0]):
            row = bins[qi].numpy().view(np.uint64)
            m, ncand = int(row[cap]) & 0xffffffff, int(row[cap]) >> 32
            if m == 0xffffffff:  # the sender's list overflowed: this shard traverses the query itself
                self.query_shard(q[qi:qi + 1], bv, bb, k, out_idx[qi:qi + 1], out_dist[qi:qi + 1], out_pos[qi:qi + 1], out_count[qi:qi + 1])
                continue
            u_ids, u_d = o.query_unsorted(q[qi].numpy(), bv, bb)  # distance of the candidate at global visiting position p = u_d[p]
            assert ncand == len(u_ids)
            ids, d, pos = [], [], []
            for e in row[:m].tolist():
                b, g0 = int(e) & 0xffffffff, int(e) >> 32
                mem = tab[b]
                assert np.array_equal(u_ids[g0:g0 + len(mem)], mem)  # the listed start really is the bin's place in the visiting order
                sel = (mem >= self.lo) & (mem < self.hi)
                ids.append(mem[sel]); d.append(u_d[g0:g0 + len(mem)][sel]); pos.append(np.arange(g0, g0 + len(mem), dtype=np.uint32)[sel])
            ids, d, pos = (np.concatenate(a) if a else np.empty(0, t) for a, t in ((ids, np.uint32), (d, np.float32), (pos, np.uint32)))
            order = np.lexsort((pos, d))[:k]
            n = len(order)
            out_idx[qi] = -1
            out_pos[qi] = -1
            out_dist[qi] = float("inf")
            out_idx[qi, :n] = torch.from_numpy(ids[order].astype(np.int64)).to(torch.int32)
            out_dist[qi, :n] = torch.from_numpy(d[order])
            out_pos[qi, :n] = torch.from_numpy(pos[order].astype(np.int64)).to(torch.int32)
            out_count[qi] = ncand

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


def _worker(rank, world, port, q, exchange="alltoall", traversal="replicated", bin_cap=None, pipelined=False, nq=7):
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
        queries = torch.from_numpy(fx.queries[:nq])  # 7 % 2 != 0 and 7 % 3 != 0: the last query slice is padded (8 ranks: 13 queries, the last rank's slice is EMPTY)
        k, bv, bb = 20, 300, 100
        # a small capacity: some queries' lists overflow and take the traverse-it-yourself fallback
        timer = sh.ExchangeTimer(cuda=False)
        timer.on = True
        if pipelined == "batches":
            # two WHOLE batches in flight: three consecutive steps alternate between two buffer sets; every step must give the full answer
            fl = sh.BatchesInFlight((eng, eng), world, queries.shape[0], k, "cpu", bin_cap=bin_cap or None)
            for _ in range(3):
                oi, od, cnt = fl.step(dist, world, queries, bv, bb, k, exchange=exchange, traversal=traversal, timer=timer)
                oi, od, cnt = oi.clone(), od.clone(), cnt.clone()
            assert fl.n == 3 and fl.last == 0 and fl.result()[0].data_ptr() == fl.bufs[0].out_idx.data_ptr()
            buf = None
        elif pipelined:
            # two half batches in flight (4 + 3 queries: both halves have padded slices), interleaved stage by stage
            pbuf = sh.PipelineBuffers(world, queries.shape[0], k, "cpu", bin_cap=bin_cap or None)
            oi, od, cnt = sh.sharded_query_pipelined((eng, eng), dist, world, queries, bv, bb, k, pbuf, exchange=exchange, traversal=traversal, timer=timer)
            buf = None
        else:
            buf = sh.ShardBuffers(world, queries.shape[0], k, "cpu", bin_cap=bin_cap or None)
            oi, od, cnt = sh.sharded_query(eng, dist, world, queries, bv, bb, k, buf, exchange=exchange, traversal=traversal, timer=timer)
        tm = timer.means_ms()
        want = (["bins_allgather"] if traversal == "sharded" else []) + (["topk_allgather"] if exchange == "allgather" else ["topk_alltoall", "merged_allgather"])
        assert sorted(n_ for n_ in tm if n_ != "calls_timed") == sorted(want) and tm["calls_timed"] == (3 if pipelined == "batches" else 2 if pipelined else 1), tm
        if traversal == "sharded" and buf is not None:
            trailer = buf.bins_all[:queries.shape[0], buf.bin_cap].numpy().view(np.uint64) & np.uint64(0xffffffff)
            over = int((trailer == 0xffffffff).sum())
            assert (over > 0) == bool(bin_cap and bin_cap < 16), (over, bin_cap)  # the fallback is exercised exactly when the capacity is small
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


def _free_port():
    """a port nobody listens on right now (bind to 0): the fixed pid-derived ports of the first version could collide with a socket of an
    earlier case still in TIME_WAIT -- one rank then died in the rendezvous and the others waited for it forever"""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def _run_world(target, world, extra, timeout=180):
    """world ranks of target(rank, world, port, queue, *extra) as daemon processes; their (rank, ok) results.  A rank that dies before it
    reports fails the test at once (with its exit code) instead of leaving the others in a collective, and whatever is still alive when
    the test ends -- pass or fail -- is terminated, so a failure here can never hang the pytest process at exit."""
    import queue as _queue, time
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + tuple(extra), daemon=True) for r in range(world)]
    for p in procs:
        p.start()
    res, deadline = [], time.time() + timeout
    try:
        while len(res) < world:
            try:
                res.append(q.get(timeout=1.0))
                continue
            except _queue.Empty:
                pass
            dead = [(i, p.exitcode) for i, p in enumerate(procs) if p.exitcode not in (None, 0)]
            assert not dead, "ranks died before reporting (rank, exit code): %r; reported so far: %r" % (dead, sorted(res))
            assert time.time() < deadline, "timeout after %d s; reported so far: %r" % (timeout, sorted(res))
        for p in procs:
            p.join(timeout=60)
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
        for p in procs:
            p.join(timeout=10)
    return res



def test_three_rank_gloo_build_time_bin_count_exchange():
    """Shard-by-shard build: per-bin global population / lower / local counts from ONE padded all-gather."""
    res = _run_world(_worker_counts, 3, (), timeout=180)
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
    res = _run_world(_worker, world, (exchange,), timeout=180)
    assert sorted(res) == [(r, True) for r in range(world)]


@pytest.mark.parametrize("world,exchange,bin_cap", [(2, "alltoall", None), (3, "alltoall", None), (3, "allgather", 3), (2, "alltoall", 3), (2, "alltoall", 256)])
def test_gloo_query_sharded_traversal_equals_unsharded(world, exchange, bin_cap):
    """The second exchange: every rank traverses only its query slice, ONE all-gather of the per-query bin lists,
    every rank resolves them against its own slice of the database -- same result as the unsharded engine; with a small list
    capacity the overflowed queries take the traverse-it-yourself fallback."""
    fixture("odd")
    res = _run_world(_worker, world, (exchange, "sharded", bin_cap), timeout=180)
    assert sorted(res) == [(r, True) for r in range(world)]


@pytest.mark.parametrize("world,exchange,traversal,bin_cap", [(2, "alltoall", "sharded", None), (3, "alltoall", "sharded", 3), (3, "allgather", "sharded", None),
                                                              (2, "alltoall", "replicated", None)])
def test_gloo_two_half_batches_in_flight_equal_unsharded(world, exchange, traversal, bin_cap):
    """sharded_query_pipelined: the batch as two halves whose stages (and collectives) are interleaved -- every rank issues the
    collectives of both halves in the same order, and the assembled answer is the unsharded engine's bit for bit."""
    fixture("odd")
    res = _run_world(_worker, world, (exchange, traversal, bin_cap, True), timeout=180)
    assert sorted(res) == [(r, True) for r in range(world)]


@pytest.mark.parametrize("world,exchange,traversal", [(2, "alltoall", "sharded"), (3, "allgather", "replicated")])
def test_gloo_two_whole_batches_in_flight(world, exchange, traversal):
    """BatchesInFlight: consecutive steps alternate between two buffer sets / engines (on the GPU: two streams), every rank issues all
    collectives in program order, and each step returns the unsharded engine's answer."""
    fixture("odd")
    res = _run_world(_worker, world, (exchange, traversal, None, "batches"), timeout=180)
    assert sorted(res) == [(r, True) for r in range(world)]


@pytest.mark.parametrize("exchange,traversal,bin_cap,pipelined,nq", [("alltoall", "sharded", 256, False, 13), ("allgather", "replicated", None, False, 13),
                                                                    ("alltoall", "sharded", 3, "batches", 13), ("alltoall", "sharded", None, True, 7)])
def test_gloo_eight_ranks(exchange, traversal, bin_cap, pipelined, nq):
    """BASELINE configs[3]'s world size: 8 ranks, a query count that is not a multiple of 8 -- 13 queries: slices of
    ceil(13 / 8) = 2, rank 6 owns one query, rank 7 none; 7 queries: slices of 1, rank 7 none --, the 256-entry bin lists, the small ones
    whose overflow sends queries to the traverse-it-yourself fallback, one batch at a time, two whole batches and two half batches in flight."""
    world = 8
    res = _run_world(_worker, world, (exchange, traversal, bin_cap, pipelined, nq), timeout=300)
    assert sorted(res) == [(r, True) for r in range(world)]
