"""Multi-GPU driver logic of the range-sharded query: pure host code, no compute.

One process per GPU.  Rank r owns the database slice with vector ids in shard_range(r, world, N) -- its rows of the
line store and its members of every bin -- while the tree, the heuristic prefix and every bin's GLOBAL population are
replicated, so all ranks apply the identical cut.  Each rank runs the traversal for the whole query batch, reranks
its own slice, and the per-shard top-k lists (id, distance, global visiting position) are exchanged over RCCL/xGMI -- by
query slice (all-to-all) with an all-gather of the merged slices, or with ONE all-gather of the whole lists -- and merged
by (distance, position): exactly the order of the unsharded engine.

The traversal itself does not depend on which vectors a shard holds, so it is sharded by QUERIES (traversal="sharded", the
default of bench.py): rank r traverses only the queries of its slice [r*qs, (r+1)*qs) and hands out, per query, the included
populated bins as (bin id, global visiting position of the first member) in visiting order -- at most BIN_CAP entries of 8
bytes plus one trailer word; ONE all-gather makes every list known everywhere, and each rank resolves the listed bins against
its own bin table (local start, local members, members on lower ranks) before the rerank.  A query whose list does not fit is
traversed by every rank itself.  With W ranks the replicated part of a step shrinks from a whole traversal to 1/W of it plus a
table look-up per listed bin.

The functions take an `engine` exposing
    query_shard(q, bv, bb, k, out_idx, out_dist, out_pos, out_count)
    traverse_bins(q, bv, bb, cap, out_bins)                                  # int64 [n][cap + 1], see include/pqt_hip.h
    query_shard_bins(q, bv, bb, k, bins, cap, out_idx, out_dist, out_pos, out_count)
    merge_topk(world, qn, k, idx0, dist0, pos0, out_idx, out_dist, shard_stride)   # shard s at +s*shard_stride words
so the same code drives the HIP library (PqtShardEngine below) and, in the CPU test-suite, a stand-in over gloo.
"""
import torch


BIN_CAP = 128  # default per-query capacity of the exchanged bin lists (pqt_traverse_bins accepts 1..256; bin_cap_for()); a longer list falls back


def bin_cap_for(bb):
    """Capacity to exchange for a call with bound_bins = bb: the wide traversal (bb > 512) lists more bins per query."""
    return 256 if bb > 512 else BIN_CAP


def shard_range(rank, world, n):
    """Half-open id range [lo, hi) of rank `rank` (contiguous, ordered by rank, covers [0, n))."""
    return rank * n // world, (rank + 1) * n // world


def local_bin_lists(bin_keys, id_lo):
    """CSR of one shard from the bin id of each of its vectors (rows id_lo, id_lo + 1, ...): returns (keys ascending,
    sizes, members) with the members of a bin in ascending id order = the reference's insertion order
    (treequantizer.hpp:212-217, std::map<uint, vector<uint>>::push_back).  Pure tensor plumbing on bin_keys' device."""
    key = bin_keys.to(torch.int64) & 0xffffffff
    order = torch.argsort(key, stable=True)
    ukeys, counts = torch.unique_consecutive(key[order], return_counts=True)
    return ukeys, counts, order + id_lo


def merge_bin_counts(all_keys, all_counts, rank):
    """Pure part of the build-time exchange: from every shard's (bin ids ascending, local populations) derive, for the union
    of bins in ascending id order, gsize = global population (drives the identical cut on every shard), lower = members
    on lower ranks (offset of this shard's members inside the bin's global member list), lsize = members on `rank`."""
    dev = all_keys[0].device
    ks = torch.cat([k.to(torch.int64) for k in all_keys])
    cs = torch.cat([c.to(torch.int64) for c in all_counts])
    rs = torch.cat([torch.full((k.numel(),), r, dtype=torch.int64, device=dev) for r, k in enumerate(all_keys)])
    uk, inv = torch.unique(ks, return_inverse=True)  # sorted ascending
    z = torch.zeros(uk.numel(), dtype=torch.int64, device=dev)
    gsize = z.clone().index_add_(0, inv, cs)
    lower = z.clone().index_add_(0, inv[rs < rank], cs[rs < rank])
    lsize = z.clone().index_add_(0, inv[rs == rank], cs[rs == rank])
    return uk, gsize, lower, lsize


def global_bin_counts(dist, world, rank, keys, counts, force_collectives=False):
    """Build-time exchange of a database built shard by shard (the CSR merge of test/test1B.cpp:783-871 reduced to the
    per-bin counts): every rank contributes (bin id, local population) of its non-empty bins through ONE padded
    all-gather and derives merge_bin_counts() of the gathered lists.
    keys/counts: int64 tensors (keys ascending, unique).  Returns (ukeys, gsize, lower, lsize) as int64 tensors."""
    dev = keys.device
    keys, counts = keys.to(torch.int64), counts.to(torch.int64)
    if world == 1 and not force_collectives:
        return keys, counts, torch.zeros_like(counts), counts
    n = torch.tensor([keys.numel()], dtype=torch.int64, device=dev)
    ns = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(ns, n)
    ns = ns.tolist()
    m = max(max(ns), 1)
    pad = torch.zeros((2, m), dtype=torch.int64, device=dev)
    pad[0, :keys.numel()] = keys
    pad[1, :keys.numel()] = counts
    allp = torch.empty((world * 2, m), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(allp, pad)
    allp = allp.view(world, 2, m)
    return merge_bin_counts([allp[r, 0, :ns[r]] for r in range(world)], [allp[r, 1, :ns[r]] for r in range(world)], rank)


class ShardBuffers:
    """Device buffers of the sharded hot path, allocated once.  The per-shard top-k of a batch is one [3][W*qs][k] block of
    32-bit words (idx | dist bits | global visiting position; qs = ceil(qn / W) queries per slice, the rows beyond qn are
    permanent padding: idx 0xffffffff, dist +inf), written in place by the shard kernels."""

    def __init__(self, world, qn, k, device, bin_cap=None):
        i32 = torch.int32
        self.world, self.qn, self.k = world, qn, k
        self.qs = qs = (qn + world - 1) // world
        self.pack = torch.empty((3, world * qs, k), dtype=i32, device=device)
        self.pack[0].fill_(-1)
        self.pack[1].view(torch.float32).fill_(float("inf"))
        self.pack[2].fill_(-1)
        self.sh_idx, self.sh_pos = self.pack[0][:qn], self.pack[2][:qn]
        self.sh_dist = self.pack[1][:qn].view(torch.float32)
        self.count = torch.empty(qn, dtype=i32, device=device)
        # exchange = "allgather": every rank receives every shard's whole message and merges all queries
        self.gathered = torch.empty((world, 3, world * qs, k), dtype=i32, device=device)
        # exchange = "alltoall": rank r receives, from every shard, the rows of ITS query slice only, merges those qs
        # queries, and the merged slices are all-gathered
        self.send = torch.empty((world, 3, qs, k), dtype=i32, device=device)
        self.recv = torch.empty((world, 3, qs, k), dtype=i32, device=device)
        self.slice_out = torch.empty((2, qs, k), dtype=i32, device=device)
        self.all_out = torch.empty((world, 2, qs, k), dtype=i32, device=device)
        self.out_idx_pad = torch.empty((world * qs, k), dtype=i32, device=device)
        self.out_dist_pad = torch.empty((world * qs, k), dtype=torch.float32, device=device)
        self.out_idx, self.out_dist = self.out_idx_pad[:qn], self.out_dist_pad[:qn]
        # traversal = "sharded": this rank's bin lists of its query slice, and everybody's after the all-gather.  Rows of the
        # padding queries beyond qn stay zero (an empty list: trailer count 0).
        self.bin_cap = int(BIN_CAP if bin_cap is None else bin_cap)
        self.bins_local = torch.zeros((qs, self.bin_cap + 1), dtype=torch.int64, device=device)
        self.bins_all = torch.zeros((world * qs, self.bin_cap + 1), dtype=torch.int64, device=device)


class ExchangeTimer:
    """Device time of each collective of a sharded step: event pairs on the stream the step is enqueued on (the collective's
    own stream is joined to it on both sides by torch.distributed), or host clocks when the tensors live on the CPU (gloo tests).
    Armed per step by the caller (`on`); read after the closing barrier with means_ms()."""
    NAMES = ("bins_allgather", "topk_alltoall", "merged_allgather", "topk_allgather")

    def __init__(self, cuda):
        self.cuda, self.on = bool(cuda), False
        self.spans = {n: [] for n in self.NAMES}

    def begin(self, name):
        if not self.on:
            return None
        if self.cuda:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
            return (name, e0)
        import time
        return (name, time.perf_counter())

    def end(self, tok):
        if tok is None:
            return
        name, t0 = tok
        if self.cuda:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self.spans[name].append((t0, e1))
        else:
            import time
            self.spans[name].append(time.perf_counter() - t0)

    def means_ms(self):
        """{collective: mean ms per call} over the armed steps (call after a device synchronisation)."""
        out = {}
        for n, v in self.spans.items():
            if not v:
                continue
            ms = [a.elapsed_time(b) for a, b in v] if self.cuda else [x * 1e3 for x in v]
            out[n] = sum(ms) / len(ms)
        out["calls_timed"] = max([len(v) for v in self.spans.values()] + [0])
        return out


def _stages(engine, dist, world, q, bv, bb, k, buf, exchange, force_collectives, traversal, rank, timer, delay=None):
    """The stages of one sharded step as closures, in issue order; every collective is a stage of its own so that two half batches
    can be interleaved stage by stage (sharded_query_pipelined).  `delay` (seconds, test/measurement aid): called after every
    collective with the stream's tensors -- the one-device harness injects a synthetic collective latency through it."""
    qn = q.shape[0]
    qs = buf.qs
    coll = world > 1 or force_collectives
    st = {}

    def timed(name, fn):
        tok = timer.begin(name) if timer is not None else None
        fn()
        if delay is not None:
            delay()
        if timer is not None:
            timer.end(tok)

    stages = []
    if traversal == "sharded":
        r = rank if rank is not None else (dist.get_rank() if coll else 0)
        lo, hi = min(r * qs, qn), min((r + 1) * qs, qn)

        def s_traverse():
            if hi > lo:
                engine.traverse_bins(q[lo:hi], bv, bb, buf.bin_cap, buf.bins_local)
        stages.append(s_traverse)
        if coll:
            stages.append(lambda: timed("bins_allgather", lambda: dist.all_gather_into_tensor(buf.bins_all, buf.bins_local)))
        else:
            stages.append(lambda: None)

        def s_rerank():
            engine.query_shard_bins(q, bv, bb, k, buf.bins_all if coll else buf.bins_local, buf.bin_cap, buf.sh_idx, buf.sh_dist, buf.sh_pos, buf.count)
        stages.append(s_rerank)
    else:
        stages.append(lambda: None)
        stages.append(lambda: None)
        stages.append(lambda: engine.query_shard(q, bv, bb, k, buf.sh_idx, buf.sh_dist, buf.sh_pos, buf.count))
    if not coll:
        stages.append(lambda: engine.merge_topk(1, qn, k, buf.pack[0], buf.pack[1].view(torch.float32), buf.pack[2], buf.out_idx, buf.out_dist, 3 * world * qs * k))
        return stages
    if exchange == "allgather":
        stages.append(lambda: timed("topk_allgather", lambda: dist.all_gather_into_tensor(buf.gathered.view(world * 3, world * qs, k), buf.pack)))

        def s_merge_all():
            g = buf.gathered
            engine.merge_topk(world, qn, k, g[0, 0], g[0, 1].view(torch.float32), g[0, 2], buf.out_idx, buf.out_dist, 3 * world * qs * k)
        stages.append(s_merge_all)
        return stages

    def s_a2a():
        # block r of the send buffer = the three planes of the rows of slice r
        buf.send.copy_(buf.pack.view(3, world, qs, k).permute(1, 0, 2, 3))
        timed("topk_alltoall", lambda: dist.all_to_all_single(buf.recv.view(world * 3, qs, k), buf.send.view(world * 3, qs, k)))
    stages.append(s_a2a)

    def s_merge():
        r_ = buf.recv  # [source shard][3][qs][k]: the layout pqt_merge_topk reads with shard_stride = 3*qs*k
        engine.merge_topk(world, qs, k, r_[0, 0], r_[0, 1].view(torch.float32), r_[0, 2], buf.slice_out[0], buf.slice_out[1].view(torch.float32), 3 * qs * k)
    stages.append(s_merge)
    stages.append(lambda: timed("merged_allgather", lambda: dist.all_gather_into_tensor(buf.all_out.view(world * 2, qs, k), buf.slice_out)))

    def s_out():
        buf.out_idx_pad.view(world, qs, k).copy_(buf.all_out[:, 0])
        buf.out_dist_pad.view(torch.int32).view(world, qs, k).copy_(buf.all_out[:, 1])
    stages.append(s_out)
    return stages


def sharded_query(engine, dist, world, q, bv, bb, k, buf, exchange="alltoall", force_collectives=False, traversal="replicated", rank=None, timer=None, delay=None):
    """One step of the sharded hot path.  Returns (out_idx, out_dist, count) views into `buf`.

    traversal = "replicated": every rank traverses the whole batch (pqt_query_shard).
    traversal = "sharded": rank r traverses the queries of its slice only (pqt_traverse_bins), ONE all-gather of the per-query
        bin lists ([qs][BIN_CAP + 1] 8-byte words per rank: 10 MB in total for 10 k queries), then every rank resolves the lists
        against its own table and reranks its slice of the database (pqt_query_shard_bins).  Same result bit for bit.

    exchange = "alltoall" (default): the per-shard top-k lists travel by query slice -- rank r gets from every shard only
        the rows of queries [r*qs, (r+1)*qs) (1/W of each message: on xGMI's point-to-point links every pair moves its own
        12/W MB concurrently), merges those qs queries (1/W of the merge work), and ONE all-gather of the merged [2][qs][k]
        slices (idx | dist, 8 bytes per result) gives every rank the whole answer.  Per rank and batch of 10 k queries,
        k = 100, W = 8: 10.5 MB + 7 MB received instead of 84 MB, 1250 merged queries instead of 10 000.
    exchange = "allgather": the single all-gather of the whole [3][qn][k] messages + a merge of all queries on every rank
        (the protocol of round 1; same result bit for bit).
    timer: an ExchangeTimer (armed by the caller) brackets every collective."""
    for stage in _stages(engine, dist, world, q, bv, bb, k, buf, exchange, force_collectives, traversal, rank, timer, delay):
        stage()
    return buf.out_idx, buf.out_dist, buf.count


class PipelineBuffers:
    """Two half batches in flight (SURVEY.md 8e "one collective launch latency per batch, hidden by double-buffering batches"):
    the batch is split into halves [0, h) and [h, qn), each with its own ShardBuffers, engine (the index and a view of it: a
    handle serves one batch at a time) and stream; the whole answer is assembled in out_idx / out_dist / count."""

    def __init__(self, world, qn, k, device, bin_cap=None, cuda=None):
        self.qn, self.k = qn, k
        self.h = (qn + 1) // 2
        self.halves = [ShardBuffers(world, self.h, k, device, bin_cap), ShardBuffers(world, max(qn - self.h, 1), k, device, bin_cap)]
        self.out_idx = torch.empty((qn, k), dtype=torch.int32, device=device)
        self.out_dist = torch.empty((qn, k), dtype=torch.float32, device=device)
        self.count = torch.empty(qn, dtype=torch.int32, device=device)
        self.cuda = (torch.device(device).type == "cuda") if cuda is None else cuda
        self.streams = [torch.cuda.Stream(device), torch.cuda.Stream(device)] if self.cuda else [None, None]
        self.bin_cap = self.halves[0].bin_cap


def sharded_query_pipelined(engines, dist, world, q, bv, bb, k, pbuf, exchange="alltoall", force_collectives=False, traversal="sharded", rank=None,
                            timer=None, delay=None):
    """sharded_query for the two halves of the batch, interleaved stage by stage on two streams: while half A's collective is in
    flight (a launch latency, not bandwidth: 10 MB per batch) half B's kernels run, and the other way round.  The collectives of
    both halves are issued in the same order on every rank.  engines = (engine of the index, engine of a view of it).  The result
    is sharded_query's, bit for bit (each query's answer does not depend on which other queries share its launch).
    Returns (out_idx, out_dist, count) of the whole batch."""
    qn = q.shape[0]
    h = pbuf.h
    if qn - h < 1:
        oi, od, oc = sharded_query(engines[0], dist, world, q, bv, bb, k, pbuf.halves[0], exchange, force_collectives, traversal, rank, timer, delay)
        pbuf.out_idx.copy_(oi), pbuf.out_dist.copy_(od), pbuf.count.copy_(oc)
        return pbuf.out_idx, pbuf.out_dist, pbuf.count
    parts = [(0, h), (h, qn)]
    plans = [_stages(engines[i], dist, world, q[a:b], bv, bb, k, pbuf.halves[i], exchange, force_collectives, traversal, rank, timer, delay)
             for i, (a, b) in enumerate(parts)]

    def out_copy(i):
        a, b = parts[i]
        hb = pbuf.halves[i]
        pbuf.out_idx[a:b].copy_(hb.out_idx)
        pbuf.out_dist[a:b].copy_(hb.out_dist)
        pbuf.count[a:b].copy_(hb.count)

    if not pbuf.cuda:
        for sidx in range(len(plans[0])):
            for i in range(2):
                plans[i][sidx]()
        out_copy(0), out_copy(1)
        return pbuf.out_idx, pbuf.out_dist, pbuf.count
    cur = torch.cuda.current_stream()
    fork = torch.cuda.Event()
    fork.record(cur)
    for s_ in pbuf.streams:
        s_.wait_event(fork)
        if q.is_cuda:
            q.record_stream(s_)  # (the halves' streams read slices of q: see BatchesInFlight.step)
    for sidx in range(len(plans[0])):
        for i in range(2):
            with torch.cuda.stream(pbuf.streams[i]):
                plans[i][sidx]()
    for i in range(2):
        with torch.cuda.stream(pbuf.streams[i]):
            out_copy(i)
        done = torch.cuda.Event()
        done.record(pbuf.streams[i])
        cur.wait_event(done)
    return pbuf.out_idx, pbuf.out_dist, pbuf.count


class BatchesInFlight:
    """Two WHOLE batches in flight (the collective latency "hidden by double-buffering batches"): consecutive steps
    alternate between two slots -- each with its own ShardBuffers, engine (the index and a view of it: a handle serves one batch at
    a time) and stream -- so the collectives of one batch pass while the other batch's kernels run, and every kernel keeps its full
    size (splitting ONE batch into halves, sharded_query_pipelined, doubles the tails of the persistent rerank launches and was
    measured slower on the one-device harness, profiles/r04_pipeline_one_device_*.json).  Every rank issues the collectives of all
    steps in the same program order.  step() returns the slot's (out_idx, out_dist, count) views: valid once the slot's stream has
    run (wait() / a device synchronisation), overwritten two steps later."""

    def __init__(self, engines, world, qn, k, device, bin_cap=None, cuda=None):
        self.engines = list(engines)
        self.bufs = [ShardBuffers(world, qn, k, device, bin_cap), ShardBuffers(world, qn, k, device, bin_cap)]
        self.cuda = (torch.device(device).type == "cuda") if cuda is None else cuda
        self.streams = [torch.cuda.Stream(device), torch.cuda.Stream(device)] if self.cuda else [None, None]
        self.n = 0
        self.last = None
        self.done = [None, None]  # event behind the last step of each slot: ready(slot) / wait_slot(slot)
        if self.cuda:  # whatever the caller enqueued so far (the queries) is visible to both streams
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            for s_ in self.streams:
                s_.wait_event(ev)

    def step(self, dist, world, q, bv, bb, k, **kw):
        slot = self.n & 1
        self.n += 1
        self.last = slot
        if self.cuda:
            # what the caller enqueued on its stream up to here (a new query batch, say) is visible to the slot's stream
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.streams[slot].wait_event(ev)
            if q.is_cuda:
                # the slot's stream reads q (and slices of it) after step() has returned: tell the caching allocator, so that a caller
                # which drops or reuses the batch right away does not get its memory handed out while the slot still reads it
                q.record_stream(self.streams[slot])
            with torch.cuda.stream(self.streams[slot]):
                out = sharded_query(self.engines[slot], dist, world, q, bv, bb, k, self.bufs[slot], **kw)
            self.done[slot] = torch.cuda.Event()
            self.done[slot].record(self.streams[slot])
            return out
        return sharded_query(self.engines[slot], dist, world, q, bv, bb, k, self.bufs[slot], **kw)

    def wait(self):
        """The caller's current stream waits for everything issued on both slots."""
        if self.cuda:
            cur = torch.cuda.current_stream()
            for s_ in self.streams:
                ev = torch.cuda.Event()
                ev.record(s_)
                cur.wait_event(ev)

    def wait_slot(self, slot=None):
        """The caller's current stream waits for the last step issued on `slot` (default: the most recent step): its views are then valid
        for work enqueued on that stream; they are overwritten by the step after next."""
        slot = self.last if slot is None else slot
        if self.cuda and slot is not None and self.done[slot] is not None:
            torch.cuda.current_stream().wait_event(self.done[slot])

    def result(self):
        b = self.bufs[self.last if self.last is not None else 0]
        return b.out_idx, b.out_dist, b.count


class PqtShardEngine:
    """Adapter of a sharded PqtIndex (HIP) to the engine protocol; enqueues on the current torch stream."""

    def __init__(self, index):
        self.index = index

    def _stream(self):
        return torch.cuda.current_stream().cuda_stream

    def query_shard(self, q, bv, bb, k, out_idx, out_dist, out_pos, out_count):
        self.index.query_shard_dev(q, bv, bb, k, out_idx, out_dist, out_pos, out_count, stream=self._stream())

    def traverse_bins(self, q, bv, bb, cap, out_bins):
        self.index.traverse_bins_dev(q, bv, bb, cap, out_bins, stream=self._stream())

    def query_shard_bins(self, q, bv, bb, k, bins, cap, out_idx, out_dist, out_pos, out_count):
        self.index.query_shard_bins_dev(q, bv, bb, k, bins, cap, out_idx, out_dist, out_pos, out_count, stream=self._stream())

    def merge_topk(self, world, qn, k, all_idx, all_dist, all_pos, out_idx, out_dist, shard_stride):
        self.index.merge_topk_dev(world, qn, k, all_idx, all_dist, all_pos, out_idx, out_dist, stream=self._stream(),
                                  shard_stride=shard_stride)
