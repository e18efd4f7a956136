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


def sharded_query(engine, dist, world, q, bv, bb, k, buf, exchange="alltoall", force_collectives=False, traversal="replicated", rank=None):
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
        (the protocol of round 1; same result bit for bit)."""
    qn = q.shape[0]
    qs = buf.qs
    if traversal == "sharded":
        if rank is None:
            rank = dist.get_rank() if (world > 1 or force_collectives) else 0
        lo, hi = min(rank * qs, qn), min((rank + 1) * qs, qn)
        if hi > lo:
            engine.traverse_bins(q[lo:hi], bv, bb, buf.bin_cap, buf.bins_local)
        if world == 1 and not force_collectives:
            bins = buf.bins_local
        else:
            dist.all_gather_into_tensor(buf.bins_all, buf.bins_local)
            bins = buf.bins_all
        engine.query_shard_bins(q, bv, bb, k, bins, buf.bin_cap, buf.sh_idx, buf.sh_dist, buf.sh_pos, buf.count)
    else:
        engine.query_shard(q, bv, bb, k, buf.sh_idx, buf.sh_dist, buf.sh_pos, buf.count)
    if world == 1 and not force_collectives:
        engine.merge_topk(1, qn, k, buf.pack[0], buf.pack[1].view(torch.float32), buf.pack[2], buf.out_idx, buf.out_dist, 3 * world * qs * k)
        return buf.out_idx, buf.out_dist, buf.count
    if exchange == "allgather":
        dist.all_gather_into_tensor(buf.gathered.view(world * 3, world * qs, k), buf.pack)
        g = buf.gathered
        engine.merge_topk(world, qn, k, g[0, 0], g[0, 1].view(torch.float32), g[0, 2], buf.out_idx, buf.out_dist, 3 * world * qs * k)
        return buf.out_idx, buf.out_dist, buf.count
    # block r of the send buffer = the three planes of the rows of slice r
    buf.send.copy_(buf.pack.view(3, world, qs, k).permute(1, 0, 2, 3))
    dist.all_to_all_single(buf.recv.view(world * 3, qs, k), buf.send.view(world * 3, qs, k))
    r = buf.recv  # [source shard][3][qs][k]: the layout pqt_merge_topk reads with shard_stride = 3*qs*k
    engine.merge_topk(world, qs, k, r[0, 0], r[0, 1].view(torch.float32), r[0, 2], buf.slice_out[0], buf.slice_out[1].view(torch.float32), 3 * qs * k)
    dist.all_gather_into_tensor(buf.all_out.view(world * 2, qs, k), buf.slice_out)
    buf.out_idx_pad.view(world, qs, k).copy_(buf.all_out[:, 0])
    buf.out_dist_pad.view(torch.int32).view(world, qs, k).copy_(buf.all_out[:, 1])
    return buf.out_idx, buf.out_dist, buf.count


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
