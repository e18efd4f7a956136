#!/usr/bin/env python3
"""bench.py -- PQT query hot path on MI355X: queries/sec + recall, HBM roofline of the dominant kernel,
CPU baseline beside it.

One "step" = one pass of the whole hot path (distance tables -> traversal -> bin enumeration -> ADC line
rerank -> top-k) over one batch of QN synthetic SIFT-shaped queries, inputs and outputs resident in HBM.

    python bench.py                       # N=1, BASELINE.json configs[1] (SIFT1M shape, batch 10k)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Multi-GPU (--gpus N > 1), one process per GPU.  Two ways the path shards:
  * default for this workload (the 1 M-vector index is 70 MB, it fits every GPU): queries are the units -- every rank
    holds the whole index and answers its own batch of QN queries; no data-path collective; per-GPU work is fixed
    as N grows => "scaling": "weak", value = N*QN*steps / time.
  * --shard-db (the north-star layout for databases that do not fit one GPU, SIFT1B): the database is range-sharded
    by vector id, every rank runs the traversal for the whole batch, reranks its own slice, and the per-shard top-k
    lists are merged after ONE RCCL all-gather => "scaling": "strong", value = QN*steps / time.  A short leg in this
    layout also runs after the default one and is reported as config.db_sharded (never as `value`).
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)

WORKLOADS = {
    # BASELINE.json configs[1]: "SIFT1M d=128, p=4, c1=32, c2=32 on 1xMI355X, batch=10k queries" (lineparts=16 per configs[0])
    "sift1m": dict(D=128, P=4, C1=32, C2=32, W=2, LP=16, n_base=1_000_000, n_train=100_000, qn=10_000),
    # BASELINE.json configs[2] shape: "Synthetic 100M x d=128 float, p=4, c1=64, c2=64, lineparts=32 on 1xMI355X" (optional
    # run, not the bench line; the index is synthesised chunk by chunk with the product's own build kernel)
    "synth100m": dict(D=128, P=4, C1=64, C2=64, W=1, LP=32, n_base=100_000_000, n_train=200_000, qn=10_000, chunk=4_000_000),
    "synth10m": dict(D=128, P=4, C1=64, C2=64, W=1, LP=32, n_base=10_000_000, n_train=200_000, qn=10_000, chunk=2_000_000),
    # small variant for quick checks (not a bench line)
    "tiny": dict(D=128, P=4, C1=32, C2=32, W=2, LP=16, n_base=50_000, n_train=20_000, qn=1_000),
}


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print(*a, file=sys.stderr, flush=True)


# ------------------------------------------------------------------------------------------------------
# synthetic SIFT-shaped data (generated on the device; plumbing, not the product)
# ------------------------------------------------------------------------------------------------------
GEN = dict(n_centers=4096, latent=24, lat_noise=4.0, iso_noise=6.0, center_scale=30.0)


def sift_like(n, D, seed, dev):
    n_centers, latent = GEN['n_centers'], GEN['latent']
    g = torch.Generator(device=dev)
    g.manual_seed(0xC0DE00)  # the mixture itself is shared by train/base/query
    A = torch.randn(latent, D, generator=g, device=dev)
    centers = torch.randn(n_centers, latent, generator=g, device=dev) * GEN['center_scale']
    g.manual_seed(seed)
    out = torch.empty((n, D), dtype=torch.float32, device=dev)
    step = 1 << 18
    for s in range(0, n, step):
        m = min(step, n - s)
        which = torch.randint(0, n_centers, (m,), generator=g, device=dev)
        z = centers[which] + torch.randn(m, latent, generator=g, device=dev) * GEN['lat_noise']
        x = 100.0 + (z @ A) * 0.8 + torch.randn(m, D, generator=g, device=dev) * GEN['iso_noise']
        out[s:s + m] = x.round().clamp_(0, 255)
    return out


def kmeans(x, k, iters, g):
    n = x.shape[0]
    if n == 0:
        return torch.rand((k, x.shape[1]), device=x.device, generator=g) * 255.0
    cen = x[torch.randperm(n, device=x.device, generator=g)[:k]].clone()
    if cen.shape[0] < k:  # tiny cell: pad with jittered copies (distinct centroids, no exact duplicates)
        extra = cen[torch.randint(0, cen.shape[0], (k - cen.shape[0],), device=x.device, generator=g)]
        cen = torch.cat([cen, extra + torch.rand(extra.shape, device=x.device, generator=g) * 4.0 + 0.5])
    for _ in range(iters):
        a = torch.cdist(x, cen).argmin(1)
        s = torch.zeros_like(cen).index_add_(0, a, x)
        c = torch.zeros(k, device=x.device).index_add_(0, a, torch.ones(n, device=x.device))
        cen = torch.where(c[:, None] > 0, s / c.clamp(min=1)[:, None], cen)
    return cen


def train_codebooks(train, P, C1, C2, seed):
    dev = train.device
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    D = train.shape[1]
    S = D // P
    cb1 = torch.zeros((C1, D), device=dev)
    cb2 = torch.zeros((P, C1, C2, S), device=dev)
    for p in range(P):
        seg = train[:, p * S:(p + 1) * S].contiguous()
        cen = kmeans(seg, C1, 12, g)
        cb1[:, p * S:(p + 1) * S] = cen
        a = torch.cdist(seg, cen).argmin(1)
        for c in range(C1):
            cb2[p, c] = kmeans(seg[a == c], C2, 8, g)
    return cb1.cpu().numpy(), cb2.cpu().numpy()


def brute_force_gt(base, queries, k):
    out = torch.empty((queries.shape[0], k), dtype=torch.int64, device=base.device)
    bn = (base * base).sum(1)
    for s in range(0, queries.shape[0], 2048):
        q = queries[s:s + 2048]
        d = bn[None, :] - 2.0 * (q @ base.T)
        out[s:s + 2048] = d.topk(k, dim=1, largest=False).indices
    return out


# ------------------------------------------------------------------------------------------------------
def build_index(pkg, w, dev_index, seed_base=0xC0DE02, shard=None):
    """Synthesise the database and load it into a PqtIndex.  Returns (index, base (device), meta)."""
    dev = torch.device("cuda", dev_index)
    D, P, C1, C2, W, LP = (w[k] for k in ("D", "P", "C1", "C2", "W", "LP"))
    t0 = time.time()
    train = sift_like(w["n_train"], D, 0xC0DE01, dev)
    base = sift_like(w["n_base"], D, seed_base, dev)
    cb1, cb2 = train_codebooks(train, P, C1, C2, 0xC0DE04)
    del train
    t1 = time.time()
    idx = pkg.PqtIndex(D, P, C1, C2, W, LP, device=dev_index)
    idx.set_codebooks(cb1, cb2)
    n = base.shape[0]
    bins = torch.empty(n, dtype=torch.int32, device=dev)
    codes = torch.empty((n, LP), dtype=torch.int32, device=dev)
    idx.assign_encode_dev(base, bins, codes, stream=torch.cuda.current_stream(dev).cuda_stream)  # product kernel: insert = id() + prepareReranking (same stream as the data synthesis)
    torch.cuda.synchronize(dev)
    t2 = time.time()
    # CSR by bin id (vector ids ascending inside a bin = the reference's insertion order)
    key = (bins.to(torch.int64) & 0xffffffff)
    order = torch.argsort(key, stable=True)
    ukeys, counts = torch.unique_consecutive(key[order], return_counts=True)
    bin_ids = ukeys.cpu().numpy().astype(np.uint32)
    sizes = counts.cpu().numpy().astype(np.uint32)
    members = order.cpu().numpy().astype(np.uint32)
    if shard is None:
        idx.set_bins(bin_ids, sizes, members)
        idx.set_lines_dev(codes, 0)
    else:
        lo, hi = shard
        idx.set_bins_shard(bin_ids, sizes, members, lo, hi)
        local = codes[lo:hi].clone()
        del codes
        idx.set_lines_dev(local, lo)
    t3 = time.time()
    meta = dict(n_bins=int(bin_ids.shape[0]), max_bin=int(sizes.max()), t_data=t1 - t0, t_encode=t2 - t1, t_csr=t3 - t2,
                cb1=cb1, cb2=cb2, bin_ids=bin_ids, sizes=sizes, members=members)
    return idx, base, meta


def usable_cores(omp_max):
    """Host threads this process may really run: affinity mask and cgroup CPU quota, not the machine's core count."""
    n = min(omp_max, len(os.sched_getaffinity(0)))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def build_index_chunked(pkg, w, dev_index):
    """Large databases: vectors are generated, assigned and line-encoded chunk by chunk (never resident as a whole)."""
    dev = torch.device("cuda", dev_index)
    D, P, C1, C2, W, LP = (w[k] for k in ("D", "P", "C1", "C2", "W", "LP"))
    n, chunk = w["n_base"], w["chunk"]
    t0 = time.time()
    train = sift_like(w["n_train"], D, 0xC0DE01, dev)
    cb1, cb2 = train_codebooks(train, P, C1, C2, 0xC0DE04)
    del train
    idx = pkg.PqtIndex(D, P, C1, C2, W, LP, device=dev_index)
    idx.set_codebooks(cb1, cb2)
    bins = torch.empty(n, dtype=torch.int32, device=dev)
    codes = torch.empty((n, LP), dtype=torch.int32, device=dev)
    t1 = time.time()
    for ci, s0 in enumerate(range(0, n, chunk)):
        m = min(chunk, n - s0)
        x = sift_like(m, D, 0xC0DE02 + 7919 * ci, dev)
        idx.assign_encode_dev(x, bins[s0:s0 + m], codes[s0:s0 + m], stream=torch.cuda.current_stream(dev).cuda_stream)
        del x
    torch.cuda.synchronize(dev)
    t2 = time.time()
    key = bins.to(torch.int64) & 0xffffffff
    del bins
    order = torch.argsort(key, stable=True)
    ukeys, counts = torch.unique_consecutive(key[order], return_counts=True)
    del key
    bin_ids = ukeys.cpu().numpy().astype(np.uint32)
    sizes = counts.cpu().numpy().astype(np.uint32)
    members = order.cpu().numpy().astype(np.uint32)
    del order, ukeys, counts
    torch.cuda.empty_cache()
    idx.set_bins(bin_ids, sizes, members)
    idx.set_lines_dev(codes, 0)
    t3 = time.time()
    meta = dict(n_bins=int(bin_ids.shape[0]), max_bin=int(sizes.max()), t_data=t1 - t0, t_encode=t2 - t1, t_csr=t3 - t2,
                cb1=cb1, cb2=cb2, bin_ids=bin_ids, sizes=sizes, members=members)
    return idx, None, meta


def brute_force_gt_chunked(w, queries, dev):
    """Exact nearest neighbour over the chunk-generated database (chunks are regenerated from their seeds)."""
    n, chunk, D = w["n_base"], w["chunk"], w["D"]
    best_d = torch.full((queries.shape[0],), float("inf"), device=dev)
    best_i = torch.zeros(queries.shape[0], dtype=torch.int64, device=dev)
    for ci, s0 in enumerate(range(0, n, chunk)):
        m = min(chunk, n - s0)
        x = sift_like(m, D, 0xC0DE02 + 7919 * ci, dev)
        bn = (x * x).sum(1)
        for a in range(0, queries.shape[0], 2048):
            q = queries[a:a + 2048]
            d = bn[None, :] - 2.0 * (q @ x.T)
            v, i = d.min(1)
            v = v + (q * q).sum(1)
            upd = v < best_d[a:a + 2048]
            best_d[a:a + 2048] = torch.where(upd, v, best_d[a:a + 2048])
            best_i[a:a + 2048] = torch.where(upd, i + s0, best_i[a:a + 2048])
        del x, bn
    return best_i


def recall_at(ids, gt0, r):
    r = min(r, ids.shape[1])
    return float((ids[:, :r] == gt0[:, None]).any(1).float().mean())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="sift1m", choices=list(WORKLOADS))
    ap.add_argument("--bv", type=int, default=20000, help="boundVectors (reference default: query(20000, 500, ...))")
    ap.add_argument("--bb", type=int, default=500, help="boundBins")
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--extras", action="store_true", help="also run the side legs (knob set (4096,4096), exact re-rank of the top-k); "
                    "off by default so that a profile of the default command contains only the headline path's launches")
    ap.add_argument("--shard-db", action="store_true", help="multi-GPU: range-shard the database instead of the queries")
    ap.add_argument("--iso-noise", type=float, default=GEN["iso_noise"])
    ap.add_argument("--lat-noise", type=float, default=GEN["lat_noise"])
    ap.add_argument("--centers", type=int, default=GEN["n_centers"])
    ap.add_argument("--center-scale", type=float, default=GEN["center_scale"])
    ap.add_argument("--query-mode", default="fresh", choices=["fresh", "perturbed"])
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world == 1 and args.gpus > 1:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("PQT_BENCH_BACKEND", "nccl")  # "gloo" + PQT_BENCH_SAME_DEVICE=1: functional check on a 1-GPU box
        if os.environ.get("PQT_BENCH_SAME_DEVICE"):
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    # everything (data synthesis and the library's kernels) is enqueued on ONE explicit stream: the C-ABI treats a NULL
    # stream as "the handle's own non-blocking stream", which is not ordered with torch's legacy default stream
    work_stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(work_stream)

    GEN.update(iso_noise=args.iso_noise, lat_noise=args.lat_noise, n_centers=args.centers, center_scale=args.center_scale)
    pkg = importlib.import_module("product-quantization-tree_amd")
    pkg.lib()  # fails loudly if the HIP library is missing
    w = WORKLOADS[args.workload]
    n = w["n_base"]
    sharding = importlib.import_module("product-quantization-tree_amd.sharding")
    mode = "single" if world == 1 else ("shard_db" if args.shard_db else "replica")
    shard = sharding.shard_range(rank, world, n) if mode == "shard_db" else None
    chunked = "chunk" in w
    if chunked:
        if mode != "single":
            raise SystemExit("the chunk-built workloads are single-GPU runs")
        idx, base, meta = build_index_chunked(pkg, w, local_rank)
    else:
        idx, base, meta = build_index(pkg, w, local_rank, shard=shard)
    t0 = time.time()
    idx.build_heuristic(max(args.bb, 1))
    log("[bench] index: N=%d bins=%d max_bin=%d  data %.1fs encode %.1fs csr %.1fs heuristic %.1fs" %
        (n, meta["n_bins"], meta["max_bin"], meta["t_data"], meta["t_encode"], meta["t_csr"], time.time() - t0))

    # queries = perturbed base rows (seed 0xC0DE03), ground truth by exact brute force
    g = torch.Generator(device=dev)
    g.manual_seed(0xC0DE03)
    qn = w["qn"]
    if args.query_mode == "perturbed":
        pick = torch.randint(0, n, (qn,), generator=g, device=dev)
        queries = (base[pick] + torch.randn(qn, w["D"], generator=g, device=dev) * 8.0).round().clamp_(0, 255).contiguous()
    else:  # fresh draws from the same mixture (like SIFT's separate query set)
        queries = sift_like(qn, w["D"], 0xC0DE03 + (1000 * rank if mode == "replica" else 0), dev)
    if os.environ.get("PQT_EXP_QPERM"):
        # experiment: give every XCD (workgroup b of the static rerank schedule runs on XCD b % 8) the queries of one
        # contiguous slab of the last part's first-level cells, so that an XCD's L2 sees 1/8 of the code store
        S_ = w["D"] // w["P"]
        cb1_t = torch.from_numpy(meta["cb1"]).to(dev)[:, (w["P"] - 1) * S_:]
        cell = torch.cdist(queries[:, (w["P"] - 1) * S_:], cb1_t).argmin(1)
        srt = torch.argsort(cell, stable=True)
        pos = torch.arange(qn, device=dev)
        xcd = ((pos % 2048) // 8) % 8 if os.environ["PQT_EXP_QPERM"] == "xcd" else (pos * 8 // qn)
        dest = torch.argsort(xcd, stable=True)          # batch positions grouped by the XCD that will serve them
        perm = torch.empty(qn, dtype=torch.int64, device=dev)
        perm[dest] = srt                                # position dest[i] receives the i-th query of the cell order
        queries = queries[perm].contiguous()
    if chunked:
        gt = brute_force_gt_chunked(w, queries, dev)
        raw_u8 = None
    else:
        gt = brute_force_gt(base, queries, 1)[:, 0]
        raw_u8 = base.to(torch.uint8) if (args.extras and mode != "shard_db") else None  # raw vectors for the optional exact re-rank (8f-4)
    del base
    torch.cuda.empty_cache()

    k = args.k
    out_idx = torch.empty((qn, k), dtype=torch.int32, device=dev)
    out_dist = torch.empty((qn, k), dtype=torch.float32, device=dev)
    out_cnt = torch.empty(qn, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    if mode == "shard_db":
        sbuf = sharding.ShardBuffers(world, qn, k, dev)
        engine = sharding.PqtShardEngine(idx)

    def step():
        if mode != "shard_db":
            idx.query_dev(queries, args.bv, args.bb, k, out_idx, out_dist, out_cnt, stream=stream)
        else:
            # traversal for the whole batch + rerank of the local slice, ONE RCCL all-gather, exact merge
            oi, od, oc = sharding.sharded_query(engine, dist, world, queries, args.bv, args.bb, k, sbuf)
            out_idx.copy_(oi)
            out_dist.copy_(od)
            out_cnt.copy_(oc)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    # per-stage device times of the timed steps themselves: the library records HIP events around every kernel on the
    # stream it launches on (ring of the last 32 calls); they are read only now, after the closing barrier.
    hist = idx.stage_ms_history(min(args.steps, 32))
    st = idx.stats()
    stage = dict(zip(("tables", "traverse", "gap", "rerank_select", "select"), hist.mean(0).tolist()))
    if os.environ.get("PQT_TSTAMP"):
        import ctypes
        ts = np.zeros((qn, 16), np.uint64)
        L = pkg.lib()
        L.pqt_debug_tstamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
        if L.pqt_debug_tstamps(idx.h, ts.ctypes.data, qn) == 0:
            d = np.diff(ts[:, :9].astype(np.int64), axis=1)
            log("[tstamp] phase cycles median:", np.median(d, axis=0).astype(int).tolist(), " total median", int(np.median(ts[:, 8].astype(np.int64) - ts[:, 0].astype(np.int64))),
                " p90", int(np.percentile(ts[:, 8].astype(np.int64) - ts[:, 0].astype(np.int64), 90)))
            if os.path.isdir("gpurun_out"): np.save("gpurun_out/tstamps.npy", ts)
            r = ts[:, 9:14].astype(np.int64)
            tot = r[:, 4] - r[:, 0]
            log("[tstamp] rerank_select per query (shader clocks): total median %d p90 %d max %d | rows wait %d  adc+filter %d  flush %d (medians)"
                % (np.median(tot), np.percentile(tot, 90), tot.max(), np.median(r[:, 1]), np.median(r[:, 2]), np.median(r[:, 3])))
            log("[tstamp] kernel span (first start .. last end): %d ; sum of per-query totals / 2048 wave slots: %d" % (r[:, 4].max() - r[:, 0].min(), tot.sum() // 2048))
    if os.environ.get("PQT_DBG_SWEEP"):
        # debug: stage times with parts of the kernels switched off (results wrong), same index, no rebuild
        for v in os.environ["PQT_DBG_SWEEP"].split(","):
            idx.set_option("debug_bits", int(v))
            for _ in range(6):
                step()
            barrier()
            h_ = idx.stage_ms_history(5).mean(0).tolist()
            log("[dbg-sweep] bits %s: traverse %.4f  rerank_select %.4f ms" % (v, h_[1], h_[3]))
        idx.set_option("debug_bits", 0)
        step()
        barrier()
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())

    ids_t = out_idx.to(torch.int64) & 0xffffffff
    r1, r10, r100 = recall_at(ids_t, gt, 1), recall_at(ids_t, gt, 10), recall_at(ids_t, gt, 100)
    ncand_mean = float(out_cnt.to(torch.int64).float().mean())
    cq = torch.quantile(out_cnt.to(torch.float32), torch.tensor([0.5, 0.9, 0.99, 1.0], device=dev)).tolist()
    log("[bench] candidates per query: median %.0f p90 %.0f p99 %.0f max %.0f" % tuple(cq))
    cand_local = st["candidates"]  # local candidates reranked on this rank in the last step
    bins_visited = st["bins_visited"] / max(1, st["queries"])

    # optional "next" row 8f-4 (not part of the timed path): exact re-rank of the k results against the raw uint8 vectors
    exact = None
    if args.extras and raw_u8 is not None and k <= 512:
        ri = torch.empty_like(out_idx)
        rd = torch.empty_like(out_dist)
        idx.rerank_exact_dev(queries, k, out_idx, raw_u8, ri, rd, stream=stream)
        torch.cuda.synchronize(dev)
        te = time.perf_counter()
        for _ in range(5):
            idx.rerank_exact_dev(queries, k, out_idx, raw_u8, ri, rd, stream=stream)
        torch.cuda.synchronize(dev)
        te = (time.perf_counter() - te) / 5 * 1e3
        rt = ri.to(torch.int64) & 0xffffffff
        exact = {"recall@1": recall_at(rt, gt, 1), "recall@10": recall_at(rt, gt, 10), "ms_per_batch": te}
    ms_per_step = elapsed / args.steps * 1e3
    units = qn * (world if mode == "replica" else 1)  # queries answered by the whole job per step
    qps = units * args.steps / elapsed
    if world > 1:  # job-wide recall / candidate statistics (outside the timed region)
        agg = torch.tensor([r1, r10, r100, ncand_mean], dtype=torch.float64, device=dev)
        dist.all_reduce(agg)
        r1, r10, r100, ncand_mean = (agg / world).tolist()

    # ---- roofline of the dominant kernel (largest mean launch duration over the timed steps) ---------------------
    # algorithmic bytes per launch (SURVEY.md 8d per-unit figures x the units one launch processes):
    #   traversal  (a1-a6): query 4D + L1virt out 4*LP*C1 + heuristic rows 16*Bb + bin probes 2*16*Bb
    #                       + candidate ids in/out 8*nCand                                   per query
    #   rerank+sel (a7-a8): code rows 4*LP*nCand + candidate ids 4*nCand + L1virt in 4*LP*C1 + results 8*k per query
    LP, C1 = w["LP"], w["C1"]
    fused_rs = args.k <= 128
    bytes_trav = qn * (4 * w["D"] + 4 * LP * C1 + 48 * bins_visited) + 8 * cand_local
    bytes_rs = cand_local * (4 * LP + 4 + (0 if fused_rs else 4)) + qn * (4 * LP * C1 + 8 * k)
    rs_name = ("pqt_k_rerank_select_wg" if 4 * LP * C1 * C1 > 65536 else "pqt_k_rerank_select") if fused_rs else "pqt_k_rerank"
    kern = {"traverse": ("pqt_k_traverse", bytes_trav), "rerank_select": (rs_name, bytes_rs)}
    dominant = max(("traverse", "rerank_select"), key=lambda n_: stage[n_])
    rr_name, rr_bytes = kern[dominant]
    rr_ms = float(stage[dominant])
    rr_gbs = rr_bytes / (rr_ms * 1e-3) / 1e9 if rr_ms > 0 else 0.0
    # whole-path algorithmic bytes per query: 4D + 8*Bb_visited + 4*nCand + 4*LP*nCand + 8k
    path_bytes_q = 4 * w["D"] + 8 * bins_visited + 4 * ncand_mean + 4 * LP * ncand_mean + 8 * k

    # HBM traffic of the dominant kernel from the committed PMC profile of this very command (profiles/pmc_latest.json,
    # produced by scripts/profile.sh: separate --pmc FETCH_SIZE / WRITE_SIZE passes; FETCH_SIZE factor from the
    # calibration in profiles/r01_pmc_calibration.json: 1.0 for 64-B code rows, 2.0 for 128-B rows)
    traffic = None
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
        if pm.get("workload") == args.workload and pm.get("bv") == args.bv and pm.get("bb") == args.bb and pm.get("k") == args.k and world == 1:
            ent = pm["kernels"].get(rr_name)
            if ent:
                traffic = (ent["FETCH_SIZE_KiB"] * pm["fetch_factor"] + ent["WRITE_SIZE_KiB"]) * 1024.0
    except Exception:
        traffic = None

    out = {
        "metric": "queries/sec + recall@1/@100, SIFT1M (1 GPU) and SIFT1B (8 GPUs)",
        "value": qps, "unit": "queries/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong" if mode == "shard_db" else "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": ("SIFT1M-shape synthetic" if args.workload in ("sift1m", "tiny") else "synthetic SIFT-shaped (chunk-built)") + ": N=%d d=%d p=%d c1=%d c2=%d w=%d lineparts=%d, batch=%d queries, "
                               "query(boundVectors=%d, boundBins=%d), k=%d" %
                               (n, w["D"], w["P"], w["C1"], w["C2"], w["W"], w["LP"], qn, args.bv, args.bb, k),
                   "parallelism": {"single": "1 GPU", "replica": "%d GPUs: index replicated, queries sharded (%d per rank per step), no data-path collective" % (world, qn),
                                   "shard_db": "%d GPUs: db range-sharded + one RCCL all-gather of per-shard top-k" % world}[mode],
                   "global_batch": units,
                   "recall@1": r1, "recall@10": r10, "recall@100": r100, "mean_candidates": ncand_mean,
                   "mean_bins_visited": bins_visited, "n_bins": meta["n_bins"], "max_bin": meta["max_bin"],
                   "exact_rerank_of_topk": exact,
                   "algorithmic_bytes_per_query": path_bytes_q,
                   "path_GBps": path_bytes_q * qps / 1e9, "path_frac_of_hbm_peak": path_bytes_q * qps / 1e9 / HBM_PEAK_GBS,
                   "stage_ms": stage, "dominant_kernel_by_time": rr_name},
        "roofline": {"bound": "hbm", "kernel": rr_name, "achieved": rr_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": rr_gbs / HBM_PEAK_GBS, "traffic": traffic, "avg_launch_ms": rr_ms,
                     "algorithmic_bytes_per_launch": rr_bytes,
                     "timing": "mean over the timed steps of the kernel's own duration: start/stop HIP events attached to the dispatch "
                               "(hipExtLaunchKernel) on the launch stream, read after the closing barrier",
                     "other_kernels": {kern[n_][0]: {"avg_launch_ms": float(stage[n_]), "algorithmic_bytes_per_launch": kern[n_][1],
                                                      "GBps": kern[n_][1] / max(stage[n_], 1e-9) / 1e6}
                                       for n_ in kern if n_ != dominant}},
    }

    # ---- measured stream bandwidth of this device (device-to-device copy of 1 GiB: read + write), reported beside the nominal
    # peak the fractions above are priced with
    if mode == "single":
        try:
            a_ = torch.empty(1 << 28, dtype=torch.float32, device=dev)
            b_ = torch.empty_like(a_)
            a_.fill_(1.0)
            for _ in range(2):
                b_.copy_(a_)
            e0_, e1_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0_.record()
            for _ in range(5):
                b_.copy_(a_)
            e1_.record()
            torch.cuda.synchronize(dev)
            gbs_ = 5 * 2 * a_.numel() * 4 / (e0_.elapsed_time(e1_) * 1e-3) / 1e9
            out["roofline"]["measured_stream_GBps"] = gbs_
            out["roofline"]["frac_of_measured_stream"] = rr_gbs / gbs_
            del a_, b_
            torch.cuda.empty_cache()
        except Exception as e:
            out["roofline"]["measured_stream_GBps"] = None

    # ---- second knob set of BASELINE.md (the CUDA library's defaults k1/maxBins: boundVectors = boundBins = 4096), short leg,
    # reported beside the headline (never as `value`)
    if args.extras and mode == "single" and (args.bv, args.bb) == (20000, 500) and not chunked:
        try:
            idx.build_heuristic(4096)
            for _ in range(2):
                idx.query_dev(queries, 4096, 4096, k, out_idx, out_dist, out_cnt, stream=stream)
            torch.cuda.synchronize(dev)
            t2 = time.perf_counter()
            for _ in range(5):
                idx.query_dev(queries, 4096, 4096, k, out_idx, out_dist, out_cnt, stream=stream)
            torch.cuda.synchronize(dev)
            t2 = (time.perf_counter() - t2) / 5
            i2 = out_idx.to(torch.int64) & 0xffffffff
            out["config"]["knobs_4096_4096"] = {"queries_per_sec": qn / t2, "ms_per_step": t2 * 1e3, "recall@1": recall_at(i2, gt, 1),
                                                "recall@100": recall_at(i2, gt, 100), "mean_candidates": float(out_cnt.float().mean()),
                                                "launch_structure": "fused traversal in wide mode (boundBins > 512: rows in blocks of 512, populated rows listed) + fused rerank/select"}
            idx.query_dev(queries, args.bv, args.bb, k, out_idx, out_dist, out_cnt, stream=stream)  # restore the headline outputs
            torch.cuda.synchronize(dev)
        except Exception as e:
            out["config"]["knobs_4096_4096"] = {"error": repr(e)[:200]}

    # ---- batch split over two handles on two streams (same index data): the traversal of one half overlaps the rerank of
    # the other.  Reported beside the headline (never as `value`: the per-kernel roofline accounting above is single-stream)
    if mode == "single" and not chunked and not os.environ.get("PQT_BENCH_NO_PIPELINE"):
        try:
            idx2 = pkg.PqtIndex(w["D"], w["P"], w["C1"], w["C2"], w["W"], w["LP"], device=dev.index or 0)
            idx2.set_codebooks(meta["cb1"], meta["cb2"])
            idx2.build_heuristic(max(args.bb, 1))
            idx2.set_bins(meta["bin_ids"], meta["sizes"], meta["members"])
            idx2.set_lines_dev(idx._keep[0], 0)
            half = qn // 2
            qa, qb = queries[:half].contiguous(), queries[half:].contiguous()
            oa = (torch.empty((half, k), dtype=torch.int32, device=dev), torch.empty((half, k), dtype=torch.float32, device=dev), torch.empty(half, dtype=torch.int32, device=dev))
            ob = (torch.empty((qn - half, k), dtype=torch.int32, device=dev), torch.empty((qn - half, k), dtype=torch.float32, device=dev), torch.empty(qn - half, dtype=torch.int32, device=dev))
            s2 = torch.cuda.Stream(dev)
            torch.cuda.synchronize(dev)

            def step2():
                idx.query_dev(qa, args.bv, args.bb, k, oa[0], oa[1], oa[2], stream=stream)
                idx2.query_dev(qb, args.bv, args.bb, k, ob[0], ob[1], ob[2], stream=s2.cuda_stream)
            for _ in range(3):
                step2()
            torch.cuda.synchronize(dev)
            t3 = time.perf_counter()
            for _ in range(args.steps):
                step2()
            torch.cuda.synchronize(dev)
            t3 = (time.perf_counter() - t3) / args.steps
            same = bool(torch.equal(torch.cat([oa[0], ob[0]]), out_idx) and torch.equal(torch.cat([oa[1], ob[1]]), out_dist))
            out["config"]["two_handles_two_streams"] = {"queries_per_sec": qn / t3, "ms_per_step": t3 * 1e3, "results_identical": same,
                                                        "what": "the same batch as two halves on two handles/streams: one half's traversal overlaps the other's rerank"}
            idx2.close()
            idx.query_dev(queries, args.bv, args.bb, k, out_idx, out_dist, out_cnt, stream=stream)
            torch.cuda.synchronize(dev)
        except Exception as e:
            out["config"]["two_handles_two_streams"] = {"error": repr(e)[:200]}

    # ---- CPU baseline (rank 0, N=1 only): the oracle restatement of cpu_version's query(), bounded sample ----------
    if mode == "single" and not args.no_cpu and not chunked:
        from oracle import Oracle
        o = Oracle(w["D"], w["P"], w["C1"], w["C2"], w["W"], w["LP"], heur_keep=1)
        o.set_heuristic(idx.heuristic(max(args.bb, 1)))
        o.set_codebooks(meta["cb1"], meta["cb2"])
        o.import_bins(meta["bin_ids"], meta["sizes"], meta["members"])
        codes_host = idx._keep[0].cpu().numpy().view(np.uint32)
        o.import_codes(codes_host)
        qh = queries.cpu().numpy()
        cores = usable_cores(o.max_threads())
        # one pass over the batch to size the sample, then as many passes as fit the budget (thread start-up and
        # scheduling noise need seconds, not milliseconds, of work to amortise)
        t = time.perf_counter()
        o.query_batch(qh, args.bv, args.bb, k, nthreads=cores)
        pass_t = time.perf_counter() - t
        reps = int(max(1, min(500, args.cpu_seconds / max(pass_t, 1e-6))))
        t = time.perf_counter()
        for _ in range(reps):
            ci, cd, cc = o.query_batch(qh, args.bv, args.bb, k, nthreads=cores)
        cpu_t = time.perf_counter() - t
        s1 = int(max(16, min(qn, 3.0 / max(pass_t * cores / qn, 1e-9))))
        t = time.perf_counter()
        o.query_batch(qh[:s1], args.bv, args.bb, k, nthreads=1)
        cpu1_t = time.perf_counter() - t
        # parity of the benchmark output itself: index lists identical to the checker's (its stable tie order is the
        # canonical one) on the first 1000 queries
        o.set_sort_mode(1)
        chk = min(1000, qn)
        ci, cd, cc = o.query_batch(qh[:chk], args.bv, args.bb, k, nthreads=cores)
        o.set_sort_mode(0)
        gi = out_idx[:chk].cpu().numpy().view(np.uint32)
        gd = out_dist[:chk].cpu().numpy()
        same = float(np.mean([np.array_equal(gi[i], ci[i]) and np.array_equal(gd[i].view(np.uint32), cd[i].view(np.uint32)) for i in range(chk)]))
        out["cpu_baseline"] = {"value": reps * qn / cpu_t, "unit": "queries/sec", "cores": cores, "kind": "port",
                               "sample": "%d passes over the %d bench queries (%.1f s of work), same index, all host threads (OpenMP over queries, one context per thread)"
                                         % (reps, qn, cpu_t),
                               "single_thread_qps": s1 / cpu1_t, "single_thread_ms_per_query": cpu1_t / s1 * 1e3,
                               "result_lists_identical_frac": same}
    # ---- north-star layout on the same job (short leg, never the headline value) ----------------------------------
    if mode == "replica":
        try:
            idx.close()
            del idx
            torch.cuda.empty_cache()
            sidx, sbase, smeta = build_index(pkg, w, local_rank, shard=sharding.shard_range(rank, world, n))
            del sbase
            sidx.build_heuristic(max(args.bb, 1))
            squeries = sift_like(qn, w["D"], 0xC0DE03, dev)  # the SAME batch on every rank
            sbuf = sharding.ShardBuffers(world, qn, k, dev)
            engine = sharding.PqtShardEngine(sidx)
            for _ in range(2):
                sharding.sharded_query(engine, dist, world, squeries, args.bv, args.bb, k, sbuf)
            barrier()
            ts = time.perf_counter()
            nstep = max(3, min(args.steps, 10))
            for _ in range(nstep):
                sharding.sharded_query(engine, dist, world, squeries, args.bv, args.bb, k, sbuf)
            barrier()
            tsh = torch.tensor([time.perf_counter() - ts], dtype=torch.float64, device=dev)
            dist.all_reduce(tsh, op=dist.ReduceOp.MAX)
            # every rank must hold the same merged result
            chk = sbuf.out_idx.to(torch.int64).sum().reshape(1)
            lo, hi = chk.clone(), chk.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            out["config"]["db_sharded"] = {"queries_per_sec": qn * nstep / float(tsh.item()), "ms_per_step": float(tsh.item()) / nstep * 1e3,
                                           "scaling": "strong", "ranks_agree": bool(lo.item() == hi.item()),
                                           "layout": "db range-sharded x%d, replicated traversal, one RCCL all-gather of [3][QN][k] words, exact merge" % world}
        except Exception as e:  # the headline measurement above stands on its own
            out["config"]["db_sharded"] = {"error": repr(e)[:300]}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
