#!/usr/bin/env python3
"""bench.py -- PQT query hot path on MI355X: queries/sec + recall, HBM roofline of the dominant kernel,
CPU baseline beside it.

One "step" = one pass of the whole hot path (distance tables -> traversal -> bin enumeration -> ADC line
rerank -> top-k) over one batch of QN synthetic SIFT-shaped queries, inputs and outputs resident in HBM.

    python bench.py                       # N=1: BASELINE.json configs[1] (SIFT1M shape, batch 10k) is `value`; the same line
                                          #      carries config.hbm_roofline_leg = configs[2] (100 M vectors) at both knob sets
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Multi-GPU (--gpus N > 1), one process per GPU.  ONE workload for the whole scaling sweep: for every N >= 2 the
database of BASELINE.json configs[2] (100 M vectors) is RANGE-SHARDED by vector id (every rank synthesises, assigns and
line-encodes only its own id range; the per-bin global populations come from one all-gather at build time); per batch the
traversal is sharded by QUERIES (rank r traverses QN/N of them, one all-gather of the per-query bin lists), every rank reranks
its own slice of the database, and the per-shard top-k lists are exchanged over RCCL by query slice (all-to-all), merged, and
the merged slices all-gathered => "scaling": "strong", value = QN*steps / time.  Because 100 M vectors also fit one GPU,
rank 0 afterwards builds the database whole and times the single-GPU path on the same queries: config.same_workload_1gpu is
the denominator and the top-level `scaling_vs_1gpu` the ratio of the run's own strong scaling.
With --gpus 8 `value` is BASELINE configs[3]'s size (1 B vectors, 125 M per GPU: the metric names SIFT1B on 8 GPUs) and the
100 M strong-scaling leg of the sweep rides beside it (config.strong_scaling_leg, `scaling_vs_1gpu`).  --workload overrides.
  * --traversal replicated: every rank traverses the whole batch (no second exchange; the protocol of rounds 1-2).
  * --replicas: queries are the units instead -- every rank holds the whole SIFT1M-shape index and answers its own
    batch, no data-path collective => "scaling": "weak", value = N*QN*steps / time (never the default: it does not
    exercise the exchange step).
--dataset-dir DIR: real data (sift_base/sift_learn/sift_query .fvecs|.bvecs + sift_groundtruth.ivecs as fetched by the
reference's scripts/prepare_data.sh): tree and database are built by the product's own front-end (tool_createdb: createTree +
buildKBestDB), recall@1/@10/@100 of the engine is reported beside the checker's on the same index.
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
    # BASELINE.json configs[3] size: 1 B vectors, range-sharded (16 GB of line codes per GPU at 8 GPUs)
    "synth1b": dict(D=128, P=4, C1=64, C2=64, W=1, LP=32, n_base=1_000_000_000, n_train=200_000, qn=10_000, chunk=4_000_000),
    # cfg3 shape at 1 M vectors: functional checks of the chunk-built / sharded path (not a bench line)
    "synth1m": dict(D=128, P=4, C1=64, C2=64, W=1, LP=32, n_base=1_000_000, n_train=100_000, qn=2_000, chunk=300_000),
    # BASELINE.json configs[4] shape (d=256 p=8 c1=128 c2=64, lineparts=32) at a single-GPU size.  THROUGHPUT ONLY, NO REFERENCE COUNTERPART:
    # the reference counts its heuristic rows in uint32, (W*C2)^P = 64^8 wraps to 0 and its query enumerates nothing at this shape (the
    # engine's default does the same: tests/test_gpu_parity.py::test_config5_shape...).  This workload lifts that limit (option
    # "enumerate_beyond_wrap") and supplies the first rows of the sum-of-squares order produced best-first (heuristic.py); bin ids keep
    # the reference's uint32 wrap-around ((C1*C2)^p = 2^13p: only parts 0..2 reach the bin id).
    "synth_cfg5": dict(D=256, P=8, C1=128, C2=64, W=1, LP=32, n_base=4_000_000, n_train=150_000, qn=10_000, chunk=1_000_000, beyond_wrap=True),
    "tiny_cfg5": dict(D=256, P=8, C1=128, C2=64, W=1, LP=32, n_base=200_000, n_train=40_000, qn=1_000, chunk=100_000, beyond_wrap=True),
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
CHUNK_SEED = 0xC0DE02  # chunk ci of the database is sift_like(m, D, CHUNK_SEED + 7919 * ci)


def chunk_ranges(w, lo, hi):
    """(chunk index, first id of the chunk, chunk length, [a, b) = part of the chunk inside [lo, hi)) for every chunk of the
    database that intersects [lo, hi).  The chunk grid is global, so a vector's value does not depend on the sharding."""
    n, chunk = w["n_base"], w.get("chunk", w["n_base"])
    out = []
    for ci in range(lo // chunk, (max(hi, lo + 1) - 1) // chunk + 1):
        s0 = ci * chunk
        m = min(chunk, n - s0)
        a, b = max(lo, s0), min(hi, s0 + m)
        if b > a:
            out.append((ci, s0, m, a, b))
    return out


def make_codebooks(w, dev, dist=None, world=1, force_collectives=False):
    """Tree of the benchmark index; rank 0's codebooks are broadcast (the k-means M step uses atomics: not bit-reproducible)."""
    D, P, C1, C2 = (w[k] for k in ("D", "P", "C1", "C2"))
    train = sift_like(w["n_train"], D, 0xC0DE01, dev)
    cb1, cb2 = train_codebooks(train, P, C1, C2, 0xC0DE04)
    del train
    if world > 1 or (force_collectives and dist is not None):
        t1, t2 = torch.from_numpy(cb1).to(dev), torch.from_numpy(cb2).to(dev)
        dist.broadcast(t1, 0)
        dist.broadcast(t2, 0)
        cb1, cb2 = t1.cpu().numpy(), t2.cpu().numpy()
    return cb1, cb2


def build_index(pkg, w, dev_index, shard=None, dist=None, world=1, rank=0, codebooks=None, force_collectives=False):
    """Synthesise the database chunk by chunk with the product's own build kernel (insert = id() + prepareReranking) and load
    it into a PqtIndex.  shard = (lo, hi): only that id range is generated, encoded and held (range shard built by the
    shard itself; the per-bin global counts come from ONE all-gather, sharding.global_bin_counts).
    Returns (index, base vectors on the device if the database is a single unsharded chunk else None, meta)."""
    dev = torch.device("cuda", dev_index)
    if torch.cuda.current_stream(dev).cuda_stream == 0:
        # called on torch's legacy default stream (tests, scripts): its handle is NULL, which the C-ABI reads as "the handle's
        # own non-blocking stream" -- not ordered with the data synthesis.  Build on an explicit stream instead.
        with torch.cuda.stream(torch.cuda.Stream(dev)):
            out = build_index(pkg, w, dev_index, shard, dist, world, rank, codebooks, force_collectives)
        torch.cuda.synchronize(dev)
        return out
    sharding = importlib.import_module("product-quantization-tree_amd.sharding")
    D, P, C1, C2, W, LP = (w[k] for k in ("D", "P", "C1", "C2", "W", "LP"))
    n = w["n_base"]
    lo, hi = shard if shard is not None else (0, n)
    t0 = time.time()
    cb1, cb2 = codebooks if codebooks is not None else make_codebooks(w, dev, dist, world, force_collectives)
    idx = pkg.PqtIndex(D, P, C1, C2, W, LP, device=dev_index)
    idx.set_codebooks(cb1, cb2)
    nl = hi - lo
    bins = torch.empty(nl, dtype=torch.int32, device=dev)
    codes = torch.empty((nl, LP), dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream  # same stream as the data synthesis
    t1 = time.time()
    base = None
    ranges = chunk_ranges(w, lo, hi)
    for ci, s0, m, a, b in ranges:
        x = sift_like(m, D, CHUNK_SEED + 7919 * ci, dev)
        idx.assign_encode_dev(x[a - s0:b - s0], bins[a - lo:b - lo], codes[a - lo:b - lo], stream=stream)
        if shard is None and len(ranges) == 1:
            base = x
        del x
    torch.cuda.synchronize(dev)
    t2 = time.time()
    # CSR by bin id (vector ids ascending inside a bin = the reference's insertion order)
    keys, counts, members = sharding.local_bin_lists(bins, lo)
    del bins
    meta = dict(cb1=cb1, cb2=cb2)
    if shard is None:
        bin_ids = keys.cpu().numpy().astype(np.uint32)
        sizes = counts.cpu().numpy().astype(np.uint32)
        mem = members.cpu().numpy().astype(np.uint32)
        del keys, counts, members
        torch.cuda.empty_cache()
        idx.set_bins(bin_ids, sizes, mem)
        idx.set_lines_dev(codes, 0)
        meta.update(n_bins=int(bin_ids.shape[0]), max_bin=int(sizes.max()), bin_ids=bin_ids, sizes=sizes, members=mem)
    else:
        uk, gs, low, ls = sharding.global_bin_counts(dist, world, rank, keys, counts, force_collectives)
        assert int(gs.max()) < 2 ** 32 and int(gs.sum()) == n, "global bin counts do not add up to the database size"
        idx.set_bins_local(uk.cpu().numpy(), gs.cpu().numpy(), low.cpu().numpy(), ls.cpu().numpy(), members.cpu().numpy(), n)
        idx.set_lines_dev(codes, lo)
        meta.update(n_bins=int(uk.numel()), max_bin=int(gs.max()), local_vectors=nl)
        del keys, counts, members
        torch.cuda.empty_cache()
    t3 = time.time()
    meta.update(t_data=t1 - t0, t_encode=t2 - t1, t_csr=t3 - t2)
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


def brute_force_gt_chunked(w, queries, dev, lo=0, hi=None, dist=None, world=1, force_collectives=False):
    """Exact nearest neighbour over the chunk-generated database (chunks are regenerated from their seeds); with a range
    shard every rank scans its own ids and the per-query minimum is all-reduced (ties -> lowest id)."""
    hi = w["n_base"] if hi is None else hi
    D = w["D"]
    best_d = torch.full((queries.shape[0],), float("inf"), device=dev)
    best_i = torch.full((queries.shape[0],), 2 ** 62, dtype=torch.int64, device=dev)
    qq = (queries * queries).sum(1)
    for ci, s0, m, a, b in chunk_ranges(w, lo, hi):
        x = sift_like(m, D, CHUNK_SEED + 7919 * ci, dev)[a - s0:b - s0]
        bn = (x * x).sum(1)
        for qa in range(0, queries.shape[0], 2048):
            q = queries[qa:qa + 2048]
            d = bn[None, :] - 2.0 * (q @ x.T)
            v, i = d.min(1)
            v = v + qq[qa:qa + 2048]
            upd = v < best_d[qa:qa + 2048]
            best_d[qa:qa + 2048] = torch.where(upd, v, best_d[qa:qa + 2048])
            best_i[qa:qa + 2048] = torch.where(upd, i + a, best_i[qa:qa + 2048])
        del x, bn
    if world > 1 or (force_collectives and dist is not None):
        gd = best_d.clone()
        dist.all_reduce(gd, op=dist.ReduceOp.MIN)
        best_i = torch.where(best_d == gd, best_i, torch.full_like(best_i, 2 ** 62))
        dist.all_reduce(best_i, op=dist.ReduceOp.MIN)
    return best_i


def recall_at(ids, gt0, r):
    r = min(r, ids.shape[1])
    return float((ids[:, :r] == gt0[:, None]).any(1).float().mean())


def kernel_bytes(w, qn, k, He, cand_local, fused_rs):
    """Algorithmic bytes per launch of the two kernels of the path: SURVEY.md 8(d) terms only.
         traversal  (a1-a6): 4*D (query) + 8*Bb (bin look-up: one (offset, count) pair per enumerated heuristic row)   per query
         rerank+sel (a7-a8): 4*nCand (id gather) + 4*LP*nCand (line codes) + 8*k (results)                             per query
       What the two-launch structure moves besides (NOT algorithmic: created by not fusing the two kernels) is returned as
       `intermediate`: L1virt[LP][C1] written by the traversal and read by the rerank, the candidate list written and read
       (+ candDist written and read when the staged k > 128 select runs)."""
    LP, C1, D = w["LP"], w["C1"], w["D"]
    trav = qn * (4 * D + 8 * He)
    rs = cand_local * (4 + 4 * LP) + qn * 8 * k
    inter_trav = qn * 4 * LP * C1 + 4 * cand_local
    inter_rs = qn * 4 * LP * C1 + 4 * cand_local + (0 if fused_rs else 8 * cand_local)
    return {"traverse": (trav, inter_trav), "rerank_select": (rs, inter_rs)}


def time_steps(step, barrier, warmup, steps, after_warmup=None):
    for _ in range(warmup):
        step()
    barrier()
    if after_warmup is not None:
        after_warmup()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    return time.perf_counter() - t0


STAGES = ("tables", "traverse", "gap", "rerank_select", "select")


class Ctx:
    """Process-wide state of one bench run: rank layout, device, the one stream everything is enqueued on."""
    pass


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no rank environment: re-execute the very same command line under
    torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1 at a free port).  Rank 0 prints the one JSON line; the
    launcher passes the ranks' stdout/stderr through.  The torchrun form of the docstring keeps working (WORLD_SIZE is set then)."""
    import socket
    have = torch.cuda.device_count()
    if have < args.gpus and not os.environ.get("PQT_BENCH_SAME_DEVICE"):
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible (PQT_BENCH_SAME_DEVICE=1 + PQT_BENCH_BACKEND=gloo runs all ranks on device 0: a functional check)"
                         % (args.gpus, have))
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), PQT_BENCH_SELF_LAUNCHED="1")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("[bench] --gpus %d without a rank environment: re-executing under torch.distributed.run (port %d)" % (args.gpus, port))
    sys.stdout.flush()
    sys.stderr.flush()
    os.execve(sys.executable, cmd, env)


def init_ctx(args):
    c = Ctx()
    c.world = int(os.environ.get("WORLD_SIZE", "1"))
    c.rank = int(os.environ.get("RANK", "0"))
    c.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != c.world and c.world == 1 and args.gpus > 1:
        raise SystemExit("internal: --gpus %d reached init_ctx without ranks (self_launch() should have re-executed the command)" % args.gpus)
    c.dist = None
    c.backend = None
    # PQT_BENCH_FORCE_SHARD=1: run the range-sharded layout with whatever world size there is -- with one rank this drives
    # every collective of the path (broadcast, build-time all-gather, all-to-all / all-gather per batch, all-reduce) through
    # RCCL on a 1-GPU box: an API/dtype check of the N > 1 code, not a measurement
    c.force_shard = bool(os.environ.get("PQT_BENCH_FORCE_SHARD"))
    if c.world > 1 or c.force_shard:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_PORT", "29655")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        c.backend = os.environ.get("PQT_BENCH_BACKEND", "nccl")  # "gloo" + PQT_BENCH_SAME_DEVICE=1: functional check on a 1-GPU box
        if os.environ.get("PQT_BENCH_SAME_DEVICE"):
            c.local_rank = 0
        torch.cuda.set_device(c.local_rank)
        if c.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", c.local_rank))
        else:
            dist.init_process_group(c.backend)
        c.dist = dist
    c.dev = torch.device("cuda", c.local_rank)
    torch.cuda.set_device(c.dev)
    # everything (data synthesis and the library's kernels) is enqueued on ONE explicit stream: the C-ABI treats a NULL
    # stream as "the handle's own non-blocking stream", which is not ordered with torch's legacy default stream
    c.work_stream = torch.cuda.Stream(c.dev)
    torch.cuda.set_stream(c.work_stream)
    c.stream = torch.cuda.current_stream(c.dev).cuda_stream
    c.pkg = importlib.import_module("product-quantization-tree_amd")
    c.pkg.lib()  # fails loudly if the HIP library is missing
    c.sharding = importlib.import_module("product-quantization-tree_amd.sharding")
    c.collectives = c.world > 1 or c.force_shard
    return c


def barrier(ctx):
    if ctx.collectives:
        ctx.dist.barrier()
    torch.cuda.synchronize(ctx.dev)


def set_heuristic_rows(ctx, idx, w, rows):
    """The traversal heuristic of a bench index: the reference's table (built by the library) -- or, for the configs[4]-shape workloads
    whose table cannot exist (2^48 rows; the reference enumerates none), the best-first prefix with the wrap limit lifted."""
    if w.get("beyond_wrap"):
        hp = importlib.import_module("product-quantization-tree_amd.heuristic")
        idx.set_option("enumerate_beyond_wrap", 1)
        idx.set_heuristic(hp.heuristic_prefix_best_first(w["W"] * w["C2"], w["P"], rows))
    else:
        idx.build_heuristic(rows)


def build_workload(ctx, args, wl_name, mode, want_gt=True, codebooks=None):
    """Index + query batch (+ exact ground truth) of one workload.  mode: single | replica | shard_db."""
    w = WORKLOADS[wl_name]
    n = w["n_base"]
    shard = ctx.sharding.shard_range(ctx.rank, ctx.world, n) if mode == "shard_db" else None
    idx, base, meta = build_index(ctx.pkg, w, ctx.local_rank, shard=shard, dist=ctx.dist, world=ctx.world if mode == "shard_db" else 1, rank=ctx.rank,
                                  codebooks=codebooks, force_collectives=ctx.force_shard)
    for ov in args.option:
        name_, val_ = ov.split("=")
        idx.set_option(name_, int(val_))
    t0 = time.time()
    set_heuristic_rows(ctx, idx, w, max(args.bb, 1))
    log("[bench] %s index: N=%d%s bins=%d max_bin=%d  data %.1fs encode %.1fs csr %.1fs heuristic %.1fs" %
        (wl_name, n, (" (this rank: ids [%d, %d))" % shard) if shard else "", meta["n_bins"], meta["max_bin"], meta["t_data"], meta["t_encode"],
         meta["t_csr"], time.time() - t0))
    # queries: fresh draws from the same mixture (like SIFT's separate query set), ground truth by exact brute force
    dev = ctx.dev
    g = torch.Generator(device=dev)
    g.manual_seed(0xC0DE03)
    qn = w["qn"]
    if args.query_mode == "perturbed" and base is not None:
        pick = torch.randint(0, n, (qn,), generator=g, device=dev)
        queries = (base[pick] + torch.randn(qn, w["D"], generator=g, device=dev) * 8.0).round().clamp_(0, 255).contiguous()
    else:
        queries = sift_like(qn, w["D"], 0xC0DE03 + (1000 * ctx.rank if mode == "replica" else 0), dev)
    if mode == "shard_db":
        ctx.dist.broadcast(queries, 0)  # the SAME batch on every rank
    raw_u8 = None
    if args.no_gt or not want_gt:
        gt = torch.full((qn,), -1, dtype=torch.int64, device=dev)
    elif base is not None:
        gt = brute_force_gt(base, queries, 1)[:, 0]
        raw_u8 = base.to(torch.uint8) if args.extras else None  # raw vectors for the optional exact re-rank (8f-4)
    else:
        lo_, hi_ = shard if shard else (0, n)
        gt = brute_force_gt_chunked(w, queries, dev, lo_, hi_, ctx.dist, ctx.world if mode == "shard_db" else 1, ctx.force_shard)
    del base
    torch.cuda.empty_cache()
    return dict(name=wl_name, w=w, n=n, shard=shard, idx=idx, meta=meta, queries=queries, gt=gt, have_gt=not (args.no_gt or not want_gt), raw_u8=raw_u8,
                chunked=w.get("chunk", n) < n, mode=mode, qn=qn)


def time_path(ctx, args, W, bv, bb, k, steps, warmup, period, pipeline=None, fresh=False, single_pipeline=False):
    """W warm-up steps, then exactly `steps` timed steps of the hot path between barriers (+ device synchronisation); the
    per-kernel HIP events ride on every period-th call.  Returns the timing, the stage means and the outputs.
    Range-sharded mode: pipeline = 2 (default, --pipeline) runs every step as two half batches in flight (sharding.py:
    sharded_query_pipelined -- half B's kernels under half A's collectives), 1 as one batch; the collectives of the event-carrying
    steps are bracketed by HIP events (exchange_ms).
    One GPU (single / replica), single_pipeline=True and --pipeline >= 2: two WHOLE batches in flight as well -- consecutive steps
    alternate between the index and a view of it (pqt_index_create_view: own scratch, same loaded index), each on its own stream with
    its own result arrays, so one batch's traversal runs under the other batch's rerank (the two launches of a batch depend on each
    other; the launches of different batches do not).  Every step is still one complete batch; all of them are complete at the closing
    barrier."""
    idx, queries, qn, dev, mode = W["idx"], W["queries"], W["qn"], ctx.dev, W["mode"]
    out_idx = torch.empty((qn, k), dtype=torch.int32, device=dev)
    out_dist = torch.empty((qn, k), dtype=torch.float32, device=dev)
    out_cnt = torch.empty(qn, dtype=torch.int32, device=dev)
    sbuf = engine = timer = view = None
    two_single = bool(single_pipeline and mode != "shard_db" and (args.pipeline if pipeline is None else pipeline) >= 2)
    pipeline = (args.pipeline if pipeline is None else pipeline) if (mode == "shard_db" and qn >= 2) else (2 if two_single else 1)
    slots = None
    if two_single:
        if "view" not in W:
            W["view"] = idx.view()
        view = W["view"]
        if "slot_stream" not in W:
            W["slot_stream"] = torch.cuda.Stream(dev)
        slots = [(idx, ctx.stream, (out_idx, out_dist, out_cnt)),
                 (view, W["slot_stream"].cuda_stream, (torch.empty_like(out_idx), torch.empty_like(out_dist), torch.empty_like(out_cnt)))]
    if mode == "shard_db":
        timer = ctx.sharding.ExchangeTimer(cuda=True)
        engine = ctx.sharding.PqtShardEngine(idx)
        inflight = None
        if pipeline >= 2:
            if "view" not in W:
                W["view"] = idx.view()
            view = W["view"]
            engines = (engine, ctx.sharding.PqtShardEngine(view))
            if pipeline == 2:
                inflight = ctx.sharding.BatchesInFlight(engines, ctx.world, qn, k, dev, bin_cap=ctx.sharding.bin_cap_for(bb))
                sbuf = inflight.bufs[0]
            else:
                sbuf = ctx.sharding.PipelineBuffers(ctx.world, qn, k, dev, bin_cap=ctx.sharding.bin_cap_for(bb))
        else:
            sbuf = ctx.sharding.ShardBuffers(ctx.world, qn, k, dev, bin_cap=ctx.sharding.bin_cap_for(bb))
    calls = [0]
    # fresh=True (single GPU): every step answers its OWN batch of queries (drawn before the timed region), so nothing a step reads --
    # candidate rows, bin-table entries, intermediates -- was put into a cache by the step before it
    qlist = None
    if fresh and mode != "shard_db":
        qlist = [sift_like(qn, W["w"]["D"], 0xF2E5A000 + 7 * i, dev) for i in range(warmup + steps)]
        torch.cuda.synchronize(dev)

    def step():
        if mode != "shard_db":
            qq = queries if qlist is None else qlist[calls[0] % len(qlist)]
            if slots is not None:
                h_, s_, o_ = slots[calls[0] & 1]
                calls[0] += 1
                h_.query_dev(qq, bv, bb, k, o_[0], o_[1], o_[2], stream=s_)
                return
            calls[0] += 1
            idx.query_dev(qq, bv, bb, k, out_idx, out_dist, out_cnt, stream=ctx.stream)
            return
        timer.on = calls[0] % period == 0  # the calls that carry the library's per-kernel events also carry the exchange events
        calls[0] += 1
        if pipeline == 2:
            inflight.step(ctx.dist, ctx.world, queries, bv, bb, k, exchange=args.exchange, force_collectives=ctx.force_shard, traversal=args.traversal, timer=timer)
        elif pipeline >= 3:
            ctx.sharding.sharded_query_pipelined(engines, ctx.dist, ctx.world, queries, bv, bb, k, sbuf, exchange=args.exchange, force_collectives=ctx.force_shard,
                                                 traversal=args.traversal, timer=timer)
        else:
            ctx.sharding.sharded_query(engine, ctx.dist, ctx.world, queries, bv, bb, k, sbuf, exchange=args.exchange, force_collectives=ctx.force_shard,
                                       traversal=args.traversal, timer=timer)

    # per-kernel HIP events on every P-th call (the first call after the option is set is a timed one): the timed region holds
    # the calls warmup .. warmup + steps - 1
    period = max(1, period)
    timed_in_region = [i for i in range(steps) if (warmup + i) % period == 0]
    if not timed_in_region:
        period, timed_in_region = 1, list(range(steps))
    # (two batches in flight on one device: each handle sees every other step and carries the events on every period-th of ITS calls, so
    # one step in `period` carries them as before)
    idx.set_option("stage_timing", period)
    if view is not None:
        view.set_option("stage_timing", period)
    def drop_warmup_spans():  # exchange_ms covers the timed region only
        if timer is not None:
            for v_ in timer.spans.values():
                v_.clear()
    elapsed = time_steps(step, lambda: barrier(ctx), warmup, steps, drop_warmup_spans)
    idx.set_option("stage_timing", 1)  # later legs time every call
    if mode == "shard_db" and pipeline == 2:
        inflight.wait()
        sbuf = inflight.bufs[inflight.last]
    if mode == "shard_db":
        out_idx.copy_(sbuf.out_idx)
        out_dist.copy_(sbuf.out_dist)
        out_cnt.copy_(sbuf.count)
    # per-stage device times of the timed steps themselves: the library records HIP events around every kernel on the
    # stream it launches on (ring of the last 32 calls); they are read only now, after the closing barrier.
    hist = idx.stage_ms_history(min(len(timed_in_region), 32))
    st = idx.stats()
    stage = dict(zip(STAGES, hist.mean(0).tolist())) if hist.shape[0] else dict.fromkeys(STAGES, 0.0)
    path = idx.last_path()
    view_slot_identical = None
    if two_single and qlist is None and steps >= 2:
        # the view slot's outputs are checked too -- outside the timed region, both slots answered the SAME batch
        o0, o1 = slots[0][2], slots[1][2]
        view_slot_identical = bool(torch.equal(o0[0], o1[0]) and torch.equal(o0[1], o1[1]) and torch.equal(o0[2], o1[2]))
    if two_single:
        # both slots answered the same batch (or, with fresh batches, their own): the view's kernels count like the index's
        nh_ = max(1, min(len(timed_in_region) // 2, 32))
        hist = idx.stage_ms_history(nh_)
        h2 = view.stage_ms_history(nh_)
        if h2.shape[0] and hist.shape[0]:
            both = np.concatenate([hist, h2], 0)
            stage = dict(zip(STAGES, both.mean(0).tolist()))
        path += " | two batches in flight on one device (index + view, two streams)"
        view.set_option("stage_timing", 0)
    elif view is not None:
        if pipeline >= 3:
            # two half batches: a kernel's time per step is the sum over the halves (they overlap in time); statistics likewise
            h2 = view.stage_ms_history(min(len(timed_in_region), 32))
            if h2.shape[0]:
                for n_, v_ in zip(STAGES, h2.mean(0).tolist()):
                    stage[n_] += v_
            st2 = view.stats()
            for n_ in ("queries", "candidates", "bins_visited", "bins_nonempty", "ties_l1", "ties_l2", "ties_bins", "ties_final", "filter_fallbacks"):
                st[n_] = st[n_] + st2[n_]
            path += " | two half batches in flight"
        else:
            path += " | two batches in flight"  # whole batches alternate between the index and its view: this handle's figures are one batch's
        view.set_option("stage_timing", 0)
    # shared-row pass: what its evaluating kernel read and wrote for one batch of this leg, counted on the device by ONE more call outside
    # the timed region (option "sr_stats": a statistics launch behind the pass's preparation; no stage events, so the history read above
    # stays the timed steps').  roofline_block prices pqt_k_sr_adc with these: the rows it reads once, not SURVEY 8(d)'s row per candidate.
    sr = None
    if "-shared" in path and mode != "shard_db":
        try:
            idx.set_option("stage_timing", 0)
            idx.set_option("sr_stats", 1)
            qq_ = queries if qlist is None else qlist[(calls[0] - 1) % len(qlist)]
            idx.query_dev(qq_, bv, bb, k, out_idx, out_dist, out_cnt, stream=ctx.stream)
            torch.cuda.synchronize(dev)
            sr = idx.shared_rows_stats()
            sr["candidates"] = int(idx.stats()["candidates"])
        except Exception as e:
            sr = {"error": repr(e)[:200]}
        finally:
            idx.set_option("sr_stats", 0)
            idx.set_option("stage_timing", 1)
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if ctx.collectives:
        ctx.dist.all_reduce(tmax, op=ctx.dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    units = qn * (ctx.world if mode == "replica" else 1)  # queries answered by the whole job per step
    exchange_ms = per_rank = None
    if mode == "shard_db":
        exchange_ms = timer.means_ms()
        mine = {"rank": ctx.rank, "stage_ms": stage, "exchange_ms": exchange_ms, "candidates": int(st["candidates"])}
        per_rank = [mine]
        if ctx.collectives:
            per_rank = [None] * ctx.world
            ctx.dist.all_gather_object(per_rank, mine)
    return dict(elapsed=elapsed, steps=steps, warmup=warmup, qps=units * steps / elapsed, ms_per_step=elapsed / steps * 1e3, units=units, stage=stage, st=st,
                out_idx=out_idx, out_dist=out_dist, out_cnt=out_cnt, sbuf=sbuf, step=step, period=period, n_timed=len(timed_in_region), bv=bv, bb=bb, k=k,
                path=path, exchange_ms=exchange_ms, per_rank=per_rank, pipeline=pipeline, view_slot_identical=view_slot_identical, sr=sr)


LIVE_TRAFFIC_BUDGET_S = [480.0]  # what is left for the child runs under rocprofv3 --pmc of this bench run (all legs together)


def live_traffic_wanted(ctx, args):
    """N = 1 on a GPU, not switched off (--no-live-traffic / PQT_BENCH_NO_LIVE_TRAFFIC: the test suite), not inside a profiler run."""
    return (ctx.world == 1 and not args.no_live_traffic and torch.cuda.is_available() and not os.environ.get("PQT_BENCH_NO_LIVE_TRAFFIC")
            and not any(e.startswith(("ROCPROF", "ROCP_")) for e in os.environ))


def live_pmc(args, wl_name, bv, bb, k, kernel, passes):
    """Per-dispatch means of hardware counters for `kernel`, collected NOW: this very command (same workload and knobs, 3 + 2 steps, no checker
    legs) is run as a child under `rocprofv3 --pmc <counters of one pass>` once per pass (--kernel-trace only: MI355X_MICROARCH.md's recipe).
    Returns (dict counter -> mean or None, dispatches, description).  Outside the timed region; bounded."""
    import csv, glob, re, shutil, subprocess, tempfile
    rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rp):
        return None, 0, "rocprofv3 not found"
    if LIVE_TRAFFIC_BUDGET_S[0] < 40.0:
        return None, 0, "the run's budget for live counter passes is spent"
    opts = list(args.option)
    if not any(o.startswith("overlap=") for o in opts):
        opts.append("overlap=0")  # every call in one piece: all dispatches of the kernel are full-size launches
    child = [sys.executable, os.path.abspath(__file__), "--no-cpu", "--no-gt", "--no-hbm-leg", "--no-live-traffic", "--pipeline", "1", "--steps", "3", "--warmup", "2",
             "--workload", wl_name, "--bv", str(bv), "--bb", str(bb), "--k", str(k), "--iso-noise", str(args.iso_noise), "--lat-noise", str(args.lat_noise),
             "--centers", str(args.centers), "--center-scale", str(args.center_scale), "--query-mode", args.query_mode]
    for o in opts:
        child += ["--option", o]
    env = dict(os.environ, TMPDIR="/tmp", PQT_BENCH_NO_PIPELINE="1")
    pat = re.compile(r"\b%s[<(]" % re.escape(kernel))
    means, disp, t_all = {}, 0, time.time()
    for ctrs in passes:
        vals = {c_: [] for c_ in ctrs}
        for attempt in (1, 2):  # (rocprofv3 --pmc occasionally hangs at process start on this image: bounded, one retry)
            d = tempfile.mkdtemp(prefix="pqt_pmc_", dir="/tmp")
            try:
                subprocess.run([rp, "--pmc"] + list(ctrs) + ["--kernel-trace", "--output-format", "csv", "-d", d, "-o", "t", "--"] + child, env=env, cwd="/tmp",
                               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=max(30.0, min(75.0 if WORKLOADS[wl_name]["n_base"] <= 10_000_000 else 150.0, LIVE_TRAFFIC_BUDGET_S[0] - (time.time() - t_all))))
                for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                    for r in csv.DictReader(open(f)):
                        if r["Counter_Name"] in vals and pat.search(r["Kernel_Name"]):
                            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
            except Exception as e:  # timeout, missing counters, unreadable output
                log("[bench] live counter pass %s attempt %d failed: %r" % (" ".join(ctrs), attempt, e))
            finally:
                shutil.rmtree(d, ignore_errors=True)
            if all(vals[c_] for c_ in ctrs):
                break
        if not all(vals[c_] for c_ in ctrs):
            LIVE_TRAFFIC_BUDGET_S[0] -= time.time() - t_all
            return None, 0, "rocprofv3 --pmc %s pass gave no dispatch of %s" % (" ".join(ctrs), kernel)
        for c_ in ctrs:
            means[c_], disp = sum(vals[c_]) / len(vals[c_]), len(vals[c_])
    LIVE_TRAFFIC_BUDGET_S[0] -= time.time() - t_all
    return means, disp, "%.0f s outside the timed region" % (time.time() - t_all)


def live_traffic(args, wl_name, bv, bb, k, kernel, fetch_factor):
    """HBM-side bytes of one launch of `kernel`: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (the two do not fit one), combined as
    FETCH * fetch_factor + WRITE with MI355X_MICROARCH.md's gfx950 corrections, the same as scripts/r04_profile.sh.  Returns (bytes or None, description)."""
    means, disp, why = live_pmc(args, wl_name, bv, bb, k, kernel, (("FETCH_SIZE",), ("WRITE_SIZE",)))
    if means is None:
        return None, why
    return (means["FETCH_SIZE"] * fetch_factor + means["WRITE_SIZE"]) * 1024.0, (
        "measured in this run: two child runs of this command under rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only), mean of %d "
        "dispatches, FETCH_SIZE x %.0f (gfx950 correction for this row width, profiles/r01_pmc_calibration.json) + WRITE_SIZE, KiB -> bytes; %s"
        % (disp, fetch_factor, why))


def live_issue(args, wl_name, bv, bb, k, kernel, launch_ms, n_cus):
    """where the kernel stands against the INSTRUCTION-ISSUE ceiling (the bound of the cache-resident configs[1]): instructions
    issued per launch by class (SQ_INSTS_*), per SIMD, against the launch's duration.  A SIMD with >= 4 wavefronts issues one instruction per
    ~4.4 engine clocks whatever the class (scripts/micro/valu_rate.hip, profiles/r04_valu_issue_rate.txt: 2.1 s_memtime ticks), i.e. a wave64
    instruction occupies its 16-lane SIMD for four clocks."""
    means, disp, why = live_pmc(args, wl_name, bv, bb, k, kernel, (("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM", "SQ_INSTS_BRANCH"),
                                                                   ("SQ_INSTS", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES")))
    if means is None:
        return {"error": why}
    simds = 4 * n_cus
    clk_ghz = 2.4  # MI355X_MICROARCH.md: max engine clock 2400 MHz (profiled runs measure 1.9 - 2.3 GHz effective: the fraction below is a lower bound)
    cycles = launch_ms * 1e-3 * clk_ghz * 1e9
    insts = means["SQ_INSTS"]
    return {"kernel": kernel, "instructions_per_launch": insts, "valu_share": means["SQ_INSTS_VALU"] / max(insts, 1.0),
            "by_class": {c_[9:].lower(): means[c_] for c_ in means if c_.startswith("SQ_INSTS_")},
            "other_share (s_nop, s_waitcnt, ...)": 1.0 - sum(means[c_] for c_ in means if c_.startswith("SQ_INSTS_")) / max(insts, 1.0),
            "instructions_per_simd": insts / simds, "launch_ms": launch_ms, "engine_clock_GHz": clk_ghz,
            "clocks_per_instruction_per_simd": cycles / max(insts / simds, 1.0),
            "frac_of_issue_ceiling": (insts / simds) * 4.0 / max(cycles, 1.0),
            "ceiling": "one wave64 instruction per 4 engine clocks per SIMD (16 lanes); with the clock the device REPORTS (it may run lower under load, which makes the fraction a lower bound)",
            "source": "two child runs of this command under rocprofv3 --pmc (SQ_INSTS_* classes; SQ_INSTS + cycles), mean of %d dispatches; %s" % (disp, why)}


def roofline_block(ctx, args, W, R, live=False):
    """`roofline` object of the dominant kernel (largest mean launch duration over the timed steps) + the whole-path figures."""
    w, qn, k, st, stage = W["w"], W["qn"], R["k"], R["st"], R["stage"]
    LP, C1 = w["LP"], w["C1"]
    cand_local = st["candidates"]  # local candidates reranked on this rank in the last step
    He = st["bins_visited"] / max(1, st["queries"])  # heuristic rows enumerated per query
    fused_rs = k <= 128
    kb = kernel_bytes(w, qn, k, He, cand_local, fused_rs)
    # big coarse tables: band-filtered exact rerank (pqt_k_rerank_select, MODE 2) unless --option exact_filter=0 (workgroup kernel)
    rs_name = ("pqt_k_rerank_select_wg" if (4 * LP * C1 * C1 > 65536 and "exact_filter=0" in args.option) else "pqt_k_rerank_select") if fused_rs else \
              ("pqt_k_rerank_select_big" if k <= 4096 else "pqt_k_rerank")
    shared = "-shared" in R.get("path", "")
    if shared:
        # shared-row pass (pqt_shared_rows.h): stage "rerank_select" = its preparation kernels + pqt_k_sr_adc (the rows, each read once for all
        # the queries of the batch that include its bin), stage "select" = pqt_k_sr_select over the distances it wrote.  The SURVEY 8(d) bytes of
        # the rerank (one code row per candidate and query) are priced on the kernel that stands for them; what it really moves is `traffic`.
        rs_name = "pqt_k_sr_adc"
    kname = {"traverse": "pqt_k_traverse", "rerank_select": rs_name}
    dominant = max(("traverse", "rerank_select"), key=lambda n_: stage[n_])
    rr_name = kname[dominant]
    rr_bytes, rr_inter = kb[dominant]
    rr_ms = float(stage[dominant])
    # whole-path algorithmic bytes per query: 4D + 8*Bb_visited + 4*nCand + 4*LP*nCand + 8k (this rank's candidates)
    ncand_rank = cand_local / max(1, qn)
    path_bytes_q = 4 * w["D"] + 8 * He + 4 * ncand_rank + 4 * LP * ncand_rank + 8 * k
    # Shared-row pass: SURVEY 8(d) prices a code row per candidate and query; pqt_k_sr_adc evaluates a (query, bin) pair once
    # whatever the number of visits and reads a bin's rows once for up to 8 queries, so 8(d)'s bytes are NOT the units this launch processes
    # (pricing them gave frac 1.7 > 1).  What one launch must move at the least: every DISTINCT row of the batch once (code row + its bias
    # word: 4*LP + 4 bytes) and one 4-byte filter distance per candidate written -- counted on the device in this run (time_path: sr_stats).
    # `achieved` / `frac` price those; the 8(d) figure stays in the line as a speed-up over the per-candidate formulation, not as a fraction.
    dedup = None
    sr = R.get("sr") or {}
    if shared and dominant == "rerank_select" and "distinct_rows" in sr:
        once = sr["distinct_rows"] * (4 * LP + 4) + 4 * sr["distances_written"]
        sel_bytes = 4 * sr["distances_written"] + qn * (8 * 256 * 2 + 8 * k)  # selection: the distances read back, the <= 256 best keys out and in, results
        dedup = {"distinct_rows": sr["distinct_rows"], "rows_read_by_the_kernel": sr["rows_read"], "pairs_query_bin": sr["pairs"], "bins": sr["bins"], "items": sr["items"],
                 "distances_written": sr["distances_written"], "candidates": sr.get("candidates"), "uncovered_queries": sr["uncovered_queries"], "capacity_flag": sr["capacity_flag"],
                 "bytes_per_launch": once, "survey_8d_bytes_per_launch": rr_bytes,
                 "algorithmic_equivalent_GBps": rr_bytes / (rr_ms * 1e-3) / 1e9 if rr_ms > 0 else 0.0,
                 "algorithmic_equivalent_speedup": rr_bytes / max(once, 1),
                 "what": "bytes_per_launch = distinct_rows * (4*LP + 4) + 4 * distances_written, counted on the device for one batch of this leg "
                         "(pqt_get_shared_rows_stats); algorithmic_equivalent_GBps = SURVEY 8(d)'s per-candidate bytes over the same launch time: the rate a "
                         "row-per-candidate kernel would have to sustain to match it (a speed-up, may exceed the HBM peak, not a roofline fraction)"}
        rr_bytes = once
        # the whole step moves, at the least: the traversal's bytes, the pass's, the selection's
        path_bytes_q = (kb["traverse"][0] + once + sel_bytes) / max(1, qn)
    rr_gbs = rr_bytes / (rr_ms * 1e-3) / 1e9 if rr_ms > 0 else 0.0
    shard = W["shard"]
    n_local = (shard[1] - shard[0]) if shard else W["n"]
    store_bytes = n_local * LP * 4
    resident = "infinity_cache" if store_bytes < (256 << 20) else "hbm"
    # HBM traffic of the dominant kernel: NOT measured in this run -- replayed from the committed PMC profile of this very command
    # (profiles/pmc_latest.json, produced by scripts/r03_profile.sh: separate --pmc FETCH_SIZE / WRITE_SIZE passes; FETCH_SIZE
    # factor from the calibration in profiles/r01_pmc_calibration.json: 1.0 for 64-B code rows, 2.0 for 128-B rows)
    traffic = None
    traffic_source = None
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
        pms = next((r_ for r_ in pm.get("runs", []) if r_.get("workload") == W["name"] and r_.get("bv") == R["bv"] and r_.get("bb") == R["bb"] and r_.get("k") == k), {})
        if pms and ctx.world == 1 and not args.option:
            ent = pms["kernels"].get(rr_name)
            if ent:
                traffic = (ent["FETCH_SIZE_KiB"] * pms["fetch_factor"] + ent["WRITE_SIZE_KiB"]) * 1024.0
                traffic_source = "committed profile (profiles/pmc_latest.json: %s); not collected in this run" % pms.get("source", "rocprofv3 --pmc passes")
    except Exception:
        traffic = None
    traffic_committed = traffic
    if live:
        # default N = 1 line: collected live, so that a change of the kernel's memory behaviour shows in the driver's line; the committed profile's
        # figure rides along (roofline.traffic_committed_profile) and stands in only when the collection fails
        lt, why = live_traffic(args, W["name"], R["bv"], R["bb"], k, rr_name, 2.0 if LP * 4 >= 128 else 1.0)
        if lt is not None:
            traffic, traffic_source = lt, why
        else:
            traffic_source = "live collection failed (%s); %s" % (why, traffic_source or "no committed profile of this command")
    issue = None
    if live and resident != "hbm":
        try:
            issue = live_issue(args, W["name"], R["bv"], R["bb"], k, rr_name, rr_ms, 256)
        except Exception as e:
            issue = {"error": repr(e)[:200]}
    roof = {"bound": "hbm" if resident == "hbm" else "issue", "kernel": rr_name, "achieved": rr_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": rr_gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source, "traffic_committed_profile": traffic_committed,
            "traffic_ratio": (traffic / rr_bytes) if (traffic and rr_bytes) else None,
            "resident": resident, "line_store_bytes": store_bytes,
            "avg_launch_ms": rr_ms,
            "algorithmic_bytes_per_launch": rr_bytes,
            "intermediate_bytes": rr_inter,
            "accounting": "SURVEY 8(d) terms only -- rerank+select: 4*nCand + 4*LP*nCand + 8k per query; traversal: 4*D + 8*Bb per query; "
                          "intermediate_bytes (L1virt and candidate-list round trips between the two launches) are listed, not priced",
            "timing": "mean of the kernel's own duration over the %d of the %d timed steps that carry events: start/stop HIP events attached to "
                      "the dispatch (hipExtLaunchKernel) on the launch stream, read after the closing barrier" % (R["n_timed"], R["steps"]),
            "other_kernels": {kname[n_]: {"avg_launch_ms": float(stage[n_]), "algorithmic_bytes_per_launch": kb[n_][0], "intermediate_bytes": kb[n_][1],
                                          "GBps": kb[n_][0] / max(stage[n_], 1e-9) / 1e6, "frac": kb[n_][0] / max(stage[n_], 1e-9) / 1e6 / HBM_PEAK_GBS}
                              for n_ in kname if n_ != dominant}}
    if issue is not None:
        roof["issue"] = issue
    if shared:
        sel_ms = float(stage["select"])
        roof["selection_kernel"] = {"kernel": "pqt_k_sr_select (scan + band launches)", "avg_launch_ms": sel_ms,
                                    "what": "the wave-per-query selection over the 4-byte filter distances pqt_k_sr_adc wrote"}
        if dedup is not None:
            roof["selection_kernel"].update({"bytes_per_launch": sel_bytes, "GBps": sel_bytes / max(sel_ms, 1e-9) / 1e6, "frac": sel_bytes / max(sel_ms, 1e-9) / 1e6 / HBM_PEAK_GBS,
                                             "bytes_are": "4 * distances read + 2 * 8 * 256 best keys per query (scan -> band through HBM) + 8k results per query"})
            roof["deduplicated"] = dedup
            roof["accounting"] = ("shared-row pass: `achieved` / `frac` price what ONE launch of pqt_k_sr_adc (+ its four preparation launches, stage rerank_select) must move at the "
                                  "least -- distinct rows * (4*LP + 4) + 4 bytes per filter distance written, counted on the device in this run (roofline.deduplicated) -- because the "
                                  "kernel reads a bin's rows once for all the queries and visits of the batch; SURVEY 8(d)'s per-candidate bytes are kept as "
                                  "deduplicated.algorithmic_equivalent_GBps (a speed-up over the per-candidate formulation, not a fraction)")
        else:
            roof["accounting"] += ("; shared-row pass WITHOUT device statistics (%r): `achieved` prices SURVEY 8(d)'s per-candidate bytes on a kernel that reads a row once -- not a "
                                   "roofline fraction when above 1" % (sr.get("error"),))
    if resident == "infinity_cache":
        roof["bound_note"] = ("configs[1]: the line store is cache resident and the dominant kernel runs at the instruction-issue ceiling (roofline.issue, "
                              "profiles/r04_cfg2_sift1m_sq_inst_mix.txt); `frac` is the contract's HBM figure, not the bound")
        roof["note"] = ("the %d MB line store of this workload stays in the 256 MiB Infinity Cache across batches: `frac` is priced against the HBM "
                        "peak but is not an HBM-bound result; the HBM-roofline configuration (BASELINE configs[2], 100 M vectors) is config.hbm_roofline_leg" % (store_bytes >> 20))
    extra = dict(He=He, cand_local=cand_local, ncand_rank=ncand_rank, path_bytes_q=path_bytes_q, n_local=n_local, dominant=rr_name)
    return roof, extra


def recalls(W, out_idx):
    if not W["have_gt"]:
        return None, None, None
    ids_t = out_idx.to(torch.int64) & 0xffffffff
    return recall_at(ids_t, W["gt"], 1), recall_at(ids_t, W["gt"], 10), recall_at(ids_t, W["gt"], 100)


def make_line(ctx, args, W, R):
    """The JSON line of one timed workload."""
    w, qn, n, mode, k = W["w"], W["qn"], W["n"], W["mode"], R["k"]
    world, dev, dist = ctx.world, ctx.dev, ctx.dist
    r1, r10, r100 = recalls(W, R["out_idx"])
    out_cnt = R["out_cnt"]
    ncand_mean = float(out_cnt.to(torch.int64).float().mean())  # GLOBAL candidates per query (all shards)
    cq = torch.quantile(out_cnt.to(torch.float32), torch.tensor([0.5, 0.9, 0.99, 1.0], device=dev)).tolist()
    log("[bench] %s candidates per query: median %.0f p90 %.0f p99 %.0f max %.0f   path: %s" % ((W["name"],) + tuple(cq) + (R["path"],)))
    if mode == "replica" and W["have_gt"]:  # job-wide recall / candidate statistics (outside the timed region)
        agg = torch.tensor([r1, r10, r100, ncand_mean], dtype=torch.float64, device=dev)
        dist.all_reduce(agg)
        r1, r10, r100, ncand_mean = (agg / world).tolist()
    roof, ex = roofline_block(ctx, args, W, R, live=live_traffic_wanted(ctx, args))
    meta, st, stage = W["meta"], R["st"], R["stage"]
    shard_par = "%d GPUs: db range-sharded by vector id (%d vectors per rank, built by the rank itself), %s, %s" % (
        world, ex["n_local"],
        "traversal sharded by queries (QN/W per rank; one all-gather of the per-query bin lists, <= %d x 8 B per query)" % ctx.sharding.BIN_CAP
        if args.traversal == "sharded" else "traversal replicated",
        "per-shard top-k exchanged by query slice (all-to-all of [3][QN/W][k] words per peer), exact (dist,pos) merge of the own slice, all-gather of the merged "
        "[2][QN/W][k] slices" if args.exchange == "alltoall" else "ONE all-gather of per-shard top-k [3][QN][k] words per batch + exact (dist,pos) merge of all queries on every rank")
    out = {
        "metric": "queries/sec + recall@1/@100, SIFT1M (1 GPU) and SIFT1B (8 GPUs)",
        "value": R["qps"], "unit": "queries/sec", "n_gpus": world, "steps": R["steps"], "warmup": R["warmup"],
        "ms_per_step": R["ms_per_step"], "higher_is_better": True,
        # the sweep the driver runs (N = 1, 2, 4, 8) is a STRONG-scaling design: for N >= 2 the total work (one database, one batch)
        # is fixed; --replicas is the weak-scaling variant
        "scaling": "weak" if mode == "replica" else "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": ("SIFT1M-shape synthetic" if W["name"] in ("sift1m", "tiny") else
                                ("configs[4]-shape synthetic (chunk-built), THROUGHPUT ONLY -- no reference counterpart: the reference enumerates 0 heuristic rows at this shape "
                                 "((W*C2)^P wraps to 0 in uint32); limit lifted, best-first sum-of-squares prefix supplied" if w.get("beyond_wrap") else
                                 "synthetic SIFT-shaped (chunk-built)")) + ": N=%d d=%d p=%d c1=%d c2=%d w=%d lineparts=%d, batch=%d queries, "
                               "query(boundVectors=%d, boundBins=%d), k=%d" %
                               (n, w["D"], w["P"], w["C1"], w["C2"], w["W"], w["LP"], qn, R["bv"], R["bb"], k),
                   "workload_name": W["name"],
                   "parallelism": {"single": "1 GPU", "replica": "%d GPUs: index replicated, queries sharded (%d per rank per step), no data-path collective" % (world, qn),
                                   "shard_db": shard_par}[mode],
                   "exchange": args.exchange if mode == "shard_db" else None,
                   "traversal": args.traversal if mode == "shard_db" else None,
                   # range-sharded run: mean device time of each collective of a step (HIP events on the issuing stream around the call, on the
                   # steps that also carry the per-kernel events; with two half batches in flight: per half-batch call), this rank and all ranks
                   "pipeline": {1: "one batch at a time", 2: "two whole batches in flight (consecutive steps alternate between the index and a view of it on two streams: one batch's collectives pass under the other's kernels)",
                                3: "two half batches in flight (half B's kernels run under half A's collectives)"}[R["pipeline"]] if mode == "shard_db" else
                               {1: "one batch at a time on one stream",
                                2: "two whole batches in flight (consecutive steps alternate between the index and a view of it on two streams: one batch's traversal "
                                   "runs under the other batch's rerank; every step is one complete batch, all complete at the closing barrier; "
                                   "config.one_batch_at_a_time = the same steps issued one after the other)"}[R["pipeline"]],
                   "exchange_ms": R["exchange_ms"],
                   "per_rank_stage_ms": R["per_rank"],
                   "collective_backend": ({"nccl": "rccl"}.get(ctx.backend, ctx.backend) if ctx.collectives else None), "collective_world_size": world,
                   "options": args.option,
                   "kernel_timing": "per-kernel start/stop HIP events on %d of the %d timed steps (every %s call%s; the events cost ~10 us per call)"
                                    % (R["n_timed"], R["steps"], {1: "", 2: "2nd"}.get(R["period"], "%dth" % R["period"]),
                                       " of each of the two handles" if (mode != "shard_db" and R["pipeline"] == 2) else ""),
                   "kernel_path": R["path"],
                   "global_batch": R["units"],
                   "recall@1": r1, "recall@10": r10, "recall@100": r100, "mean_candidates": ncand_mean,
                   "mean_candidates_this_rank": ex["ncand_rank"],
                   "mean_bins_visited": ex["He"], "n_bins": meta["n_bins"], "max_bin": meta["max_bin"], "filter_fallbacks": st.get("filter_fallbacks"),
                   "algorithmic_bytes_per_query": ex["path_bytes_q"],
                   "path_GBps": ex["path_bytes_q"] * qn * R["steps"] / R["elapsed"] / 1e9,
                   "path_frac_of_hbm_peak": ex["path_bytes_q"] * qn * R["steps"] / R["elapsed"] / 1e9 / HBM_PEAK_GBS,
                   "device_bytes": W["idx"].device_bytes(),  # this rank's handle: the line store and its derived copies, tables, scratch
                   "stage_ms": stage, "dominant_kernel_by_time": ex["dominant"], "build_s": {k_: meta[k_] for k_ in ("t_data", "t_encode", "t_csr")}},
        "roofline": roof,
    }
    return out


def same_workload_1gpu(ctx, args, W, R):
    """Range-sharded run: the SAME database on ONE GPU (rank 0 builds it whole and times the single-GPU path while the others
    wait), so the line carries the denominator of its own strong-scaling ratio."""
    w, n, qn, k, dev = W["w"], W["n"], W["qn"], R["k"], ctx.dev
    ref1 = None
    try:
        if not args.no_ref1 and n <= 200_000_000:
            if ctx.rank == 0:
                ridx, _, rmeta = build_index(ctx.pkg, w, ctx.local_rank, codebooks=(W["meta"]["cb1"], W["meta"]["cb2"]))
                set_heuristic_rows(ctx, ridx, w, max(R["bb"], 1))
                for ov in args.option:
                    ridx.set_option(ov.split("=")[0], int(ov.split("=")[1]))
                ro = (torch.empty((qn, k), dtype=torch.int32, device=dev), torch.empty((qn, k), dtype=torch.float32, device=dev), torch.empty(qn, dtype=torch.int32, device=dev))
                nst = max(3, min(R["steps"], 10))
                t1g = time_steps(lambda: ridx.query_dev(W["queries"], R["bv"], R["bb"], k, ro[0], ro[1], ro[2], stream=ctx.stream),
                                 lambda: torch.cuda.synchronize(dev), 2, nst)
                same = bool(torch.equal(ro[0], R["out_idx"]) and torch.equal(ro[1], R["out_dist"]) and torch.equal(ro[2], R["out_cnt"]))
                h1 = ridx.stage_ms_history(nst).mean(0).tolist()
                ref1 = {"queries_per_sec": qn * nst / t1g, "ms_per_step": t1g / nst * 1e3, "results_identical_to_sharded": same,
                        "stage_ms": dict(zip(STAGES, h1)), "kernel_path": ridx.last_path(),
                        "speedup_of_this_run": R["qps"] / (qn * nst / t1g)}
                ridx.close()
                del ridx, ro
                torch.cuda.empty_cache()
            ctx.dist.barrier()
    except Exception as e:
        ref1 = {"error": repr(e)[:300]}
    return ref1


def ranks_agree(ctx, R):
    chk = (R["sbuf"].out_idx.to(torch.int64) & 0xffffffff).sum().reshape(1)
    lo_c, hi_c = chk.clone(), chk.clone()
    ctx.dist.all_reduce(lo_c, op=ctx.dist.ReduceOp.MIN)
    ctx.dist.all_reduce(hi_c, op=ctx.dist.ReduceOp.MAX)
    return bool(lo_c.item() == hi_c.item())


def knob_leg(ctx, args, W, bv, bb, k, steps, warmup, live=False, fresh=False):
    """One knob set on an index that is already built: q/s, stage ms, roofline of its dominant kernel.
    fresh=True (the hbm_roofline_leg): the leg's figures are those of steps that each answer their OWN batch of queries (nothing a
    step reads was cached by the step before); the same-batch-every-step figures ride along as `same_batch_every_step`."""
    set_heuristic_rows(ctx, W["idx"], W["w"], max(bb, 1))
    Rw = time_path(ctx, args, W, bv, bb, k, steps, warmup, period=2)
    R = time_path(ctx, args, W, bv, bb, k, steps, warmup, period=2, fresh=True) if fresh else Rw
    roof, ex = roofline_block(ctx, args, W, R, live=live)
    r1, r10, r100 = recalls(W, Rw["out_idx"])
    leg = {"query": "query(boundVectors=%d, boundBins=%d), k=%d" % (bv, bb, k), "queries_per_sec": R["qps"], "ms_per_step": R["ms_per_step"], "steps": steps, "warmup": warmup,
           "batches": "a fresh batch of queries every step" if fresh else "the same batch every step",
           "stage_ms": R["stage"], "kernel_path": R["path"], "mean_candidates": float(Rw["out_cnt"].to(torch.int64).float().mean()),
           "device_bytes": W["idx"].device_bytes(),
           "mean_bins_visited": ex["He"], "filter_fallbacks": R["st"].get("filter_fallbacks"),
           "recall@1": r1, "recall@100": r100,
           "algorithmic_bytes_per_query": ex["path_bytes_q"], "path_frac_of_hbm_peak": ex["path_bytes_q"] * W["qn"] * steps / R["elapsed"] / 1e9 / HBM_PEAK_GBS,
           "roofline": {k_: roof[k_] for k_ in ("bound", "kernel", "achieved", "peak", "unit", "frac", "avg_launch_ms", "algorithmic_bytes_per_launch", "traffic", "traffic_source", "traffic_committed_profile",
                                                 "traffic_ratio", "resident", "line_store_bytes", "other_kernels", "selection_kernel", "deduplicated", "accounting") if k_ in roof}}
    if fresh:
        leg["same_batch_every_step"] = {"queries_per_sec": Rw["qps"], "ms_per_step": Rw["ms_per_step"], "stage_ms": Rw["stage"], "vs_fresh": Rw["qps"] / R["qps"]}
        # what a step spends outside its kernels' own durations (launch gaps, the handed-back queries' kernels)
        leg["ms_per_step_outside_the_stage_kernels"] = R["ms_per_step"] - sum(R["stage"].values())
    leg["_Rfresh"] = R if fresh else None
    return leg, Rw


def dram_side_figures(ctx, args, W, bv, bb, k, steps, kleg):
    """How much of the dominant launch's algorithmic rate can be DRAM traffic (FETCH_SIZE counts Infinity-Cache hits,
    and the bench re-issues one batch).  (1) the same knob set with a FRESH batch of queries every step; (2) the distinct database rows one
    batch touches: every candidate of every query (pqt_query_candidates: the whole lists) -> distinct vector ids -> unique_row_bytes =
    distinct * 4 * LP, reuse_factor = candidates / distinct.  A launch must bring at least the distinct rows that no cache held before it:
    with fresh batches and distinct rows far beyond the 256 MiB Infinity Cache that is (nearly) all of them, so
    dram_GBps_lower_bound = unique_row_bytes / launch time; the repeated touches of a row inside a launch are what caches can serve."""
    w, qn, dev, idx = W["w"], W["qn"], ctx.dev, W["idx"]
    Rf = kleg.pop("_Rfresh", None) or time_path(ctx, args, W, bv, bb, k, steps, 2, period=2, fresh=True)
    same_qps = kleg.get("same_batch_every_step", {}).get("queries_per_sec", kleg["queries_per_sec"])
    # distinct rows of the standard batch
    cap = int(min(bv + W["meta"]["max_bin"] + 64, 2 ** 31 - 1))
    oi = torch.empty((qn, cap), dtype=torch.int32, device=dev)
    od = torch.empty((qn, cap), dtype=torch.float32, device=dev)
    oc = torch.empty(qn, dtype=torch.int32, device=dev)
    idx.query_candidates_dev(W["queries"], bv, bb, cap, oi, od, oc, stream=ctx.stream, sync=True)
    del od
    valid = torch.arange(cap, device=dev)[None, :] < oc[:, None].clamp(max=cap)
    ids = oi[valid]
    total = int(ids.numel())
    distinct = int(torch.unique(ids).numel())
    del oi, valid, ids
    torch.cuda.empty_cache()
    row_bytes = 4 * w["LP"] + 4  # code row + its id
    launch_ms_fresh = float(Rf["stage"]["rerank_select"])  # (shared-row pass: preparation + pqt_k_sr_adc, the kernel that reads the rows)
    out = {"fresh_queries_per_step": {"queries_per_sec": Rf["qps"], "ms_per_step": Rf["ms_per_step"], "stage_ms": Rf["stage"],
                                      "vs_same_batch_every_step": Rf["qps"] / same_qps},
           "candidates_per_batch": total, "distinct_rows_per_batch": distinct, "reuse_factor": total / max(distinct, 1),
           "unique_row_bytes": distinct * row_bytes, "algorithmic_row_bytes": total * row_bytes,
           "infinity_cache_bytes": 256 << 20,
           "dram_GBps_lower_bound": distinct * row_bytes / max(launch_ms_fresh, 1e-9) / 1e6,
           "dram_frac_of_hbm_peak_lower_bound": distinct * row_bytes / max(launch_ms_fresh, 1e-9) / 1e6 / HBM_PEAK_GBS,
           "what": "fresh batch per step: nothing a launch reads was cached by the launch before; distinct rows x row bytes must come from DRAM when they exceed the "
                   "Infinity Cache many times over (lower bound of the DRAM read rate of the rerank launch); achieved / frac above price ALL touches, repeated ones included"}
    return out


def hbm_roofline_leg(ctx, args):
    """BASELINE configs[2] inside the default N = 1 command: 100 M vectors (12.8 GB of line codes: HBM resident), 10 k queries, the
    reference-default knobs and the CUDA library's (4096, 4096).  `value` stays the SIFT1M-shape number."""
    t0 = time.time()
    W = build_workload(ctx, args, args.hbm_workload, "single", want_gt=not args.no_gt)
    leg = {"workload": "BASELINE configs[2]: synthetic SIFT-shaped (chunk-built) N=%d d=128 p=4 c1=64 c2=64 w=1 lineparts=32, batch=%d queries, 1 GPU" % (W["n"], W["qn"]),
           "workload_name": W["name"], "n_bins": W["meta"]["n_bins"], "max_bin": W["meta"]["max_bin"],
           "build_s": {k_: W["meta"][k_] for k_ in ("t_data", "t_encode", "t_csr")}}
    steps = max(4, min(args.steps, 10))
    for bv, bb in ((20000, 500), (4096, 4096)):
        leg["knobs_%d_%d" % (bv, bb)], _ = knob_leg(ctx, args, W, bv, bb, args.k, steps, 2, live=args.live_traffic_hbm and live_traffic_wanted(ctx, args), fresh=True)
        try:
            leg["knobs_%d_%d" % (bv, bb)]["dram_side"] = dram_side_figures(ctx, args, W, bv, bb, args.k, steps, leg["knobs_%d_%d" % (bv, bb)])
        except Exception as e:
            leg["knobs_%d_%d" % (bv, bb)]["dram_side"] = {"error": repr(e)[:300]}
        kl_ = leg["knobs_%d_%d" % (bv, bb)]
        kl_.pop("_Rfresh", None)
        ds_ = kl_.get("dram_side", {})
        if kl_["roofline"].get("kernel") == "pqt_k_sr_adc" and "distinct_rows_per_batch" in ds_:
            # independent cross-check of roofline.deduplicated (device counters of the pass): distinct rows by torch.unique over the whole candidate
            # lists of the standard batch (pqt_query_candidates), candidates from the same lists
            once_ = ds_["distinct_rows_per_batch"] * (4 * W["w"]["LP"] + 4) + ds_["candidates_per_batch"] * 4
            ms_ = max(kl_["roofline"]["avg_launch_ms"], 1e-9)
            kl_["roofline"]["crosscheck_by_candidate_lists"] = {"distinct_rows": ds_["distinct_rows_per_batch"], "candidates": ds_["candidates_per_batch"], "bytes_per_launch": once_,
                                                               "frac": once_ / ms_ / 1e6 / HBM_PEAK_GBS,
                                                               "note": "the standard batch; the leg's own figures are those of fresh batches (deduplicated.*: the last timed batch)"}
    W["idx"].close()
    del W
    torch.cuda.empty_cache()
    leg["leg_seconds"] = time.time() - t0
    return leg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=None, choices=list(WORKLOADS),
                    help="default: sift1m on 1 GPU (BASELINE configs[1]); synth100m range-sharded on every N >= 2 (configs[2], the scaling sweep's one workload), "
                         "with synth1b (configs[3]) as the headline beside it on 8")
    ap.add_argument("--bv", type=int, default=20000, help="boundVectors (reference default: query(20000, 500, ...))")
    ap.add_argument("--bb", type=int, default=500, help="boundBins")
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-gt", action="store_true", help="skip the brute-force ground truth (recall fields become null): for rocprofv3 --pmc passes, "
                    "where torch's reduce kernels crash the profiler on this image")
    ap.add_argument("--extras", action="store_true", help="also run the side legs (knob set (4096,4096), exact re-rank of the top-k, "
                    "opt-in ADC modes); off by default so that a profile of the default command contains only the headline path's launches")
    ap.add_argument("--no-live-traffic", action="store_true", help="N = 1: do not collect roofline.traffic live (two child runs under rocprofv3 --pmc, ~1 min); "
                                                                   "the committed profile's figure is used instead")
    ap.add_argument("--live-traffic-hbm", action="store_true", help="also collect the traffic of the hbm_roofline_leg's launches live (four child runs that build the 100 M index: "
                                                                    "minutes; scripts/r04_profile_all.sh does); by default that leg carries the committed profile's figures")
    ap.add_argument("--no-hbm-leg", action="store_true", help="N = 1: skip config.hbm_roofline_leg (the 100 M-vector configuration beside the headline)")
    ap.add_argument("--hbm-workload", default="synth100m", choices=list(WORKLOADS), help="workload of the hbm_roofline_leg (tests use a small one)")
    ap.add_argument("--shard-db", action="store_true", help="(default for --gpus N > 1) range-shard the database")
    ap.add_argument("--replicas", action="store_true", help="multi-GPU: replicate the index and shard the queries instead (weak scaling, no collective)")
    ap.add_argument("--exchange", default="alltoall", choices=["alltoall", "allgather"],
                    help="range-sharded run: per-shard top-k exchanged by query slice (all-to-all, merged slices all-gathered) or by one all-gather of the whole lists")
    ap.add_argument("--traversal", default=None, choices=["sharded", "replicated"],
                    help="range-sharded run: traversal sharded by queries with one all-gather of the per-query bin lists (default) or replicated on every rank")
    ap.add_argument("--pipeline", type=int, default=2, choices=[1, 2, 3],
                    help="2 (default) = two whole batches in flight -- on ONE GPU as well: consecutive steps alternate between the index and a view of it on two streams, the line carries the one-batch-at-a-time figures as config.one_batch_at_a_time; range-sharded run: 2 = two whole batches in flight (consecutive steps alternate between two streams / handles: the collectives of one "
                         "batch pass under the kernels of the other), 1 = one batch at a time, 3 = every step split into two half batches in flight; the line "
                         "carries the one-batch-at-a-time step time as config.pipeline_ab (or the two-batches figure when run with --pipeline 1)")
    ap.add_argument("--no-ref1", action="store_true", help="range-sharded run: skip the single-GPU timing of the same database on rank 0")
    ap.add_argument("--no-scaling-leg", action="store_true", help="--gpus 8 default (synth1b): skip the synth100m strong-scaling leg of the sweep")
    ap.add_argument("--option", action="append", default=[], help="name=value passed to pqt_index_set_option (e.g. adc_bias=1)")
    ap.add_argument("--timing-period", type=int, default=4,
                    help="every N-th call of the timed region carries the per-kernel start/stop HIP events the roofline figures come from "
                         "(1 = every call: the events cost ~10 us per call, see DESIGN.md section 6)")
    ap.add_argument("--dataset-dir", default=None, help="real data: directory with sift_base / sift_learn / sift_query (.fvecs or .bvecs) and sift_groundtruth.ivecs")
    ap.add_argument("--dataset-train", type=int, default=20000, help="--dataset-dir: vectors of the learn set used by createTree (the reference trains on 20000)")
    ap.add_argument("--iso-noise", type=float, default=GEN["iso_noise"])
    ap.add_argument("--lat-noise", type=float, default=GEN["lat_noise"])
    ap.add_argument("--centers", type=int, default=GEN["n_centers"])
    ap.add_argument("--center-scale", type=float, default=GEN["center_scale"])
    ap.add_argument("--query-mode", default="fresh", choices=["fresh", "perturbed"])
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)  # does not return
    ctx = init_ctx(args)
    world, rank, dev, dist = ctx.world, ctx.rank, ctx.dev, ctx.dist
    if args.traversal is None:
        args.traversal = os.environ.get("PQT_BENCH_TRAVERSAL", "sharded")
    GEN.update(iso_noise=args.iso_noise, lat_noise=args.lat_noise, n_centers=args.centers, center_scale=args.center_scale)
    if args.dataset_dir:
        out = run_dataset_dir(ctx, args)
        if rank == 0:
            print(json.dumps(out))
        return
    mode = "single" if (world == 1 and not ctx.force_shard) else ("replica" if args.replicas else "shard_db")
    # the one workload of the strong-scaling sweep, and the headline of an 8-GPU run (PQT_BENCH_SWEEP_WL / PQT_BENCH_HEAD_WL: the test-suite
    # walks the 8-GPU code path with two ranks and small stand-ins)
    SWEEP_WL = os.environ.get("PQT_BENCH_SWEEP_WL", "synth100m")
    head_wl = os.environ.get("PQT_BENCH_HEAD_WL") or ("synth1b" if world >= 8 else SWEEP_WL)
    wl_name = args.workload or ("sift1m" if mode != "shard_db" else head_wl)
    # what the chosen workload costs to BUILD on this run's ranks, before anything is built -- a rank encodes ~13 M vectors/s
    # (7.5 s per 100 M, profiles/r04_bench_default.json build_s) and synthesises, sorts and brute-forces its ground truth at about the same rate
    # again; a default that would not fit the budget falls back to the sweep's workload and says so
    planned_build_s = WORKLOADS[wl_name]["n_base"] / max(1, world if mode == "shard_db" else 1) / 13e6 * 2.0 + 20.0
    build_budget_s = float(os.environ.get("PQT_BENCH_BUILD_BUDGET_S", "900"))
    head_fallback = None
    log("[bench] workload %s on %d rank(s): planned build time %.0f s per rank (budget %.0f s)" % (wl_name, world, planned_build_s, build_budget_s))
    if not args.workload and mode == "shard_db" and wl_name != SWEEP_WL and planned_build_s > build_budget_s:
        head_fallback = "the default headline workload %s would take ~%.0f s per rank to build (budget %.0f s, PQT_BENCH_BUILD_BUDGET_S): the sweep's workload %s is the headline instead" % (
            wl_name, planned_build_s, build_budget_s, SWEEP_WL)
        log("[bench] " + head_fallback)
        wl_name = SWEEP_WL
    W = build_workload(ctx, args, wl_name, mode)
    w, n, qn, idx, meta, queries, k = W["w"], W["n"], W["qn"], W["idx"], W["meta"], W["queries"], args.k
    R = time_path(ctx, args, W, args.bv, args.bb, k, args.steps, args.warmup, args.timing_period, single_pipeline=mode != "shard_db")
    out_idx, out_dist, out_cnt, gt, stream = R["out_idx"], R["out_dist"], R["out_cnt"], W["gt"], ctx.stream
    st = R["st"]
    if os.environ.get("PQT_PRINT_STATS"):  # per-stage tie counters and totals of the last batch (development aid)
        log("[stats]", {k_: int(v_) for k_, v_ in st.items() if isinstance(v_, (int, np.integer))})
    if os.environ.get("PQT_TSTAMP"):
        import ctypes
        ts = np.zeros((qn, 24), np.uint64)
        L = ctx.pkg.lib()
        L.pqt_debug_tstamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
        if L.pqt_debug_tstamps(idx.h, ts.ctypes.data, qn) == 0:
            d = np.diff(ts[:, :9].astype(np.int64), axis=1)
            log("[tstamp] phase cycles median:", np.median(d, axis=0).astype(int).tolist(), " total median", int(np.median(ts[:, 8].astype(np.int64) - ts[:, 0].astype(np.int64))),
                " p90", int(np.percentile(ts[:, 8].astype(np.int64) - ts[:, 0].astype(np.int64), 90)))
            if os.path.isdir("gpurun_out"): np.save("gpurun_out/tstamps.npy", ts)
            r = ts[:, 9:14].astype(np.int64)
            tot = r[:, 4] & 0xffffffff  # [13] = clocks of the query | wall-clock start << 32 (scripts/r02_tstamp_wg.py reads the rest)
            log("[tstamp] rerank_select per query (shader clocks): total median %d p90 %d max %d | rows wait %d  adc+filter %d  flush %d (medians)"
                % (np.median(tot), np.percentile(tot, 90), tot.max(), np.median(r[:, 1]), np.median(r[:, 2]), np.median(r[:, 3])))
            log("[tstamp] sum of per-query clocks / 3072 wave slots: %d" % (tot.sum() // 3072))
            x = ts[:, 16:20].astype(np.int64)
            log("[tstamp] rerank_select detail (medians): set-up %d  band re-evaluation %d  result write-out %d  | candidates %d" % tuple(np.median(x, axis=0).astype(int).tolist()))
    if os.environ.get("PQT_DBG_SWEEP"):
        # debug: stage times with parts of the kernels switched off (results wrong), same index, no rebuild
        for v in os.environ["PQT_DBG_SWEEP"].split(","):
            idx.set_option("debug_bits", int(v))
            for _ in range(6):
                R["step"]()
            barrier(ctx)
            h_ = idx.stage_ms_history(5).mean(0).tolist()
            log("[dbg-sweep] bits %s: traverse %.4f  rerank_select %.4f ms" % (v, h_[1], h_[3]))
        idx.set_option("debug_bits", 0)
        R["step"]()
        barrier(ctx)

    out = make_line(ctx, args, W, R)
    out["config"]["planned_build_s_per_rank"] = planned_build_s
    if head_fallback:
        out["config"]["headline_workload_fallback"] = head_fallback

    # ---- one device, two batches in flight: the same steps one batch at a time on one stream (each kernel alone on the device: the
    # per-kernel durations and roofline fractions without the overlap), beside the line's own figures
    if mode != "shard_db" and R["pipeline"] == 2 and not os.environ.get("PQT_BENCH_NO_PIPELINE"):  # (that variable: headline launches only, for kernel statistics)
        try:
            R1 = time_path(ctx, args, W, args.bv, args.bb, k, args.steps, min(args.warmup, 3), args.timing_period, pipeline=1)
            roof1, _ = roofline_block(ctx, args, W, R1, live=False)
            # both figures at the top level, each labelled -- `value` = the K timed steps with two whole batches in flight,
            # value_one_batch_at_a_time = the same K steps issued one after the other on one stream (the reference's form: one batch at a time)
            out["value_one_batch_at_a_time"] = R1["qps"]
            out["ms_per_step_one_batch_at_a_time"] = R1["ms_per_step"]
            out["value_is"] = "two whole 10 k-query batches in flight on one device (every timed step one complete batch, all complete at the closing barrier); value_one_batch_at_a_time = the same steps one batch at a time"
            out["config"]["view_slot_results_identical"] = R.get("view_slot_identical")
            out["config"]["one_batch_at_a_time"] = {
                "queries_per_sec": R1["qps"], "ms_per_step": R1["ms_per_step"], "stage_ms": R1["stage"],
                "results_identical": bool(torch.equal(R1["out_idx"], out_idx) and torch.equal(R1["out_dist"], out_dist) and torch.equal(R1["out_cnt"], out_cnt)),
                "what": "the %d timed steps issued one batch at a time on one stream (the form of `value` up to round 3 and of --pipeline 1)" % R1["steps"]}
            # `roofline` prices the dominant KERNEL: its launch alone on the device (this leg's timed region: the same K steps, same barriers, HIP events on the
            # launch stream).  Under the two batches in flight of `value` a launch shares the device with the other batch's launch and its own duration
            # stretches while the device does more per unit of time: those figures stay in the line as roofline.in_flight, and the whole step's rate is
            # config.path_GBps / path_frac_of_hbm_peak (algorithmic bytes of both launches over ms_per_step of `value`).
            rf = out["roofline"]
            # the same kernel's figures inside the timed region of `value` (it may be the other one of the two there)
            flt = {rf["kernel"]: {"avg_launch_ms": rf["avg_launch_ms"], "achieved": rf["achieved"], "frac": rf["frac"]}}
            for n_, v_ in rf["other_kernels"].items():
                flt[n_] = {"avg_launch_ms": v_["avg_launch_ms"], "achieved": v_["GBps"], "frac": v_["frac"]}
            mine = flt.get(roof1["kernel"], {})
            in_flight = dict(kernel=roof1["kernel"], **mine, all_kernels=flt, timing=rf["timing"],
                             what="the same launches inside the timed region of `value` (two whole batches in flight: every launch shares the device with the other batch's)")
            if roof1["kernel"] != rf["kernel"]:  # (the live counters were collected for the other kernel: the committed profile's figure of this one, if any)
                for key_ in ("traffic", "traffic_source", "traffic_committed_profile"):
                    rf[key_] = roof1[key_]
            for key_ in ("kernel", "avg_launch_ms", "achieved", "frac", "other_kernels", "algorithmic_bytes_per_launch", "intermediate_bytes"):
                rf[key_] = roof1[key_]
            iss = rf.get("issue")
            if iss and "instructions_per_launch" in iss and iss.get("kernel") == rf["kernel"]:
                # the counters come from one-batch-at-a-time child runs: priced against the kernel's duration ALONE on the device
                cyc = rf["avg_launch_ms"] * 1e-3 * iss["engine_clock_GHz"] * 1e9
                iss["launch_ms"] = rf["avg_launch_ms"]
                iss["clocks_per_instruction_per_simd"] = cyc / max(iss["instructions_per_simd"], 1.0)
                iss["frac_of_issue_ceiling"] = iss["instructions_per_simd"] * 4.0 / max(cyc, 1.0)
            rf["traffic_ratio"] = None
            if rf.get("traffic") and rf.get("algorithmic_bytes_per_launch"):
                rf["traffic_ratio"] = rf["traffic"] / rf["algorithmic_bytes_per_launch"]
            rf["timing"] = ("the kernel ALONE on the device: mean of its own duration over the %d of the %d steps of config.one_batch_at_a_time that carry events (the timed steps "
                            "issued one batch at a time, same barriers; start/stop HIP events attached to the dispatch on the launch stream); "
                            "roofline.in_flight = the same launches inside the timed region of `value`, where they overlap the other batch's" % (R1["n_timed"], R1["steps"]))
            rf["in_flight"] = in_flight
        except Exception as e:
            out["config"]["one_batch_at_a_time"] = {"error": repr(e)[:300]}

    # optional "next" row 8f-4 (not part of the timed path): exact re-rank of the k results against the raw uint8 vectors
    exact = None
    raw_u8 = W["raw_u8"]
    if args.extras and raw_u8 is not None and k <= 512 and mode != "shard_db":
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
    out["config"]["exact_rerank_of_topk"] = exact

    # ---- measured read bandwidth of this device (read-only streaming kernel over 4 GiB, the access shape of the group-major
    # rerank), reported beside the nominal peak the fractions above are priced with
    if mode == "single":
        try:
            gbs_ = ctx.pkg.stream_read_GBps(4 << 30, 5, dev.index or 0)
            out["roofline"]["measured_stream_GBps"] = gbs_
            out["roofline"]["measured_stream_what"] = "read-only kernel, 16-byte loads, every byte of a 4 GiB buffer once; best of 15 launch shapes (4/8/16 loads in flight, grid-stride or contiguous share per workgroup, 8/16/32 workgroups per CU: pqt_debug_stream_read)"
            out["roofline"]["frac_of_measured_stream"] = out["roofline"]["achieved"] / gbs_
        except Exception as e:
            out["roofline"]["measured_stream_GBps"] = None

    def side_leg(bv_, bb_, k_, reps=5):
        oi_ = torch.empty((qn, k_), dtype=torch.int32, device=dev)
        od_ = torch.empty((qn, k_), dtype=torch.float32, device=dev)
        for _ in range(2):
            idx.query_dev(queries, bv_, bb_, k_, oi_, od_, out_cnt, stream=stream)
        torch.cuda.synchronize(dev)
        t2 = time.perf_counter()
        for _ in range(reps):
            idx.query_dev(queries, bv_, bb_, k_, oi_, od_, out_cnt, stream=stream)
        torch.cuda.synchronize(dev)
        t2 = (time.perf_counter() - t2) / reps
        h_ = idx.stage_ms_history(reps).mean(0).tolist()
        st_ = idx.stats()
        i2 = oi_.to(torch.int64) & 0xffffffff
        kb_ = kernel_bytes(w, qn, k_, st_["bins_visited"] / max(1, st_["queries"]), st_["candidates"], k_ <= 128)
        return {"queries_per_sec": qn / t2, "ms_per_step": t2 * 1e3, "k": k_, "recall@1": recall_at(i2, gt, 1), "recall@100": recall_at(i2, gt, 100),
                "mean_candidates": float(out_cnt.float().mean()), "kernel_path": idx.last_path(),
                "stage_ms": dict(zip(STAGES, h_)),
                "rerank_select_GBps": kb_["rerank_select"][0] / max(h_[3], 1e-9) / 1e6,
                "rerank_select_frac": kb_["rerank_select"][0] / max(h_[3], 1e-9) / 1e6 / HBM_PEAK_GBS}, oi_, od_

    # ---- second knob set of BASELINE.md (the CUDA library's defaults k1/maxBins: boundVectors = boundBins = 4096), short leg,
    # reported beside the headline (never as `value`)
    if args.extras and mode == "single" and (args.bv, args.bb) == (20000, 500):
        try:
            idx.build_heuristic(4096)
            leg, _, _ = side_leg(4096, 4096, k)
            leg["launch_structure"] = "fused traversal in wide mode (boundBins > 512: rows in blocks of 512, populated rows listed) + fused rerank/select"
            out["config"]["knobs_4096_4096"] = leg
            leg, _, _ = side_leg(4096, 4096, 4096, reps=3)  # the reference front-end's own call: queryKNN(..., 4096) (tool_query.cpp:155)
            leg["launch_structure"] = "fused traversal (wide mode) + fused rerank/select for 128 < k <= 4096 (distances stay on chip)"
            out["config"]["knobs_4096_4096_k4096"] = leg
            idx.set_option("fused", 0)
            leg, _, _ = side_leg(4096, 4096, 4096, reps=3)
            idx.set_option("fused", 1)
            leg["launch_structure"] = "staged kernels (tables, bins, rerank -> candDist in HBM, select)"
            out["config"]["knobs_4096_4096_k4096_staged"] = leg
            if w["P"] == 4:
                # optional mode: the CUDA 1B path's 2-D anisotropic sequences choose the enumerated rows (per-query row tables:
                # the traversal runs as staged kernels, the rerank is the headline's), at the headline knobs
                idx.build_heuristic_2d(512)
                leg, _, _ = side_leg(args.bv, args.bb, k)
                leg["launch_structure"] = "staged traversal: tables, per-query rows (pqt_k_rows_2d), bins; fused rerank/select"
                out["config"]["heuristic_2d_512"] = leg
                idx.build_heuristic(4096)
            idx.query_dev(queries, args.bv, args.bb, k, out_idx, out_dist, out_cnt, stream=stream)  # restore the headline outputs
            torch.cuda.synchronize(dev)
        except Exception as e:
            out["config"]["knobs_4096_4096"] = {"error": repr(e)[:200]}

    # ---- the kept C++ front-end: pqt::PerturbationProTree::queryKNN as the reference's tool_query calls it
    # (tool_query.cpp:153-161: batches of <= 4096 queries, device query pointer, two std::vectors resized + filled), wall clock per
    # call split into kernels / D2H / host; "legacy_copy" = round 3's hand-over (two synchronous pageable copies of the padded arrays)
    if args.extras and mode == "single" and not W["chunked"]:
        try:
            fe_mod = importlib.import_module("product-quantization-tree_amd.frontend")
            codes_h = idx._keep[0].cpu().numpy().view(np.uint32)
            fe = fe_mod.FrontEnd(w["D"], w["P"], w["C1"], w["C2"], w["W"], w["LP"], meta["cb1"], meta["cb2"], meta["bin_ids"], meta["sizes"], meta["members"], codes_h,
                                 devices=(dev.index or 0,))
            del codes_h
            leg = {"what": "wall clock of PerturbationProTree::queryKNN per call (mean of 5 after 1 warm-up), this workload's index handed over as host arrays; "
                           "compact = the library's own choice: a large result with at least half of it padding is packed on the device (pqt_compact_results), only the "
                           "filled prefixes cross PCIe into pinned staging while 8 host threads write the padding with streaming stores and then scatter the rows "
                           "(packed = 1); small or dense results are copied whole (packed = 0); legacy_copy = always the whole padded [QN][nVec] arrays into the "
                           "caller's pageable vectors (round 3)"}
            for name_, qn_, nvec_, bv_, bb_ in (("qn4096_nvec4096_knobs_4096_4096", min(4096, qn), 4096, 4096, 4096), ("qn4096_nvec100", min(4096, qn), 100, args.bv, args.bb),
                                                ("qn%d_nvec100" % qn, qn, 100, args.bv, args.bb)):
                e_ = {}
                for variant in ("compact", "legacy_copy"):
                    fe.set_legacy_copy(variant == "legacy_copy")
                    fe.queryKNN(queries.data_ptr(), qn_, nvec_, bv_, bb_, reps=2, want_results=False)
                    tm_, _, _ = fe.queryKNN(queries.data_ptr(), qn_, nvec_, bv_, bb_, reps=5, want_results=False)
                    tm_["queries_per_sec"] = qn_ / (tm_["total_ms"] * 1e-3)
                    tm_["copy_share_of_call"] = tm_["d2h_ms"] / max(tm_["total_ms"], 1e-9)
                    e_[variant] = tm_
                fe.set_legacy_copy(False)
                # the padding memory switched off (every call writes the whole padding, round 4's form), and two batches in flight
                # (queryKNNAsync / queryKNNCollect: the loop of host/tool_query.cpp), wall clock per batch over 8 batches
                fe.set_keep_padding(False)
                fe.queryKNN(queries.data_ptr(), qn_, nvec_, bv_, bb_, reps=1, want_results=False)
                tm_, _, _ = fe.queryKNN(queries.data_ptr(), qn_, nvec_, bv_, bb_, reps=5, want_results=False)
                e_["compact_whole_padding_every_call"] = tm_
                fe.set_keep_padding(True)
                fe.queryKNN_inflight(queries.data_ptr(), queries.data_ptr(), qn_, nvec_, bv_, bb_, reps=3, keep_padding=True, want_results=False)
                ms_, _, _ = fe.queryKNN_inflight(queries.data_ptr(), queries.data_ptr(), qn_, nvec_, bv_, bb_, reps=8, keep_padding=True, want_results=False)
                e_["two_batches_in_flight"] = {"ms_per_batch": ms_, "queries_per_sec": qn_ / (ms_ * 1e-3)}
                leg[name_] = e_
            fe.close()
            out["config"]["frontend_queryKNN"] = leg
        except Exception as e:
            out["config"]["frontend_queryKNN"] = {"error": repr(e)[:300]}

    # ---- the headline batch without stage events on any call (the events cost ~0.01 ms per call), and as two pieces on two streams
    # ("overlap" = 1: opt-in since the rerank's statistics atomics stopped serialising the launch, DESIGN.md section 4 "Round 3")
    if mode == "single" and not W["chunked"] and not os.environ.get("PQT_BENCH_NO_PIPELINE") and not args.option:
        try:
            idx.set_option("stage_timing", 0)
            oi1, od1, oc1 = torch.empty_like(out_idx), torch.empty_like(out_dist), torch.empty_like(out_cnt)
            leg = {}
            for name, ov in (("one_piece", 0), ("two_pieces", 1)):
                idx.set_option("overlap", ov)
                for _ in range(3):
                    idx.query_dev(queries, args.bv, args.bb, k, oi1, od1, oc1, stream=stream)
                torch.cuda.synchronize(dev)
                t3 = time.perf_counter()
                for _ in range(args.steps):
                    idx.query_dev(queries, args.bv, args.bb, k, oi1, od1, oc1, stream=stream)
                torch.cuda.synchronize(dev)
                t3 = (time.perf_counter() - t3) / args.steps
                leg[name] = {"queries_per_sec": qn / t3, "ms_per_step": t3 * 1e3, "kernel_path": idx.last_path(),
                             "results_identical": bool(torch.equal(oi1, out_idx) and torch.equal(od1, out_dist) and torch.equal(oc1, out_cnt))}
            leg["what"] = ("the headline batch with no stage events on any call: in one piece (the default) and as two pieces on two streams "
                           "(option overlap = 1); `value` above carries events on every timing-period-th step")
            out["config"]["no_stage_events"] = leg
            idx.set_option("overlap", -1)
            idx.set_option("stage_timing", 1)
            idx.query_dev(queries, args.bv, args.bb, k, out_idx, out_dist, out_cnt, stream=stream)
            torch.cuda.synchronize(dev)
        except Exception as e:
            out["config"]["no_stage_events"] = {"error": repr(e)[:200]}

    # ---- the headline step with the host-to-device copy of the query batch inside the timed region (pinned host buffer,
    # same stream); `value` above starts with the queries resident in HBM
    if mode == "single" and not os.environ.get("PQT_BENCH_NO_PIPELINE"):
        try:
            qh = torch.empty(queries.shape, dtype=queries.dtype, pin_memory=True)
            qh.copy_(queries)
            qd = torch.empty_like(queries)
            oi1, od1, oc1 = torch.empty_like(out_idx), torch.empty_like(out_dist), torch.empty_like(out_cnt)
            idx.set_option("stage_timing", 0)

            def step_h2d():
                qd.copy_(qh, non_blocking=True)
                idx.query_dev(qd, args.bv, args.bb, k, oi1, od1, oc1, stream=stream)
            t_h = time_steps(step_h2d, lambda: torch.cuda.synchronize(dev), 3, args.steps) / args.steps
            idx.set_option("stage_timing", 1)
            out["config"]["h2d_included"] = {"queries_per_sec": qn / t_h, "ms_per_step": t_h * 1e3, "h2d_bytes_per_step": int(qh.numel() * 4),
                                             "results_identical": bool(torch.equal(oi1, out_idx) and torch.equal(od1, out_dist)),
                                             "what": "every step first copies its %d x %d f32 queries from pinned host memory on the launch stream (no stage events); results stay in HBM" % (qn, w["D"])}
            # two batches in flight like `value` -- consecutive steps alternate between the index and its view, each on its
            # own stream with its own device query buffer: the copy of one batch crosses PCIe under the other batch's kernels
            if "view" in W and "slot_stream" in W:
                view2, s2 = W["view"], W["slot_stream"]
                qd2 = torch.empty_like(queries)
                oi2, od2, oc2 = torch.empty_like(out_idx), torch.empty_like(out_dist), torch.empty_like(out_cnt)
                view2.set_option("stage_timing", 0)
                idx.set_option("stage_timing", 0)
                calls2 = [0]

                def step_h2d2():
                    if calls2[0] & 1:
                        with torch.cuda.stream(s2):
                            qd2.copy_(qh, non_blocking=True)
                        view2.query_dev(qd2, args.bv, args.bb, k, oi2, od2, oc2, stream=s2.cuda_stream)
                    else:
                        qd.copy_(qh, non_blocking=True)
                        idx.query_dev(qd, args.bv, args.bb, k, oi1, od1, oc1, stream=stream)
                    calls2[0] += 1
                t_h2 = time_steps(step_h2d2, lambda: torch.cuda.synchronize(dev), 4, args.steps) / args.steps
                idx.set_option("stage_timing", 1)
                out["config"]["h2d_included"]["two_batches_in_flight"] = {
                    "queries_per_sec": qn / t_h2, "ms_per_step": t_h2 * 1e3,
                    "results_identical": bool(torch.equal(oi1, out_idx) and torch.equal(od1, out_dist) and torch.equal(oi2, out_idx) and torch.equal(od2, out_dist)),
                    "what": "the same steps alternating between the index and its view on two streams, each copying its own batch on its own stream"}
                # zero copy: the batch stays in the pinned host buffer (device-visible host memory) and the traversal kernel reads its 512-byte query
                # vectors over PCIe itself -- unlike the copy engine's transfers (which do not overlap the other slot's kernels on this stack:
                # scripts/r05_h2d_overlap.py, copies only 0.09 + kernels 0.12 = the 0.27 ms of the variant above) these reads are ordinary loads
                # of a kernel and pass under the other batch's rerank
                calls3 = [0]

                def step_zero():
                    if calls3[0] & 1:
                        view2.query_dev(qh, args.bv, args.bb, k, oi2, od2, oc2, stream=s2.cuda_stream)
                    else:
                        idx.query_dev(qh, args.bv, args.bb, k, oi1, od1, oc1, stream=stream)
                    calls3[0] += 1
                idx.set_option("stage_timing", 0)
                t_z = time_steps(step_zero, lambda: torch.cuda.synchronize(dev), 4, args.steps) / args.steps
                idx.set_option("stage_timing", 1)
                out["config"]["h2d_included"]["zero_copy_two_batches_in_flight"] = {
                    "queries_per_sec": qn / t_z, "ms_per_step": t_z * 1e3,
                    "results_identical": bool(torch.equal(oi1, out_idx) and torch.equal(od1, out_dist) and torch.equal(oi2, out_idx) and torch.equal(od2, out_dist)),
                    "what": "no copy at all: every step's query pointer is the pinned HOST buffer, the traversal reads it over PCIe; steps alternate between the index and its view"}
                del qd2, oi2, od2, oc2
            del qh, qd, oi1, od1, oc1
        except Exception as e:
            out["config"]["h2d_included"] = {"error": repr(e)[:200]}

    # ---- CPU baseline (rank 0, N=1 only): the oracle restatement of cpu_version's query(), bounded sample ----------
    if mode == "single" and not args.no_cpu and not W["chunked"]:
        from oracle import Oracle
        # the checker builds its OWN heuristic table (prepareHeuristic restatement) -- it is compared with the library's below
        o = Oracle(w["D"], w["P"], w["C1"], w["C2"], w["W"], w["LP"], heur_keep=max(args.bb, 1))
        heur_same = bool(np.array_equal(o.heuristic(max(args.bb, 1)), idx.heuristic(max(args.bb, 1))))
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
                               "result_lists_identical_frac": same, "heuristic_table_identical": heur_same}
        del o, codes_host
    elif mode == "single" and not args.no_cpu and W["chunked"] and n <= 20_000_000:
        out["cpu_baseline"] = cpu_baseline_chunked(ctx.pkg, idx, w, meta, queries, args, out_idx, out_dist, k)

    if mode == "shard_db":
        out["config"]["same_workload_1gpu"] = ref1 = same_workload_1gpu(ctx, args, W, R)
        out["config"]["ranks_agree"] = ranks_agree(ctx, R)  # every rank must hold the same merged result
        try:  # the other schedule on the same index and batch, a few steps: what the half batches in flight buy (or cost)
            other = 1 if R["pipeline"] >= 2 else 2
            Ro = time_path(ctx, args, W, args.bv, args.bb, k, max(3, min(args.steps, 10)), 2, args.timing_period, pipeline=other)
            out["config"]["pipeline_ab"] = {"this_line": "pipeline=%d" % R["pipeline"], "other": "pipeline=%d" % other, "other_ms_per_step": Ro["ms_per_step"],
                                            "other_queries_per_sec": Ro["qps"], "other_exchange_ms": Ro["exchange_ms"],
                                            "results_identical": bool(torch.equal(Ro["out_idx"], R["out_idx"]) and torch.equal(Ro["out_dist"], R["out_dist"]) and torch.equal(Ro["out_cnt"], R["out_cnt"]))}
            del Ro
        except Exception as e:
            out["config"]["pipeline_ab"] = {"error": repr(e)[:300]}
        if ref1 and "speedup_of_this_run" in ref1:
            out["scaling_vs_1gpu"] = ref1["speedup_of_this_run"]
            out["scaling_vs_1gpu_what"] = "this line's queries/sec / the same database and batch on ONE GPU (config.same_workload_1gpu, timed on rank 0 in this run)"
        if W["name"] != SWEEP_WL and not args.workload and not args.no_scaling_leg:
            # --gpus 8: `value` above is configs[3]'s size; the sweep's own workload (100 M vectors, configs[2]) rides beside it so that
            # N = 2, 4, 8 all carry the ratio against ONE GPU on ONE workload
            idx.close()
            del W, R, idx, queries, out_idx, out_dist, out_cnt
            torch.cuda.empty_cache()
            W2 = build_workload(ctx, args, SWEEP_WL, mode, codebooks=(meta["cb1"], meta["cb2"]))
            R2 = time_path(ctx, args, W2, args.bv, args.bb, k, args.steps, args.warmup, args.timing_period)
            l2 = make_line(ctx, args, W2, R2)
            ref2 = same_workload_1gpu(ctx, args, W2, R2)
            out["config"]["strong_scaling_leg"] = {"workload": l2["config"]["workload"], "queries_per_sec": l2["value"], "ms_per_step": l2["ms_per_step"],
                                                   "stage_ms": l2["config"]["stage_ms"], "recall@1": l2["config"]["recall@1"], "mean_candidates_this_rank": l2["config"]["mean_candidates_this_rank"],
                                                   "roofline": {k_: l2["roofline"][k_] for k_ in ("kernel", "achieved", "frac", "avg_launch_ms")},
                                                   "same_workload_1gpu": ref2, "ranks_agree": ranks_agree(ctx, R2)}
            if ref2 and "speedup_of_this_run" in ref2:
                out["scaling_vs_1gpu"] = ref2["speedup_of_this_run"]
                out["scaling_vs_1gpu_what"] = ("strong scaling of the sweep's workload (config.strong_scaling_leg: 100 M vectors range-sharded over this run's GPUs / the same database "
                                               "on ONE GPU); `value` is the 1 B-vector configuration, which does not fit one GPU")
            W2["idx"].close()
    elif mode == "single":
        out["scaling_note"] = ("N = 1 line = BASELINE configs[1]; the strong-scaling sweep (N >= 2) runs ONE workload, the 100 M-vector configs[2] database, and every N >= 2 "
                               "line carries its own one-GPU denominator (config.same_workload_1gpu) and the ratio (scaling_vs_1gpu); this line's config.hbm_roofline_leg "
                               "is that workload on this GPU")
        if not args.no_hbm_leg and not args.workload:
            try:
                idx.close()
                del W, R, idx, queries, out_idx, out_dist, out_cnt, raw_u8
                torch.cuda.empty_cache()
                leg_ = hbm_roofline_leg(ctx, args)
                gbs_ = out["roofline"].get("measured_stream_GBps")
                if gbs_:
                    for kk_ in ("knobs_20000_500", "knobs_4096_4096"):
                        r_ = (leg_.get(kk_) or {}).get("roofline")
                        if r_:
                            r_["measured_stream_GBps"] = gbs_
                            r_["frac_of_measured_stream"] = r_["achieved"] / gbs_
                out["config"]["hbm_roofline_leg"] = leg_
            except Exception as e:
                out["config"]["hbm_roofline_leg"] = {"error": repr(e)[:300]}
    if rank == 0:
        print(json.dumps(out))
    if ctx.collectives:
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------------
# real data (BASELINE.md 2 "Real SIFT1M/SIFT1B"): the reference's own pipeline -- convert -> tool_createdb -> query -> recall
# (scripts/prepare_data.sh:3, tool_createdb.cpp:73-114, cpu_version/tools/query.cpp:29-82) -- on the files of --dataset-dir
# ------------------------------------------------------------------------------------------------------
def read_vecs(path, limit=None):
    """.fvecs / .bvecs / .ivecs (TEXMEX layout: per vector an int32 dimension followed by dim values)."""
    ext = os.path.splitext(path)[1]
    item = {".fvecs": np.float32, ".ivecs": np.int32, ".bvecs": np.uint8}[ext]
    raw = np.memmap(path, dtype=np.uint8, mode="r")
    dim = int(np.frombuffer(raw[:4].tobytes(), np.int32)[0])
    rec = 4 + dim * np.dtype(item).itemsize
    nvec = raw.shape[0] // rec
    if limit is not None:
        nvec = min(nvec, limit)
    body = np.ascontiguousarray(raw[:nvec * rec].reshape(nvec, rec)[:, 4:])
    return body.view(item).reshape(nvec, dim)


def find_vecs(d, stem):
    for ext in (".fvecs", ".bvecs", ".ivecs"):
        for pre in ("sift_", "bigann_", ""):
            p = os.path.join(d, pre + stem + ext)
            if os.path.exists(p):
                return p
    return None


def run_dataset_dir(ctx, args):
    import subprocess
    import tempfile
    d = args.dataset_dir
    pb, pl, pq, pg = find_vecs(d, "base"), find_vecs(d, "learn"), find_vecs(d, "query"), find_vecs(d, "groundtruth")
    if not (pb and pq and pg):
        raise SystemExit("--dataset-dir %s: need sift_base.{fvecs,bvecs}, sift_query.{fvecs,bvecs} and sift_groundtruth.ivecs (sift_learn optional)" % d)
    wl = dict(WORKLOADS["sift1m"])  # BASELINE configs[0]/[1] parameters
    D, P, C1, C2, Wc, LP = (wl[k_] for k_ in ("D", "P", "C1", "C2", "W", "LP"))
    base = read_vecs(pb).astype(np.float32)
    queries_h = read_vecs(pq).astype(np.float32)
    gt_h = read_vecs(pg)[:, 0].astype(np.int64)
    learn = read_vecs(pl, args.dataset_train).astype(np.float32) if pl else base[:args.dataset_train]
    assert base.shape[1] == D and queries_h.shape[1] == D, "SIFT dimension expected"
    n, qn = base.shape[0], queries_h.shape[0]
    host = os.path.join(ROOT, "product-quantization-tree_amd", "host")
    tmp = tempfile.mkdtemp(prefix="pqt_real_")
    t0 = time.time()

    def write_umem(path, a):  # the reference's .umem container: ASCII "<num>\n<dim>\n" padded to 20 bytes + uint8 payload
        hdr = ("%d\n%d\n" % a.shape).encode().ljust(20, b"\0")
        with open(path, "wb") as f:
            f.write(hdr)
            f.write(np.ascontiguousarray(a, np.uint8).tobytes())
    # learn vectors first, base after them: tool_createdb trains on the first --train rows of --dataset; a second call with the
    # finished codebook builds the database from the base file
    write_umem(os.path.join(tmp, "learn.umem"), learn)
    write_umem(os.path.join(tmp, "base.umem"), base)
    common = ["--dim", str(D), "--p", str(P), "--c1", str(C1), "--c2", str(C2), "--lineparts", str(LP), "--w", str(Wc), "--basename", os.path.join(tmp, "real"),
              "--device", str(ctx.dev.index or 0), "--hashed", "0"]
    r = subprocess.run([os.path.join(host, "tool_createdb")] + common + ["--dataset", os.path.join(tmp, "learn.umem"), "--train", str(learn.shape[0])], capture_output=True, text=True)
    if r.returncode != 0:
        raise SystemExit("tool_createdb (training) failed: " + r.stderr[-2000:] + r.stdout[-2000:])
    t_train = time.time() - t0
    r = subprocess.run([os.path.join(host, "tool_createdb")] + common + ["--dataset", os.path.join(tmp, "base.umem")], capture_output=True, text=True)
    if r.returncode != 0:
        raise SystemExit("tool_createdb (database) failed: " + r.stderr[-2000:] + r.stdout[-2000:])
    t_db = time.time() - t0 - t_train
    pre = os.path.join(tmp, "real_%d_%d_%d_%d" % (D, P, C1, C2))
    # .ppqt: ASCII header dim p p2 c1 c2 ndbs, then cb1, cb2 (PerturbationProTree.cu:60-220)
    with open(pre + ".ppqt", "rb") as f:
        hdr = [int(f.readline()) for _ in range(6)]
        cb1 = np.frombuffer(f.read(4 * C1 * D), np.float32).reshape(C1, D).copy()
        cb2 = np.frombuffer(f.read(4 * P * C1 * C2 * (D // P)), np.float32).reshape(P, C1, C2, D // P).copy()
    assert hdr[0] == D and hdr[1] == P and hdr[3] == C1 and hdr[4] == C2
    # .bins (treequantizer.hpp:745-774): nbins, then {id, size, members...}, then len, lp, codes
    raw = np.fromfile(pre + ".bins", np.uint32)
    nb, o = int(raw[0]), 1
    ids_, sizes_, mem_ = np.empty(nb, np.uint32), np.empty(nb, np.uint32), []
    for b in range(nb):
        ids_[b], sizes_[b] = raw[o], raw[o + 1]
        mem_.append(raw[o + 2:o + 2 + int(raw[o + 1])])
        o += 2 + int(raw[o + 1])
    members = np.concatenate(mem_) if mem_ else np.empty(0, np.uint32)
    assert int(raw[o]) == n and int(raw[o + 1]) == LP
    codes = raw[o + 2:o + 2 + n * LP].reshape(n, LP)
    idx = ctx.pkg.PqtIndex(D, P, C1, C2, Wc, LP, device=ctx.dev.index or 0)
    idx.set_codebooks(cb1, cb2)
    idx.build_heuristic(max(args.bb, 1))
    idx.set_bins(ids_, sizes_, members)
    idx.set_lines(codes)
    k = args.k
    q = torch.from_numpy(queries_h).to(ctx.dev)
    oi = torch.empty((qn, k), dtype=torch.int32, device=ctx.dev)
    od = torch.empty((qn, k), dtype=torch.float32, device=ctx.dev)
    oc = torch.empty(qn, dtype=torch.int32, device=ctx.dev)
    torch.cuda.synchronize(ctx.dev)
    elapsed = time_steps(lambda: idx.query_dev(q, args.bv, args.bb, k, oi, od, oc, stream=ctx.stream), lambda: torch.cuda.synchronize(ctx.dev), args.warmup, args.steps)
    gt = torch.from_numpy(gt_h).to(ctx.dev)
    it = oi.to(torch.int64) & 0xffffffff
    rec = {"recall@1": recall_at(it, gt, 1), "recall@10": recall_at(it, gt, 10), "recall@100": recall_at(it, gt, 100)}
    # the checker on the same index: recall by the reference's definition (cpu_version/tools/query.cpp:29-82: GT[0] at rank < R of the sorted candidate list)
    from oracle import Oracle
    o_ = Oracle(D, P, C1, C2, Wc, LP, heur_keep=max(args.bb, 1))
    o_.set_codebooks(cb1, cb2)
    o_.import_bins(ids_, sizes_, members)
    o_.import_codes(codes)
    o_.set_sort_mode(1)
    ci, cd, cc = o_.query_batch(queries_h, args.bv, args.bb, k, nthreads=usable_cores(o_.max_threads()))
    ct = torch.from_numpy(ci.astype(np.int64))
    gtc = gt.cpu()
    orec = {"recall@1": recall_at(ct, gtc, 1), "recall@10": recall_at(ct, gtc, 10), "recall@100": recall_at(ct, gtc, 100)}
    gi = oi.cpu().numpy().view(np.uint32)
    same = float(np.mean([np.array_equal(gi[i], ci[i]) for i in range(qn)]))
    idx.close()
    return {"metric": "queries/sec + recall@1/@100, SIFT1M (1 GPU) and SIFT1B (8 GPUs)", "value": qn * args.steps / elapsed, "unit": "queries/sec", "n_gpus": 1,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "real: %s (%d base, %d queries, tree trained on %d vectors by tool_createdb)" % (d, n, qn, learn.shape[0]),
            "config": {"workload": "real data N=%d d=%d p=%d c1=%d c2=%d w=%d lineparts=%d, batch=%d queries, query(boundVectors=%d, boundBins=%d), k=%d"
                                   % (n, D, P, C1, C2, Wc, LP, qn, args.bv, args.bb, k),
                       "built_by": "product front-end: tool_createdb (createTree on the learn set, buildKBestDB chunks + CSR merge), dumps .ppqt + .bins read back here",
                       "train_s": t_train, "build_db_s": t_db, "mean_candidates": float(oc.to(torch.int64).float().mean()),
                       "engine": rec, "checker_on_same_index": orec, "id_lists_identical_frac": same,
                       "recall_definition": "cpu_version/tools/query.cpp:29-82: fraction of queries whose ground-truth nearest neighbour appears at rank < R"}}


def cpu_baseline_chunked(pkg, idx, w, meta, queries, args, out_idx, out_dist, k):
    """cpu_baseline for a chunk-built database of <= 20 M vectors: the oracle is loaded with the engine-built bins and codes
    (insert() is bit-identical to the build kernel, tests/test_gpu_parity.py) and timed on a bounded sample."""
    from oracle import Oracle
    o = Oracle(w["D"], w["P"], w["C1"], w["C2"], w["W"], w["LP"], heur_keep=max(args.bb, 1))
    if w.get("beyond_wrap"):  # the checker follows the engine's throughput-only mode (NOT reference behaviour: the reference enumerates 0 rows here)
        o.lift_tuple_wrap(max(args.bb, 1))
        o.set_heuristic(idx.heuristic(max(args.bb, 1)))
    heur_same = bool(np.array_equal(o.heuristic(max(args.bb, 1)), idx.heuristic(max(args.bb, 1))))
    o.set_codebooks(meta["cb1"], meta["cb2"])
    o.import_bins(meta["bin_ids"], meta["sizes"], meta["members"])
    o.import_codes(idx._keep[0].cpu().numpy().view(np.uint32))
    qh = queries.cpu().numpy()
    cores = usable_cores(o.max_threads())
    ns = min(qh.shape[0], 256)
    t = time.perf_counter()
    o.query_batch(qh[:ns], args.bv, args.bb, k, nthreads=cores)
    pass_t = time.perf_counter() - t
    reps = int(max(1, min(200, args.cpu_seconds / max(pass_t, 1e-6))))
    t = time.perf_counter()
    for _ in range(reps):
        o.query_batch(qh[:ns], args.bv, args.bb, k, nthreads=cores)
    cpu_t = time.perf_counter() - t
    o.set_sort_mode(1)
    ci, cd, cc = o.query_batch(qh[:ns], args.bv, args.bb, k, nthreads=cores)
    gi = out_idx[:ns].cpu().numpy().view(np.uint32)
    gd = out_dist[:ns].cpu().numpy()
    same = float(np.mean([np.array_equal(gi[i], ci[i]) and np.array_equal(gd[i].view(np.uint32), cd[i].view(np.uint32)) for i in range(ns)]))
    return {"value": reps * ns / cpu_t, "unit": "queries/sec", "cores": cores, "kind": "port",
            "sample": "%d passes over the first %d bench queries (%.1f s of work), same index, all host threads" % (reps, ns, cpu_t),
            "result_lists_identical_frac": same, "heuristic_table_identical": heur_same}


if __name__ == "__main__":
    main()
