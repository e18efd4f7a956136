"""One rank's step of the 8-way range-sharded layout on ONE device, with and without two half batches in flight, under an injected
collective latency (no 8-GPU node in this pool): what the >= 6x projection of profiles/r03_shard8_one_device_* becomes once every step
pays three collectives.
Rank 0's shard (1/8 of a configs[2]-shape database built from the shard's side) runs the REAL step functions of sharding.py
(sharded_query / sharded_query_pipelined: pqt_traverse_bins on its query slice, pqt_query_shard_bins, pqt_merge_topk on its slice)
against a stand-in for torch.distributed whose collectives move the right bytes on the device (the bin lists of the other slices are
the true ones, traversed beforehand) and then hold the stream for `delay` microseconds (torch.cuda._sleep, calibrated) -- the launch
latency of an RCCL collective as the step sees it.  Same code path as bench.py --gpus 8 except for who fills the receive buffers.
    PQT_SHARD_WORKLOAD=synth10m|synth100m [PQT_SHARED_ROWS=0|1] python scripts/r05_pipeline_one_device.py   (round 5: + the shared-row pass switch)"""
import importlib, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module("product-quantization-tree_amd")
sharding = importlib.import_module("product-quantization-tree_amd.sharding")
wl = os.environ.get("PQT_SHARD_WORKLOAD", "synth10m")
w = bench.WORKLOADS[wl]
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
st = torch.cuda.Stream(dev); torch.cuda.set_stream(st)
n, world, k, qn = w["n_base"], 8, 100, w["qn"]
cb = bench.make_codebooks(w, dev)
queries = bench.sift_like(qn, w["D"], 0xC0DE03, dev)
# shard 0 of 8, built like a rank of the multi-GPU bench builds it; the other shards only contribute their per-bin counts
keys, counts, members, codes0 = [], [], None, None
D, LP = w["D"], w["LP"]
tmp = pkg.PqtIndex(w["D"], w["P"], w["C1"], w["C2"], w["W"], w["LP"], device=0)
tmp.set_codebooks(*cb)
for r in range(world):
    lo, hi = sharding.shard_range(r, world, n)
    bins = torch.empty(hi - lo, dtype=torch.int32, device=dev)
    codes = torch.empty((hi - lo, LP), dtype=torch.int32, device=dev)
    for ci, s0, m, a, b in bench.chunk_ranges(w, lo, hi):
        x = bench.sift_like(m, D, bench.CHUNK_SEED + 7919 * ci, dev)
        tmp.assign_encode_dev(x[a - s0:b - s0], bins[a - lo:b - lo], codes[a - lo:b - lo], stream=st.cuda_stream)
        del x
    torch.cuda.synchronize()
    kk_, cc_, mm_ = sharding.local_bin_lists(bins, lo)
    keys.append(kk_); counts.append(cc_)
    if r == 0:
        members, codes0 = mm_, codes
    else:
        del codes
    del bins
tmp.close()
uk, gs, low, ls = sharding.merge_bin_counts(keys, counts, 0)
sh = pkg.PqtIndex(w["D"], w["P"], w["C1"], w["C2"], w["W"], w["LP"], device=0)
sh.set_codebooks(*cb); sh.build_heuristic(4096)
sh.set_bins_local(uk.cpu().numpy(), gs.cpu().numpy(), low.cpu().numpy(), ls.cpu().numpy(), members.cpu().numpy(), n)
sh.set_lines_dev(codes0, 0)
if os.environ.get("PQT_SHARED_ROWS") is not None:
    sh.set_option("shared_rows", int(os.environ["PQT_SHARED_ROWS"]))  # round 5: the shared-row pass on the shard (-1 automatic, 0 off, 1 on)
# round 6: any other option of the shard handle, e.g. PQT_SHARD_OPTIONS="coop_rerank=1" or "shared_rows=1,sr_kernel=2" (set before the view is made: a view copies them)
for ov_ in filter(None, os.environ.get("PQT_SHARD_OPTIONS", "").split(",")):
    sh.set_option(ov_.split("=")[0], int(ov_.split("=")[1]))
view = sh.view()
for h_ in (sh, view):
    h_.set_option("stage_timing", 0)
eng, engv = sharding.PqtShardEngine(sh), sharding.PqtShardEngine(view)

# calibrate torch.cuda._sleep (cycles -> microseconds)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda._sleep(1000); torch.cuda.synchronize()
e0.record(st); torch.cuda._sleep(2_000_000); e1.record(st); torch.cuda.synchronize()
cyc_per_us = 2_000_000 / (e0.elapsed_time(e1) * 1e3)


class FakeDist:
    """Stand-in for torch.distributed on one device: rank 0 of `world`; every collective fills the receive buffer on the device
    (own contribution in every block; bin lists: the true lists of all slices) and then holds the stream for delay_us."""
    def __init__(self):
        self.delay_us, self.bins_full = 0.0, {}
    def get_rank(self):
        return 0
    def _hold(self):
        if self.delay_us > 0:
            torch.cuda._sleep(int(self.delay_us * cyc_per_us))
    def all_gather_into_tensor(self, out, inp):
        full = self.bins_full.get((out.shape[0], out.shape[1])) if out.dtype == torch.int64 and out.dim() == 2 else None
        if full is not None:
            out.copy_(full)
        else:
            out.view(world, -1).copy_(inp.reshape(1, -1).expand(world, -1))
        self._hold()
    def all_to_all_single(self, out, inp):
        # every "shard" sends this rank the rows of ITS slice (block 0 of the send buffer): the merged slice is then the same in every scheme
        out.view(world, -1).copy_(inp.view(world, -1)[0:1].expand(world, -1))
        self._hold()


fd = FakeDist()


def timed(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(st)
    for _ in range(reps): fn()
    b.record(st); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


out = {"shared_rows_on_the_shard": os.environ.get("PQT_SHARED_ROWS", "automatic"), "options": os.environ.get("PQT_SHARD_OPTIONS", ""), "workload": "%s: N=%d (configs[2] shape), rank 0 of an 8-way range-sharded run on ONE device, %d queries per batch, k=%d" % (wl, n, qn, k),
       "what": "per-rank step of sharding.sharded_query (one batch at a time) vs sharded_query_pipelined (two half batches in flight) vs BatchesInFlight (two whole batches in flight: "
               "consecutive steps alternate between the index and a view of it on two streams) with every "
               "collective holding its stream for delay_us after moving its bytes; unsharded = the whole database on this device",
       "sleep_cycles_per_us": cyc_per_us, "knobs": {}}
for bv, bb in ((20000, 500), (4096, 4096)):
    cap = sharding.bin_cap_for(bb)
    buf = sharding.ShardBuffers(world, qn, k, dev, bin_cap=cap)
    pbuf = sharding.PipelineBuffers(world, qn, k, dev, bin_cap=cap)
    # the true bin lists of every slice (any shard produces the same bytes): whole batch and the two halves
    for b_ in (buf, pbuf.halves[0], pbuf.halves[1]):
        a0 = 0 if b_ is not pbuf.halves[1] else pbuf.h
        nq_ = b_.qn
        full = torch.zeros_like(b_.bins_all)
        for s in range(world):
            a, b = min(s * b_.qs, nq_), min((s + 1) * b_.qs, nq_)
            if b > a:
                sh.traverse_bins_dev(queries[a0 + a:a0 + b], bv, bb, cap, full[a:b], stream=st.cuda_stream)
        torch.cuda.synchronize()
        fd.bins_full[(full.shape[0], full.shape[1])] = full
    res = {}
    fl = sharding.BatchesInFlight((eng, engv), world, qn, k, dev, bin_cap=cap)
    fd.bins_full[(fl.bufs[0].bins_all.shape[0], fl.bufs[0].bins_all.shape[1])] = fd.bins_full[(buf.bins_all.shape[0], buf.bins_all.shape[1])]

    def timed_inflight(reps=10):
        for _ in range(4): fl.step(fd, world, queries, bv, bb, k, traversal="sharded", rank=0)
        fl.wait(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st)
        for s_ in fl.streams: s_.wait_event(a)
        for _ in range(reps): fl.step(fd, world, queries, bv, bb, k, traversal="sharded", rank=0)
        fl.wait(); b.record(st); torch.cuda.synchronize()
        return a.elapsed_time(b) / reps

    for delay in (0.0, 20.0, 40.0, 80.0):
        fd.delay_us = delay
        t1 = timed(lambda: sharding.sharded_query(eng, fd, world, queries, bv, bb, k, buf, traversal="sharded", rank=0))
        t2 = timed(lambda: sharding.sharded_query_pipelined((eng, engv), fd, world, queries, bv, bb, k, pbuf, traversal="sharded", rank=0))
        t3 = timed_inflight()
        res["delay_%dus" % delay] = {"one_batch_ms": round(t1, 4), "two_half_batches_ms": round(t2, 4), "two_whole_batches_in_flight_ms": round(t3, 4),
                                     "gain_whole_batches": round(t1 / t3, 3)}
    out["knobs"]["%d_%d" % (bv, bb)] = res
    del buf, pbuf
# the denominator: the same database unsharded on this device (only when it fits comfortably)
if n <= 100_000_000 and not os.environ.get("PQT_SKIP_UNSHARDED"):
    sh.close()
    del codes0
    torch.cuda.empty_cache()
    idx, _, _ = bench.build_index(pkg, w, 0, codebooks=cb)
    idx.build_heuristic(4096)
    idx.set_option("stage_timing", 0)
    if os.environ.get("PQT_SHARED_ROWS_UNSHARDED") is not None:
        idx.set_option("shared_rows", int(os.environ["PQT_SHARED_ROWS_UNSHARDED"]))
    oi = torch.empty((qn, k), dtype=torch.int32, device=dev); od = torch.empty((qn, k), dtype=torch.float32, device=dev); oc = torch.empty(qn, dtype=torch.int32, device=dev)
    for bv, bb in ((20000, 500), (4096, 4096)):
        t = timed(lambda: idx.query_dev(queries, bv, bb, k, oi, od, oc, stream=st.cuda_stream))
        r_ = out["knobs"]["%d_%d" % (bv, bb)]
        r_["unsharded_step_ms"] = round(t, 4)
        for d_, e_ in r_.items():
            if isinstance(e_, dict):
                e_["projected_speedup_one_batch"] = round(t / e_["one_batch_ms"], 2)
                e_["projected_speedup_two_half_batches"] = round(t / e_["two_half_batches_ms"], 2)
                e_["projected_speedup_two_whole_batches"] = round(t / e_["two_whole_batches_in_flight_ms"], 2)
print(json.dumps(out, indent=1))
