"""The 8-GPU layout's per-rank work, measured on ONE device (no 8-GPU node in this pool): 8 range shards of a configs[2]-shape
index built from the shard's side like the multi-GPU bench does.  Per shard and knob set:
  replicated traversal : pqt_query_shard over the full 10 k-query batch (traversal + rerank/select stage times)
  query-sharded        : pqt_traverse_bins over the shard's query slice (QN/8 queries) + pqt_query_shard_bins over the full batch
                         (distance tables + bin-list resolution stage, rerank/select stage), results identical to the above
plus the merge of one query slice / of all queries, and the unsharded index on the same device as the denominator.  No
collective is timed here (the all-gather of the bin lists moves (128 + 1) x 8 B per query: 10 MB per batch in total; (256 + 1) x 8 B at bound_bins > 512).
    PQT_SHARD_WORKLOAD=synth10m|synth100m python scripts/r03_shard8_one_device.py
    PQT_TSTAMP=1 ... additionally prints the per-query phase clocks of shard 0's rerank (instrumented kernel: slower)"""
import importlib, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module("product-quantization-tree_amd")
sharding = importlib.import_module("product-quantization-tree_amd.sharding")
w = bench.WORKLOADS[os.environ.get("PQT_SHARD_WORKLOAD", "synth10m")]
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
st = torch.cuda.Stream(dev); torch.cuda.set_stream(st)
idx, base, meta = bench.build_index(pkg, w, 0)
idx.build_heuristic(4096)
n, world, k, qn = w["n_base"], 8, 100, w["qn"]
queries = bench.sift_like(qn, w["D"], 0xC0DE03, dev)
bin_of_vec = torch.empty(n, dtype=torch.int64, device=dev)
bin_of_vec[torch.from_numpy(meta["members"].astype(np.int64)).to(dev)] = \
    torch.repeat_interleave(torch.from_numpy(meta["bin_ids"].astype(np.int64)), torch.from_numpy(meta["sizes"].astype(np.int64))).to(dev)
codes = idx._keep[0]
ranges = [sharding.shard_range(r, world, n) for r in range(world)]
local = [sharding.local_bin_lists(bin_of_vec[lo:hi], lo) for lo, hi in ranges]
del bin_of_vec
n_sh = int(os.environ.get("PQT_SHARDS_MEASURED", "8"))  # 100 M: measure fewer shards to save build time
shards = []
for r, (lo, hi) in enumerate(ranges[:n_sh]):
    uk, gs, low, ls = sharding.merge_bin_counts([l[0] for l in local], [l[1] for l in local], r)
    sh = pkg.PqtIndex(w["D"], w["P"], w["C1"], w["C2"], w["W"], w["LP"], device=0)
    sh.set_codebooks(meta["cb1"], meta["cb2"]); sh.build_heuristic(4096)
    sh.set_bins_local(uk.cpu().numpy(), gs.cpu().numpy(), low.cpu().numpy(), ls.cpu().numpy(), local[r][2].cpu().numpy(), n)
    sh.set_lines_dev(codes[lo:hi], lo)
    shards.append(sh)
torch.cuda.synchronize()
out = {"workload": "N=%d (configs[2] shape), 8 range shards (%d measured) on one device, %d queries per batch, k=%d" % (n, n_sh, qn, k), "knobs": {}}
oi = torch.empty((qn, k), dtype=torch.int32, device=dev); od = torch.empty((qn, k), dtype=torch.float32, device=dev); oc = torch.empty(qn, dtype=torch.int32, device=dev)
cap_env, qs = os.environ.get("PQT_SHARD_BINCAP"), (qn + world - 1) // world  # bin-list capacity: default = what sharding.bin_cap_for picks per knob set


def timed(fn, reps=8):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps): fn()
    e1.record(st); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for bv, bb in ((20000, 500), (4096, 4096)):
    cap = int(cap_env) if cap_env else sharding.bin_cap_for(bb)
    step1 = timed(lambda: idx.query_dev(queries, bv, bb, k, oi, od, oc, stream=st.cuda_stream))
    h = idx.stage_ms_history(3).mean(0)
    pack = torch.empty((world, 3, qn, k), dtype=torch.int32, device=dev); Cc = torch.empty((world, qn), dtype=torch.int32, device=dev)
    pack2 = torch.empty((3, qn, k), dtype=torch.int32, device=dev); C2 = torch.empty(qn, dtype=torch.int32, device=dev)
    # the all-gathered bin lists: slice s traversed by shard s (any shard gives the same bytes; with fewer shards measured the
    # remaining slices come from shard 0)
    bins_all = torch.zeros((world * qs, cap + 1), dtype=torch.int64, device=dev)
    for s in range(world):
        a, b = min(s * qs, qn), min((s + 1) * qs, qn)
        shards[s if s < n_sh else 0].traverse_bins_dev(queries[a:b], bv, bb, cap, bins_all[a:b], stream=st.cuda_stream)
    torch.cuda.synchronize()
    overflow = int(((bins_all[:qn, cap] & 0xffffffff) == 0xffffffff).sum())
    nb = (bins_all[:qn, cap] & 0xffffffff).float()
    per = []
    ov = os.environ.get("PQT_SHARD_OVERLAP")  # e.g. "1": also time the calls without stage events and with "overlap" = that value
    for s, sh in enumerate(shards):
        t_rep = timed(lambda: sh.query_shard_dev(queries, bv, bb, k, pack[s, 0], pack[s, 1].view(torch.float32), pack[s, 2], Cc[s], stream=st.cuda_stream))
        hs = sh.stage_ms_history(3).mean(0)
        a, b = min(s * qs, qn), min((s + 1) * qs, qn)
        scratch = torch.zeros((qs, cap + 1), dtype=torch.int64, device=dev)
        t_trav = timed(lambda: sh.traverse_bins_dev(queries[a:b], bv, bb, cap, scratch, stream=st.cuda_stream))
        t_bins = timed(lambda: sh.query_shard_bins_dev(queries, bv, bb, k, bins_all, cap, pack2[0], pack2[1].view(torch.float32), pack2[2], C2, stream=st.cuda_stream))
        hb = sh.stage_ms_history(3).mean(0)
        path_bins = sh.last_path()
        untimed = None
        if ov is not None:
            untimed = {}
            for name, val in (("one_piece", 0), ("overlap_%s" % ov, int(ov))):
                sh.set_option("stage_timing", 0); sh.set_option("overlap", val)
                untimed[name] = {"replicated_step_ms": round(timed(lambda: sh.query_shard_dev(queries, bv, bb, k, pack[s, 0], pack[s, 1].view(torch.float32), pack[s, 2], Cc[s], stream=st.cuda_stream)), 4),
                                 "path_replicated": sh.last_path(),
                                 "bins_step_ms": round(timed(lambda: sh.query_shard_bins_dev(queries, bv, bb, k, bins_all, cap, pack2[0], pack2[1].view(torch.float32), pack2[2], C2, stream=st.cuda_stream)), 4),
                                 "path_bins": sh.last_path()}
            sh.set_option("stage_timing", 1); sh.set_option("overlap", -1)
        same = bool(torch.equal(pack2[0], pack[s, 0]) and torch.equal(pack2[1], pack[s, 1]) and torch.equal(pack2[2], pack[s, 2]) and torch.equal(C2, Cc[s]))
        per.append({"local_candidates_per_query": sh.stats()["candidates"] / qn,
                    "replicated": {"step_ms": round(t_rep, 4), "traverse_ms": round(float(hs[1]), 4), "rerank_select_ms": round(float(hs[3]), 4)},
                    "query_sharded": {"traverse_slice_ms": round(t_trav, 4), "step_ms": round(t_bins, 4), "tables_resolve_ms": round(float(hb[1]), 4),
                                      "rerank_select_ms": round(float(hb[3]), 4), "per_rank_ms": round(t_trav + t_bins, 4), "identical_to_replicated": same,
                                      "path": path_bins}, "without_stage_events": untimed})
    oI = torch.empty((qn, k), dtype=torch.int32, device=dev); oD = torch.empty((qn, k), dtype=torch.float32, device=dev)
    res = {"unsharded": {"step_ms": round(step1, 4), "traverse_ms": round(float(h[1]), 4), "rerank_select_ms": round(float(h[3]), 4)}, "per_shard": per,
           "bin_lists": {"mean_bins_per_query": float(nb[nb < 4e9].mean()), "max": float(nb[nb < 4e9].max()), "overflowed_queries": overflow},
           "per_rank_ms_replicated": round(float(np.mean([p["replicated"]["step_ms"] for p in per])), 4),
           "per_rank_ms_query_sharded": round(float(np.mean([p["query_sharded"]["per_rank_ms"] for p in per])), 4)}
    res["speedup_replicated"] = round(step1 / res["per_rank_ms_replicated"], 3)
    res["speedup_query_sharded"] = round(step1 / res["per_rank_ms_query_sharded"], 3)
    if n_sh == world:
        m_all = timed(lambda: shards[0].merge_topk_dev(world, qn, k, pack[0, 0], pack[0, 1].view(torch.float32), pack[0, 2], oI, oD, stream=st.cuda_stream, shard_stride=3 * qn * k))
        m_slice = timed(lambda: shards[0].merge_topk_dev(world, qn // world, k, pack[0, 0], pack[0, 1].view(torch.float32), pack[0, 2], oI, oD, stream=st.cuda_stream, shard_stride=3 * qn * k))
        shards[0].merge_topk_dev(world, qn, k, pack[0, 0], pack[0, 1].view(torch.float32), pack[0, 2], oI, oD, sync=True, shard_stride=3 * qn * k)
        idx.query_dev(queries, bv, bb, k, oi, od, oc, stream=st.cuda_stream); torch.cuda.synchronize()
        res.update({"merge_all_queries_ms": round(m_all, 4), "merge_one_slice_ms": round(m_slice, 4),
                    "merged_identical_to_unsharded": bool(torch.equal(oI, oi) and torch.equal(oD.view(torch.int32), od.view(torch.int32)))})
    if os.environ.get("PQT_TSTAMP"):
        import ctypes
        ts = np.zeros((qn, 24), np.uint64)
        shards[0].query_shard_bins_dev(queries, bv, bb, k, bins_all, cap, pack2[0], pack2[1].view(torch.float32), pack2[2], C2, stream=st.cuda_stream, sync=True)
        if pkg.lib().pqt_debug_tstamps(shards[0].h, ts.ctypes.data, qn) == 0:
            r = ts[:, 9:20].astype(np.int64)
            med = lambda c: int(np.median(c))
            res["shard0_rerank_clocks_median"] = {"total": med(r[:, 4] & 0xffffffff), "setup": med(r[:, 7]), "row_wait": med(r[:, 1]), "adc_filter": med(r[:, 2]), "flush": med(r[:, 3]),
                                                  "band_reevaluation": med(r[:, 8]), "output": med(r[:, 9]), "candidates": med(r[:, 10])}
            # wavefront-slot timeline on the 100 MHz wall clock: how much of the launch the slots are busy, and how ragged its end is
            start = (ts[:, 13] >> np.uint64(32)).astype(np.int64); end = (ts[:, 14] >> np.uint64(32)).astype(np.int64); slot = (ts[:, 14] & np.uint64(0xffff)).astype(np.int64)
            ns = int(slot.max()) + 1; busy = np.zeros(ns); last = np.zeros(ns); cnt = np.zeros(ns); t0 = start.min()
            np.add.at(busy, slot, end - start); np.add.at(cnt, slot, 1); np.maximum.at(last, slot, end - t0)
            res["shard0_rerank_timeline_us"] = {"slots": ns, "span": float(end.max() - t0) / 100, "slot_busy_mean": float(busy.mean()) / 100, "queries_per_slot_min_max": [int(cnt.min()), int(cnt.max())],
                                                "slot_last_end_p10_med_p90_max": [float(np.percentile(last, x)) / 100 for x in (10, 50, 90, 100)],
                                                "query_us_p10_med_p90_max": [float(np.percentile(end - start, x)) / 100 for x in (10, 50, 90, 100)]}
    if os.environ.get("PQT_SHARD_DUMP"):  # results of this library for a cross-library comparison (scripts/r04_run14.sh)
        idx.query_dev(queries, bv, bb, k, oi, od, oc, stream=st.cuda_stream)
        shards[0].query_shard_bins_dev(queries, bv, bb, k, bins_all, cap, pack2[0], pack2[1].view(torch.float32), pack2[2], C2, stream=st.cuda_stream, sync=True)
        torch.cuda.synchronize()
        np.savez(os.environ["PQT_SHARD_DUMP"] + "_%d_%d.npz" % (bv, bb), oi=oi.cpu().numpy(), od=od.view(torch.int32).cpu().numpy(), oc=oc.cpu().numpy(),
                 si=pack2[0].cpu().numpy(), sd=pack2[1].cpu().numpy(), sp=pack2[2].cpu().numpy(), sc=C2.cpu().numpy())
    out["knobs"]["%d_%d" % (bv, bb)] = res
print(json.dumps(out, indent=1))
