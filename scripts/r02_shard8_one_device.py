"""The 8-GPU layout's per-rank work, measured on ONE device: 8 range shards of the 10 M-vector configs[2]-shape index (built from
the shard's side like the multi-GPU bench does), each queried with the full 10 k-query batch through pqt_query_shard; per-shard stage
times, the merge of one query slice (what a rank merges after the all-to-all) and of all queries.  No collective is timed here."""
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
shards = []
for r, (lo, hi) in enumerate(ranges):
    uk, gs, low, ls = sharding.merge_bin_counts([l[0] for l in local], [l[1] for l in local], r)
    sh = pkg.PqtIndex(w["D"], w["P"], w["C1"], w["C2"], w["W"], w["LP"], device=0)
    sh.set_codebooks(meta["cb1"], meta["cb2"]); sh.build_heuristic(4096)
    sh.set_bins_local(uk.cpu().numpy(), gs.cpu().numpy(), low.cpu().numpy(), ls.cpu().numpy(), local[r][2].cpu().numpy(), n)
    sh.set_lines_dev(codes[lo:hi], lo)
    shards.append(sh)
torch.cuda.synchronize()
out = {"workload": "N=%d (configs[2] shape), 8 range shards on one device, %d queries per batch, k=%d" % (n, qn, k), "knobs": {}}
oi = torch.empty((qn, k), dtype=torch.int32, device=dev); od = torch.empty((qn, k), dtype=torch.float32, device=dev); oc = torch.empty(qn, dtype=torch.int32, device=dev)
for bv, bb in ((20000, 500), (4096, 4096)):
    for _ in range(5): idx.query_dev(queries, bv, bb, k, oi, od, oc, stream=st.cuda_stream)
    torch.cuda.synchronize()
    h = idx.stage_ms_history(3).mean(0)
    pack = torch.empty((world, 3, qn, k), dtype=torch.int32, device=dev); Cc = torch.empty((world, qn), dtype=torch.int32, device=dev)
    per = []
    for s, sh in enumerate(shards):
        for _ in range(5): sh.query_shard_dev(queries, bv, bb, k, pack[s, 0], pack[s, 1].view(torch.float32), pack[s, 2], Cc[s], stream=st.cuda_stream)
        torch.cuda.synchronize()
        hs = sh.stage_ms_history(3).mean(0)
        per.append({"traverse_ms": round(float(hs[1]), 4), "rerank_select_ms": round(float(hs[3]), 4), "local_candidates_per_query": sh.stats()["candidates"] / qn})
    oI = torch.empty((qn, k), dtype=torch.int32, device=dev); oD = torch.empty((qn, k), dtype=torch.float32, device=dev)
    def tmerge(nq):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): shards[0].merge_topk_dev(world, nq, k, pack[0, 0], pack[0, 1].view(torch.float32), pack[0, 2], oI, oD, stream=st.cuda_stream, shard_stride=3 * qn * k)
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / 20 * 1e3
    m_all, m_slice = tmerge(qn), tmerge(qn // world)
    same = bool(torch.equal(oI.view(torch.int32), oi) ) if False else None
    shards[0].merge_topk_dev(world, qn, k, pack[0, 0], pack[0, 1].view(torch.float32), pack[0, 2], oI, oD, sync=True, shard_stride=3 * qn * k)
    same = bool(torch.equal(oI, oi) and torch.equal(oD.view(torch.int32), od.view(torch.int32)))
    out["knobs"]["%d_%d" % (bv, bb)] = {"unsharded": {"traverse_ms": round(float(h[1]), 4), "rerank_select_ms": round(float(h[3]), 4)}, "per_shard": per,
        "per_shard_mean_ms": round(float(np.mean([p["traverse_ms"] + p["rerank_select_ms"] for p in per])), 4),
        "merge_all_queries_ms": round(m_all, 4), "merge_one_slice_ms": round(m_slice, 4), "merged_identical_to_unsharded": same}
print(json.dumps(out, indent=1))
