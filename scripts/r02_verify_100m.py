"""One-off full-size parity check at BASELINE configs[2] size (100 M vectors, d=128 p=4 c1=c2=64 lineparts=32): too slow for the
test-suite (the index build alone is ~20 s, the oracle needs the 12.8 GB line store on the host), run by hand on the GPU box:
    python scripts/r02_verify_100m.py > gpurun_out/r02_verify_100m.json
Checks: 64 queries of the bench batch against the oracle loaded with the same index (ids and distance bits, both knob sets);
4000 queries: the three rerank schedules agree; band-filtered exact rerank (MODE 2, bin runs) == workgroup-per-query exact kernel == candidate-list variant."""
import importlib, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from oracle import Oracle

pkg = importlib.import_module("product-quantization-tree_amd")
w = bench.WORKLOADS[os.environ.get("PQT_VERIFY_WORKLOAD", "synth100m")]
dev = torch.device("cuda", 0)
t0 = time.time()
idx, base, meta = bench.build_index(pkg, w, 0)
idx.build_heuristic(4096)
queries = bench.sift_like(4000, w["D"], 0xC0DE03, dev)  # more than 256 CUs x 12 wavefronts: the dynamic rerank schedules are in play
res = {"workload": "N=%d d=128 p=4 c1=64 c2=64 w=1 lineparts=32" % w["n_base"], "build_s": round(time.time() - t0, 1), "n_bins": meta["n_bins"], "max_bin": meta["max_bin"]}


def run(q, bv, bb, k=100):
    qn = q.shape[0]
    oi = torch.empty((qn, k), dtype=torch.int32, device=dev); od = torch.empty((qn, k), dtype=torch.float32, device=dev); oc = torch.empty(qn, dtype=torch.int32, device=dev)
    idx.query_dev(q, bv, bb, k, oi, od, oc, sync=True)
    return oi.cpu().numpy().view(np.uint32), od.cpu().numpy(), oc.cpu().numpy().view(np.uint32)


for bv, bb in ((20000, 500), (4096, 4096)):
    a = run(queries, bv, bb)
    fb = idx.stats()["filter_fallbacks"]
    same = {}
    for opt, val, back in (("exact_filter", 0, 1), ("bin_runs", 0, -1), ("balance", 0, -1), ("balance", 1, -1), ("balance", 2, -1)):
        idx.set_option(opt, val)
        b = run(queries, bv, bb)
        idx.set_option(opt, back)
        same[opt + "=%d" % val] = bool(np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)) and np.array_equal(a[2], b[2]))
    res["knobs_%d_%d" % (bv, bb)] = {"mean_candidates": float(a[2].mean()), "filter_fallbacks": int(fb), "identical_to_variant": same}

o = Oracle(w["D"], w["P"], w["C1"], w["C2"], w["W"], w["LP"], heur_keep=4096)
res["heuristic_table_identical"] = bool(np.array_equal(o.heuristic(4096), idx.heuristic(4096)))
o.set_codebooks(meta["cb1"], meta["cb2"])
o.import_bins(meta["bin_ids"], meta["sizes"], meta["members"])
o.import_codes(idx._keep[0].cpu().numpy().view(np.uint32))
o.set_sort_mode(1)
qh = queries[:64].cpu().numpy()
for bv, bb in ((20000, 500), (4096, 4096)):
    ids, dist, cnt = run(queries[:64], bv, bb)
    ok = 0
    for i in range(64):
        s_ids, s_d = o.query(qh[i], bv, bb)
        n = min(100, len(s_ids))
        ok += int(int(cnt[i]) == len(s_ids) and np.array_equal(ids[i, :n], s_ids[:n]) and np.array_equal(dist[i, :n].view(np.uint32), s_d[:n].view(np.uint32)))
    res["knobs_%d_%d" % (bv, bb)]["oracle_identical_of_64"] = ok
print(json.dumps(res, indent=1))
