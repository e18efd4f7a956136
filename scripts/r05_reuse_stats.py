"""Round 5: where the intra-batch row reuse of the HBM configurations sits (VERDICT r04 "missing" #1).
For one standard batch of a chunk-built workload: every (query, included populated bin) pair -> per distinct bin n = rows, m = queries
that include it.  Prints a table of candidate reads (n * m) and distinct rows (n) by (n bucket, m bucket), the pairs per query, and what a
bin-major pass would move: rows once per bin + one L1virt copy (4 * LP * C1 bytes) per pair + 8 bytes per candidate (filter key out and in).
usage: python scripts/r05_reuse_stats.py [--workload synth100m] [--out gpurun_out/r05_reuse_stats.json]"""
import argparse
import importlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="synth100m")
    ap.add_argument("--out", default="gpurun_out/r05_reuse_stats.json")
    a = ap.parse_args()
    pkg = importlib.import_module("product-quantization-tree_amd")
    w = bench.WORKLOADS[a.workload]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    st = torch.cuda.Stream(dev)
    torch.cuda.set_stream(st)
    idx, base, meta = bench.build_index(pkg, w, 0)
    del base
    qn = w["qn"]
    queries = bench.sift_like(qn, w["D"], 0xC0DE03, dev)
    W = {"n": w["n_base"], "queries": queries}
    stream = st.cuda_stream
    sizes = torch.from_numpy(meta["sizes"].astype(np.int64)).to(dev)
    members = torch.from_numpy(meta["members"].astype(np.int64)).to(dev)
    nb = sizes.numel()
    bin_of_id = torch.empty(W["n"], dtype=torch.int32, device=dev)
    bin_of_id[members] = torch.repeat_interleave(torch.arange(nb, device=dev, dtype=torch.int32), sizes)
    del members
    res = {"workload": a.workload, "n": W["n"], "n_bins": int(nb), "max_bin": int(sizes.max()), "mean_bin": float(sizes.float().mean())}
    for bv, bb in ((20000, 500), (4096, 4096)):
        idx.build_heuristic(bb)
        cap = int(bv + meta["max_bin"] + 64)
        oi = torch.empty((qn, cap), dtype=torch.int32, device=dev)
        od = torch.empty((qn, cap), dtype=torch.float32, device=dev)
        oc = torch.empty(qn, dtype=torch.int32, device=dev)
        idx.query_candidates_dev(W["queries"], bv, bb, cap, oi, od, oc, stream=stream, sync=True)
        del od
        pair_keys = []
        for q0 in range(0, qn, 1000):  # (query, bin) pairs, 1000 queries at a time
            sl = oi[q0:q0 + 1000]
            valid = torch.arange(cap, device=dev)[None, :] < oc[q0:q0 + 1000, None].clamp(max=cap)
            qq = (torch.arange(q0, q0 + sl.shape[0], device=dev, dtype=torch.int64)[:, None]).expand_as(sl)[valid]
            bb_ = bin_of_id[sl[valid].long()].long()
            pair_keys.append(torch.unique(qq * (1 << 32) + bb_))
        del oi
        pk = torch.cat(pair_keys)
        pq, pb = pk >> 32, pk & 0xffffffff
        per_q = torch.bincount(pq, minlength=qn)
        m = torch.bincount(pb, minlength=nb)  # queries per bin
        touched = m > 0
        n_t, m_t = sizes[touched], m[touched]
        cand = int((n_t * m_t).sum())
        distinct = int(n_t.sum())
        nbk = [1, 64, 128, 256, 512, 1024, 2048, 4096, 1 << 30]
        mbk = [1, 2, 3, 5, 9, 17, 33, 65, 1 << 30]
        table = []
        for i in range(len(nbk) - 1):
            row = []
            for j in range(len(mbk) - 1):
                s = (n_t >= nbk[i]) & (n_t < nbk[i + 1]) & (m_t >= mbk[j]) & (m_t < mbk[j + 1])
                row.append({"bins": int(s.sum()), "reads": int((n_t[s] * m_t[s]).sum()), "rows": int(n_t[s].sum())})
            table.append(row)
        l1 = 4 * w["LP"] * w["C1"]
        rowb = 4 * w["LP"] + 4
        out = {"pairs": int(pk.numel()), "pairs_per_query_mean": float(per_q.float().mean()), "pairs_per_query_max": int(per_q.max()),
               "distinct_bins": int(touched.sum()), "candidate_reads": cand, "distinct_rows": distinct, "reuse": cand / max(distinct, 1),
               "m_mean_over_bins": float(m_t.float().mean()), "m_max": int(m_t.max()),
               "n_buckets": nbk, "m_buckets": mbk, "table_n_by_m": table,
               "bytes_query_major": cand * rowb,
               "bytes_bin_major": {"rows_once": distinct * rowb, "l1virt_per_pair": int(pk.numel()) * l1, "keys_out_and_in": cand * 8,
                                   "total": distinct * rowb + int(pk.numel()) * l1 + cand * 8}}
        # the same with bin-major only for bins of >= nmin rows and >= 2 queries, query-major for the rest
        hyb = {}
        for nmin in (1, 64, 128, 256, 512):
            s = (n_t >= nmin) & (m_t >= 2)
            hyb["n>=%d,m>=2" % nmin] = {"bins": int(s.sum()), "reads_covered": int((n_t[s] * m_t[s]).sum()),
                                         "bytes": int((n_t[~s] * m_t[~s]).sum()) * rowb + int(n_t[s].sum()) * rowb + int(m_t[s].sum()) * l1 + int((n_t[s] * m_t[s]).sum()) * 8}
        out["hybrid_bytes"] = hyb
        res["knobs_%d_%d" % (bv, bb)] = out
        print("[%d,%d] pairs %d (%.1f per query, max %d)  distinct bins %d  reads %d  distinct rows %d  reuse %.2f" %
              (bv, bb, out["pairs"], out["pairs_per_query_mean"], out["pairs_per_query_max"], out["distinct_bins"], cand, distinct, out["reuse"]), flush=True)
        print("   bytes query-major %.2f GB; bin-major %.2f GB (rows %.2f + L1virt %.2f + keys %.2f)" %
              (cand * rowb / 1e9, out["bytes_bin_major"]["total"] / 1e9, distinct * rowb / 1e9, int(pk.numel()) * l1 / 1e9, cand * 8 / 1e9), flush=True)
        print("   reads by n (rows) x m (queries):  m buckets", mbk[:-1])
        for i, row in enumerate(table):
            print("   n>=%5d: " % nbk[i] + " ".join("%6.1fM" % (c["reads"] / 1e6) for c in row))
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)
    idx.close()


if __name__ == "__main__":
    main()
