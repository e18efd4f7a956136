"""How much of a candidate's 128-byte code row does the exact rerank really need?  (study for the early-abandon ADC)
Every ADC term is the squared distance from the query sub-vector to a point on a line, so the partial sum over the first g planes
(4 line parts each) is a LOWER BOUND of the final distance; a candidate whose partial sum already exceeds tau (the current
256th-best distance of its query, MODE 2 keeps 256) can be dropped without reading its remaining planes.  For a sample of queries
this script replays the candidates in visiting order with a running tau (updated every 128 candidates, like the kernel's batches)
and reports, per checkpoint scheme, the fraction of code bytes still read and the fraction of ADC terms still evaluated.
    PQT_STUDY_WORKLOAD=synth10m|synth100m python scripts/r03_abandon_study.py"""
import importlib, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module("product-quantization-tree_amd")
w = bench.WORKLOADS[os.environ.get("PQT_STUDY_WORKLOAD", "synth10m")]
dev = torch.device("cuda", 0)
idx, base, meta = bench.build_index(pkg, w, 0)
idx.build_heuristic(4096)
idx.set_option("bin_runs", 0)  # materialise the candidate lists
NQ = 48
queries = bench.sift_like(w["qn"], w["D"], 0xC0DE03, dev)[:NQ].contiguous()
codes = idx._keep[0]  # [n][LP] int32, id order
LP, C1 = w["LP"], w["C1"]
coarse = torch.from_numpy(idx.coarse()).to(dev)  # [LP][C1][C1]
out = {"workload": "N=%d configs[2] shape, %d queries" % (w["n_base"], NQ), "knobs": {}}
for bv, bb in ((20000, 500), (4096, 4096)):
    oi = torch.empty((NQ, 100), dtype=torch.int32, device=dev); od = torch.empty((NQ, 100), dtype=torch.float32, device=dev); oc = torch.empty(NQ, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    idx.query_dev(queries, bv, bb, 100, oi, od, oc, sync=True)
    dbg = idx.debug_read(NQ, segs=False, dists=False)
    schemes = {"2,4,8": (2, 4, 8), "1,2,4,8": (1, 2, 4, 8), "4,8": (4, 8), "1,2,3,4,6,8": (1, 2, 3, 4, 6, 8), "2,8": (2, 8), "3,8": (3, 8)}
    acc = {k: [0.0, 0.0] for k in schemes}
    tot = 0
    surv_hist = np.zeros(9)
    for qi in range(NQ):
        n = int(dbg["ncand"][qi])
        ids = torch.from_numpy(dbg["cand_idx"][qi, :n].astype(np.int64)).to(dev)
        virt = torch.from_numpy(dbg["l1virt"][qi]).to(dev)  # [LP][C1]
        cw = codes[ids].to(torch.int64) & 0xffffffff  # [n][LP]
        A, B = cw & 0xff, (cw >> 8) & 0xff
        lam = (cw >> 16).float() * (8.0 / 65536.0) - 4.0
        pidx = torch.arange(LP, device=dev)[None, :].expand(n, LP)
        b_ = virt[pidx, A]; a_ = virt[pidx, B]; c_ = coarse[pidx, A, B]
        term = b_ + lam * lam * c_ + lam * (a_ - b_ - c_)  # the squared distance to the point on the line, >= 0 up to rounding
        part = torch.cumsum(term, 1)  # [n][LP]
        plane = part[:, 3::4]  # partial sums after each of the 8 planes
        d = plane[:, -1]
        # running tau: the 256th smallest distance among the candidates of earlier batches of 128
        tau = torch.full((n,), float("inf"), device=dev)
        for b0 in range(128, n, 128):
            if b0 >= 256:
                tau[b0:b0 + 128] = torch.kthvalue(d[:b0], 256).values
        for name, cps in schemes.items():
            alive = torch.ones(n, dtype=torch.bool, device=dev)
            planes_read = torch.zeros(n, device=dev)
            prev = 0
            for cp in cps:
                planes_read += alive.float() * (cp - prev)
                if cp < 8:
                    alive &= plane[:, cp - 1] <= tau
                prev = cp
            acc[name][0] += float(planes_read.sum()) / 8.0
            acc[name][1] += float(planes_read.sum()) / 8.0  # terms evaluated track the planes read
        tot += n
        for g in range(1, 9):
            surv_hist[g] += float((plane[:, g - 1] <= tau).sum())
    out["knobs"]["%d_%d" % (bv, bb)] = {"mean_candidates": tot / NQ,
                                       "fraction_alive_after_plane": {str(g): surv_hist[g] / tot for g in range(1, 9)},
                                       "fraction_of_code_bytes_read": {k: v[0] / tot for k, v in acc.items()}}
print(json.dumps(out, indent=1))
