"""Kernel times of one call against the number of queries in it (headline index): does a launch of 10 k single-wavefront workgroups fill the
device?  Run from the repository root."""
import importlib, sys, os
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
pkg = importlib.import_module("product-quantization-tree_amd")
w = bench.WORKLOADS["sift1m"]
dev = torch.device("cuda", 0)
idx, base, meta = bench.build_index(pkg, w, 0)
idx.build_heuristic(500)
k = 100
s = torch.cuda.Stream(dev)
for qn in (2500, 5000, 10000, 20000, 40000):
    q = bench.sift_like(qn, w["D"], 0xC0DE03, dev)
    oi = torch.empty((qn, k), dtype=torch.int32, device=dev); od = torch.empty((qn, k), dtype=torch.float32, device=dev); oc = torch.empty(qn, dtype=torch.int32, device=dev)
    idx.set_option("stage_timing", 1)
    for _ in range(12): idx.query_dev(q, 20000, 500, k, oi, od, oc, stream=s.cuda_stream)
    torch.cuda.synchronize()
    h = idx.stage_ms_history(8).mean(0)
    print("%6d queries: traverse %.4f ms (%.2f us per 1000), rerank %.4f ms (%.2f us per 1000)  path %s" % (qn, h[1], h[1] / qn * 1e6, h[3], h[3] / qn * 1e6, idx.last_path()), flush=True)
