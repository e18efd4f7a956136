"""Host-side cost of one pqt_query call (two kernel launches + bookkeeping): wall time of the issuing loop before the device is waited for.
Run from the repository root."""
import importlib, sys, os, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
pkg = importlib.import_module("product-quantization-tree_amd")
w = bench.WORKLOADS["sift1m"]
dev = torch.device("cuda", 0)
idx, base, meta = bench.build_index(pkg, w, 0)
idx.build_heuristic(500)
q = bench.sift_like(w["qn"], w["D"], 0xC0DE03, dev)
k = 100
oi = torch.empty((w["qn"], k), dtype=torch.int32, device=dev); od = torch.empty((w["qn"], k), dtype=torch.float32, device=dev); oc = torch.empty(w["qn"], dtype=torch.int32, device=dev)
s = torch.cuda.Stream(dev)
for tp in (0, 4, 1):
    idx.set_option("stage_timing", tp)
    for n in (8, 16, 32):
        for _ in range(4): idx.query_dev(q, 20000, 500, k, oi, od, oc, stream=s.cuda_stream)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n): idx.query_dev(q, 20000, 500, k, oi, od, oc, stream=s.cuda_stream)
        ti = time.perf_counter() - t
        torch.cuda.synchronize()
        tt = time.perf_counter() - t
        print("stage_timing %d: %2d calls issued in %.3f ms (%.1f us per call), done after %.3f ms (%.1f us per call)" % (tp, n, ti * 1e3, ti / n * 1e6, tt * 1e3, tt / n * 1e6))
