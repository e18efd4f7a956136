"""What the per-kernel start/stop events cost per 10 k-query call: wall time per call with stage_timing 1 / 0."""
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
pkg = importlib.import_module("product-quantization-tree_amd"); pkg.lib()
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
st = torch.cuda.Stream(dev); torch.cuda.set_stream(st)
w = bench.WORKLOADS["sift1m"]
queries = bench.sift_like(w["qn"], w["D"], 0xC0DE03, dev)
qn, k = w["qn"], 100
oi = torch.empty((qn, k), dtype=torch.int32, device=dev); od = torch.empty((qn, k), dtype=torch.float32, device=dev); oc = torch.empty(qn, dtype=torch.int32, device=dev)
idx, base, meta = bench.build_index(pkg, w, 0); idx.build_heuristic(500); del base
for rnd in range(3):
    for tm in (1, 0):
        idx.set_option("stage_timing", tm)
        for _ in range(10): idx.query_dev(queries, 20000, 500, k, oi, od, oc, stream=st.cuda_stream)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200): idx.query_dev(queries, 20000, 500, k, oi, od, oc, stream=st.cuda_stream)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
        print("stage_timing=%d: %.4f ms per call = %.1f M queries/s" % (tm, dt * 1e3, qn / dt / 1e6), flush=True)
