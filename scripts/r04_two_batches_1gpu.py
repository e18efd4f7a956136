"""One device, the headline workload: steps issued one batch at a time on one stream against two WHOLE batches in flight (consecutive
steps alternate between the index and a view of it on two streams).  Prints ms per step for both.  Run from the repository root."""
import importlib, sys, os, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
pkg = importlib.import_module("product-quantization-tree_amd")
w = bench.WORKLOADS["sift1m"]
dev = torch.device("cuda", 0)
idx, base, meta = bench.build_index(pkg, w, 0)
idx.build_heuristic(500)
views = [idx.view(), idx.view(), idx.view()]
q = bench.sift_like(w["qn"], w["D"], 0xC0DE03, dev)
k = 100
def bufs(n): return (torch.empty((n, k), dtype=torch.int32, device=dev), torch.empty((n, k), dtype=torch.float32, device=dev), torch.empty(n, dtype=torch.int32, device=dev))
b = [bufs(w["qn"]) for _ in range(4)]
st = [torch.cuda.Stream(dev) for _ in range(4)]
hs = [idx] + views
for name, nslot, tp in (("one batch at a time", 1, 0), ("two batches in flight", 2, 0), ("two, events every 4th call", 2, 4), ("two, events every 16th call", 2, 16), ("one batch at a time", 1, 0), ("two batches in flight", 2, 0), ("two, events every 4th call", 2, 4), ("two, events every 16th call", 2, 16)):
    for h in hs: h.set_option("stage_timing", tp)
    def step(i):
        s = i % nslot
        hs[s].query_dev(q, 20000, 500, k, b[s][0], b[s][1], b[s][2], stream=st[s].cuda_stream)
    for i in range(6): step(i)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(48): step(i)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / 48 * 1e3
    print("%-24s %.4f ms per 10 k-query step = %.1f M q/s" % (name, ms, w["qn"] / ms / 1e3))
same = torch.equal(b[0][0], b[1][0]) and torch.equal(b[0][1], b[1][1])
print("results of the two slots identical:", same)
