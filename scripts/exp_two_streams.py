import importlib, sys, os, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
pkg = importlib.import_module("product-quantization-tree_amd")
w = bench.WORKLOADS["sift1m"]
dev = torch.device("cuda", 0)
ws = torch.cuda.Stream(dev); torch.cuda.set_stream(ws)
idx1, base, meta = bench.build_index(pkg, w, 0)
idx1.build_heuristic(500)
# second handle on the same data
idx2 = pkg.PqtIndex(w["D"], w["P"], w["C1"], w["C2"], w["W"], w["LP"], device=0)
idx2.set_codebooks(meta["cb1"], meta["cb2"]); idx2.build_heuristic(500)
idx2.set_bins(meta["bin_ids"], meta["sizes"], meta["members"]); idx2.set_lines_dev(idx1._keep[0], 0)
q = bench.sift_like(w["qn"], w["D"], 0xC0DE03, dev)
k = 100
def bufs(n): return (torch.empty((n, k), dtype=torch.int32, device=dev), torch.empty((n, k), dtype=torch.float32, device=dev), torch.empty(n, dtype=torch.int32, device=dev))
torch.cuda.synchronize()
for nsplit in (1, 2, 4):
    parts = np.array_split(np.arange(w["qn"]), nsplit)
    qs = [q[p[0]:p[-1] + 1].contiguous() for p in parts]
    bs = [bufs(len(p)) for p in parts]
    hs = [idx1, idx2]
    def step():
        for i, (qq, b) in enumerate(zip(qs, bs)):
            hs[i % 2].query_dev(qq, 20000, 500, k, b[0], b[1], b[2], stream=None)  # NULL -> each handle's own stream
    for _ in range(3): step()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize()
    print("splits", nsplit, "ms/batch", (time.perf_counter() - t) / 20 * 1e3)
