"""Several handles of the same index in ONE process: is the 0.154 / 0.161 ms bimodality of the rerank launch a property of the handle
(where its buffers landed) or of the process?  No timestamps."""
import importlib, os, sys
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
handles = []
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    idx, base, meta = bench.build_index(pkg, w, 0); idx.build_heuristic(500); del base
    handles.append(idx)
for rnd in range(3):
    for i, idx in enumerate(handles):
        for _ in range(25): idx.query_dev(queries, 20000, 500, k, oi, od, oc, stream=st.cuda_stream)
        torch.cuda.synchronize()
        h = idx.stage_ms_history(20).mean(0)
        print("round %d handle %d: traverse %.4f rerank+select %.4f" % (rnd, i, h[1], h[3]), flush=True)
