"""The reference front-end's own call shape -- queryKNN(.., 4096) at (4096, 4096) on the SIFT1M-shape index -- a few times, for rocprofv3 --kernel-trace --stats."""
import importlib, os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module("product-quantization-tree_amd")
w = bench.WORKLOADS["sift1m"]
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
st = torch.cuda.Stream(dev); torch.cuda.set_stream(st)
idx, base, meta = bench.build_index(pkg, w, 0)
idx.build_heuristic(4096)
for ov in sys.argv[1:]:
    idx.set_option(ov.split("=")[0], int(ov.split("=")[1]))
qn, k = w["qn"], 4096
q = bench.sift_like(qn, w["D"], 0xC0DE03, dev)
oi = torch.empty((qn, k), dtype=torch.int32, device=dev); od = torch.empty((qn, k), dtype=torch.float32, device=dev); oc = torch.empty(qn, dtype=torch.int32, device=dev)
idx.set_option("stage_timing", 0)
for _ in range(3): idx.query_dev(q, 4096, 4096, k, oi, od, oc, stream=st.cuda_stream)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(st)
for _ in range(10): idx.query_dev(q, 4096, 4096, k, oi, od, oc, stream=st.cuda_stream)
e1.record(st); torch.cuda.synchronize()
c = oc.cpu().numpy()
print(json.dumps({"ms_per_call": e0.elapsed_time(e1) / 10, "path": idx.last_path(), "lists<=1024": int((c <= 1024).sum()), "1025..2048": int(((c > 1024) & (c <= 2048)).sum()), ">2048": int((c > 2048).sum())}))
