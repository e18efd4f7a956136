import importlib, sys, os
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
pkg = importlib.import_module("product-quantization-tree_amd")
w = dict(bench.WORKLOADS["synth10m"])
if len(sys.argv) > 1: w["n_base"] = int(sys.argv[1]); 
if len(sys.argv) > 2: w["chunk"] = int(sys.argv[2])
idx, _, meta = bench.build_index_chunked(pkg, w, 0)
sizes = meta["sizes"]; print("n", w["n_base"], "chunk", w["chunk"], "bins", len(sizes), "max", sizes.max(), "top5", np.sort(sizes)[-5:])
j = sizes.argmax(); big = meta["bin_ids"][j]
starts = np.concatenate([[0], np.cumsum(sizes.astype(np.int64))])
mem = meta["members"][starts[j]:starts[j + 1]]
print("big bin id", hex(int(big)), "members by chunk", np.bincount(mem // w["chunk"], minlength=w["n_base"] // w["chunk"]))
dev = torch.device("cuda", 0)
ci = int(np.bincount(mem // w["chunk"]).argmax())
x = bench.sift_like(w["chunk"], w["D"], 0xC0DE02 + 7919 * ci, dev)
loc = torch.from_numpy((mem[mem // w["chunk"] == ci] % w["chunk"]).astype(np.int64)).to(dev)[:5]
print("sample rows of the big bin:", x[loc][:, :12].cpu().numpy())
print("row stats: mean", float(x.mean()), "frac0", float((x == 0).float().mean()), "frac255", float((x == 255).float().mean()))
