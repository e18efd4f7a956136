"""BASELINE configs[4] shape (d=256 p=8 c1=128 c2=64 lineparts=32): the only work the reference does at this shape is insert()
(its query enumerates (W*C2)^P = 2^48 -> 0 rows in uint arithmetic), so the one honest number is the throughput of
pqt_build_assign_encode, with the split a1-table (+ bin id) vs pair search from an ablated launch (debug_bits 8192: no pair
search, codes written as 0).  Prints one JSON object; run under rocprofv3 --kernel-trace --stats for the kernel's own clock.
    python scripts/r03_cfg5_build.py [n_vectors]"""
import importlib, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

pkg = importlib.import_module("product-quantization-tree_amd")
D, P, C1, C2, W, LP = 256, 8, 128, 64, 1, 32
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
dev = torch.device("cuda", 0)
s = torch.cuda.Stream(dev)
torch.cuda.set_stream(s)
g = torch.Generator(device=dev); g.manual_seed(5)
# any codebook is a valid tree for a throughput number: centroids drawn from the data distribution
x = bench.sift_like(n, 128, 0xC0DE02, dev)
x = torch.cat([x, bench.sift_like(n, 128, 0xC0DE12, dev)], 1).contiguous()  # d = 256
pick = torch.randperm(n, generator=g, device=dev)
cb1 = x[pick[:C1]].cpu().numpy()
S = D // P
cb2 = np.empty((P, C1, C2, S), np.float32)
xs = x[pick[C1:C1 + C1 * C2]].cpu().numpy().reshape(C1, C2, D)
for p in range(P):
    cb2[p] = xs[:, :, p * S:(p + 1) * S]
idx = pkg.PqtIndex(D, P, C1, C2, W, LP, device=0)
idx.set_codebooks(cb1, cb2)
bins = torch.empty(n, dtype=torch.int32, device=dev)
codes = torch.empty((n, LP), dtype=torch.int32, device=dev)
res = {"shape": "d=256 p=8 c1=128 c2=64 w=1 lineparts=32", "vectors": n, "pairs_per_line_part": C1 * (C1 - 1) // 2}
for name, bits in (("full", 0), ("no_pair_search", 8192)):
    idx.set_option("debug_bits", bits)
    idx.assign_encode_dev(x[:100000], bins[:100000], codes[:100000], stream=s.cuda_stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    idx.assign_encode_dev(x, bins, codes, stream=s.cuda_stream)
    e1.record(s)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    res[name] = {"ms": ms, "vectors_per_sec": n / ms * 1e3}
idx.set_option("debug_bits", 0)
full, nop = res["full"]["ms"], res["no_pair_search"]["ms"]
res["split"] = {"a1_table_and_bin_id_frac": nop / full, "pair_search_frac": 1 - nop / full}
# the pair search evaluates C1(C1-1)/2 pairs x LP line parts per vector, ~12 flops each (calcRatio + extractDistance)
res["pair_search_GFLOPs"] = n * LP * (C1 * (C1 - 1) // 2) * 12 / ((full - nop) * 1e-3) / 1e9
# what an fp32 MFMA GEMM could take over: the a1 table = C1 x D MACs per vector (||q - c||^2 over LP sub-segments)
res["a1_table_GFLOPs_if_gemm"] = n * C1 * D * 2 / (nop * 1e-3) / 1e9
print(json.dumps(res, indent=1))
