"""Round 5: wide enumeration (512 < bound_bins <= 4096) with and without the LDS first level of the presence bitmap (option filter_l1),
same box, same index, byte comparison.  usage: python scripts/r05_wide_ab.py [sift1m|synth100m|synth10m]"""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module("product-quantization-tree_amd")
wl = sys.argv[1] if len(sys.argv) > 1 else "sift1m"
w = bench.WORKLOADS[wl]
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
st = torch.cuda.Stream(dev); torch.cuda.set_stream(st)
idx, base, meta = bench.build_index(pkg, w, 0)
del base
idx.build_heuristic(4096)
qn, k = w["qn"], 100
q = bench.sift_like(qn, w["D"], 0xC0DE03, dev)
oi = torch.empty((qn, k), dtype=torch.int32, device=dev); od = torch.empty((qn, k), dtype=torch.float32, device=dev); oc = torch.empty(qn, dtype=torch.int32, device=dev)
res = {}
for bv, bb in ((4096, 4096), (20000, 2048), (4096, 1024)):
    outs = {}
    for f1 in (0, 1, 2, 0, 1, 2):  # (2, round 6: rows that pass the first level compacted, bitmap asked for full wavefronts of them)
        idx.set_option("filter_l1", f1)
        idx.set_option("stage_timing", 1)
        for _ in range(3): idx.query_dev(q, bv, bb, k, oi, od, oc, stream=st.cuda_stream, sync=True)
        t = time.perf_counter()
        for _ in range(10): idx.query_dev(q, bv, bb, k, oi, od, oc, stream=st.cuda_stream)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t) / 10 * 1e3
        h = idx.stage_ms_history(10).mean(0)
        print("%s (%d,%d) filter_l1=%d: wall %.4f ms  traverse %.4f  rerank %.4f  %.1f M q/s  %s" % (wl, bv, bb, f1, wall, h[1], h[3] + h[4], qn / wall / 1e3, idx.last_path()), flush=True)
        outs[f1] = (oi.cpu().numpy().copy(), od.cpu().numpy().view(np.uint32).copy(), oc.cpu().numpy().copy())
    print("   identical:", all(np.array_equal(outs[0][j], outs[f_][j]) for j in range(3) for f_ in (1, 2)), flush=True)
idx.close()
