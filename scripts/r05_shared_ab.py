"""Round 5: shared-row pass (pqt_shared_rows.h) A/B on a chunk-built workload, same box, same index, byte comparison of the results.
For each knob set: option shared_rows = 0 / 1, the same batch every step and a FRESH batch every step; stage times from the library's
own events (traverse, rerank = everything from the first kernel of the pass to the end of the selection) and wall time per step.
usage: python scripts/r05_shared_ab.py [--workload synth100m] [--steps 10] [--out gpurun_out/r05_shared_ab.json]"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="synth100m")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--out", default="gpurun_out/r05_shared_ab.json")
    ap.add_argument("--knobs", default="20000,500;4096,4096")
    ap.add_argument("--modes", default="0,1")
    a = ap.parse_args()
    pkg = importlib.import_module("product-quantization-tree_amd")
    w = bench.WORKLOADS[a.workload]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    st = torch.cuda.Stream(dev)
    torch.cuda.set_stream(st)
    t0 = time.time()
    idx, base, meta = bench.build_index(pkg, w, 0)
    del base
    print("built %s in %.1f s: bins %d max_bin %d" % (a.workload, time.time() - t0, meta["n_bins"], meta["max_bin"]), flush=True)
    qn, k = w["qn"], 100
    stream = st.cuda_stream
    batches = [bench.sift_like(qn, w["D"], 0xC0DE03 + 17 * i, dev) for i in range(4)]
    res = {"workload": a.workload}
    for kn in a.knobs.split(";"):
        bv, bb = (int(x) for x in kn.split(","))
        idx.build_heuristic(bb)
        outs = {}
        row = {}
        modes = [int(x) for x in a.modes.split(",")]
        for mode in modes:
            idx.set_option("shared_rows", mode)
            idx.set_option("stage_timing", 1)
            oi = torch.empty((qn, k), dtype=torch.int32, device=dev)
            od = torch.empty((qn, k), dtype=torch.float32, device=dev)
            oc = torch.empty(qn, dtype=torch.int32, device=dev)
            for fresh in (False, True):
                for i in range(3):
                    idx.query_dev(batches[i % 4 if fresh else 0], bv, bb, k, oi, od, oc, stream=stream, sync=True)
                torch.cuda.synchronize()
                t1 = time.time()
                for i in range(a.steps):
                    idx.query_dev(batches[i % 4 if fresh else 0], bv, bb, k, oi, od, oc, stream=stream, sync=False)
                torch.cuda.synchronize()
                wall = (time.time() - t1) * 1e3 / a.steps
                h = idx.stage_ms_history(a.steps).mean(0)
                row["shared_rows=%d%s" % (mode, " fresh" if fresh else "")] = {"ms_per_step_wall": wall, "traverse_ms": float(h[1]), "gap_ms": float(h[2]), "rerank_ms": float(h[3]),
                                                                             "path": idx.last_path(), "fallbacks": int(idx.stats()["filter_fallbacks"])}
                print("[%d,%d] shared_rows=%d %s: wall %.3f ms/step  traverse %.3f  gap %.3f  rerank %.3f  (%s) fallbacks %d" %
                      (bv, bb, mode, "fresh" if fresh else "same ", wall, h[1], h[2], h[3], idx.last_path(), idx.stats()["filter_fallbacks"]), flush=True)
            idx.query_dev(batches[0], bv, bb, k, oi, od, oc, stream=stream, sync=True)
            if os.environ.get("PQT_TSTAMP"):
                import ctypes
                ts = np.zeros((qn, 24), np.uint64)
                L = pkg.lib()
                L.pqt_debug_tstamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
                if L.pqt_debug_tstamps(idx.h, ts.ctypes.data, qn) == 0:
                    r = ts[:, 9:14].astype(np.int64)
                    tot = r[:, 4] & 0xffffffff
                    x = ts[:, 16:20].astype(np.int64)
                    print("[tstamp] mode %d per query (shader clocks, medians | means): total %d | %d  rows wait %d | %d  adc+filter %d | %d  flush %d | %d  set-up %d | %d  band %d | %d  out %d | %d  candidates %d; sum/3072 slots %d" %
                          (mode, np.median(tot), tot.mean(), np.median(r[:, 1]), r[:, 1].mean(), np.median(r[:, 2]), r[:, 2].mean(), np.median(r[:, 3]), r[:, 3].mean(),
                           np.median(x[:, 0]), x[:, 0].mean(), np.median(x[:, 1]), x[:, 1].mean(), np.median(x[:, 2]), x[:, 2].mean(), np.median(x[:, 3]), tot.sum() // 3072), flush=True)
            outs[mode] = (oi.cpu().numpy().copy(), od.cpu().numpy().view(np.uint32).copy(), oc.cpu().numpy().copy())
        same = all(np.array_equal(outs[modes[0]][j], outs[modes[-1]][j]) for j in range(3))
        row["identical"] = bool(same)
        print("[%d,%d] results identical: %s" % (bv, bb, same), flush=True)
        res["knobs_%d_%d" % (bv, bb)] = row
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)
    idx.close()


if __name__ == "__main__":
    main()
