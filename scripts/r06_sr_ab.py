"""Round 6: A/B of the shared-row pass's opt-in kernels on a chunk-built workload -- same box, same index, same batches, byte comparison
of every variant's results with the first one's.  A variant is a comma-separated list of option settings (pqt_index_set_option), e.g.
"shared_rows=1,sr_kernel=2,sr_scan_split=4,sr_scan_depth=8".  Stage times from the library's own events (rerank_select = preparation +
the evaluating kernel, select = scan (+ merge) + band) and wall time per step, the same batch every step and a fresh batch every step.
usage: python scripts/r06_sr_ab.py [--workload synth100m] [--steps 10] [--out gpurun_out/r06_sr_ab.json]"""
import argparse
import importlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

DEFAULT_VARIANTS = ["sr_kernel=1", "sr_kernel=2", "sr_kernel=1,sr_scan_depth=8", "sr_kernel=1,sr_scan_split=2,sr_scan_depth=8", "sr_kernel=1,sr_scan_split=4,sr_scan_depth=8",
                    "sr_kernel=1,sr_scan_split=4", "sr_kernel=2,sr_scan_split=4,sr_scan_depth=8"]
RESET = {"sr_kernel": 1, "sr_scan_split": 1, "sr_scan_depth": 4}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="synth100m")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--out", default="gpurun_out/r06_sr_ab.json")
    ap.add_argument("--knobs", default="20000,500;4096,4096")
    ap.add_argument("--variants", default=";".join(DEFAULT_VARIANTS))
    ap.add_argument("--force-shared", type=int, default=1)
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
    if a.force_shared:
        idx.set_option("shared_rows", 1)
    qn, k = w["qn"], 100
    stream = st.cuda_stream
    batches = [bench.sift_like(qn, w["D"], 0xC0DE03 + 17 * i, dev) for i in range(4)]
    res = {"workload": a.workload, "variants": a.variants.split(";")}
    for kn in a.knobs.split(";"):
        bv, bb = (int(x) for x in kn.split(","))
        idx.build_heuristic(bb)
        ref = None
        rows = {}
        for var in a.variants.split(";"):
            for n_, v_ in RESET.items():
                idx.set_option(n_, v_)
            for ov in var.split(","):
                idx.set_option(ov.split("=")[0], int(ov.split("=")[1]))
            idx.set_option("stage_timing", 1)
            oi = torch.empty((qn, k), dtype=torch.int32, device=dev)
            od = torch.empty((qn, k), dtype=torch.float32, device=dev)
            oc = torch.empty(qn, dtype=torch.int32, device=dev)
            row = {}
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
                row["fresh" if fresh else "same"] = {"ms_per_step_wall": wall, "traverse_ms": float(h[1]), "gap_ms": float(h[2]), "rerank_select_ms": float(h[3]), "select_ms": float(h[4]),
                                                     "path": idx.last_path(), "fallbacks": int(idx.stats()["filter_fallbacks"])}
                print("[%d,%d] %-48s %s: wall %.3f ms/step  traverse %.3f  adc %.3f  select %.3f  fallbacks %d" %
                      (bv, bb, var, "fresh" if fresh else "same ", wall, h[1], h[3], h[4], idx.stats()["filter_fallbacks"]), flush=True)
            idx.query_dev(batches[0], bv, bb, k, oi, od, oc, stream=stream, sync=True)
            cur = (oi.clone(), od.clone(), oc.clone())
            if ref is None:
                ref = cur
            row["identical_to_first_variant"] = bool(torch.equal(ref[0], cur[0]) and torch.equal(ref[1], cur[1]) and torch.equal(ref[2], cur[2]))
            print("      results identical to the first variant: %s" % row["identical_to_first_variant"], flush=True)
            rows[var] = row
        res["knobs_%d_%d" % (bv, bb)] = rows
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)
    idx.close()


if __name__ == "__main__":
    main()
