"""Per-stage device time + host issue time of one rank's sharded step (one batch) on one device -- where the per-rank 0.85 ms goes."""
import importlib, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("PQT_SHARD_WORKLOAD", "synth100m")
sys.argv = [sys.argv[0]]
exec(open(os.path.join(ROOT, "scripts", "r04_pipeline_one_device.py")).read().split("# calibrate torch.cuda._sleep")[0])
class FD:
    def get_rank(self): return 0
    def all_gather_into_tensor(self, out, inp):
        if out.dtype == torch.int64 and out.dim() == 2 and tuple(out.shape) in self.full: out.copy_(self.full[tuple(out.shape)])
        else: out.view(world, -1).copy_(inp.reshape(1, -1).expand(world, -1))
    def all_to_all_single(self, out, inp): out.copy_(inp)
fd = FD(); fd.full = {}
res = {}
for bv, bb in ((20000, 500), (4096, 4096)):
    cap = sharding.bin_cap_for(bb)
    buf = sharding.ShardBuffers(world, qn, k, dev, bin_cap=cap)
    full = torch.zeros_like(buf.bins_all)
    for s in range(world):
        a, b = min(s * buf.qs, qn), min((s + 1) * buf.qs, qn)
        sh.traverse_bins_dev(queries[a:b], bv, bb, cap, full[a:b], stream=st.cuda_stream)
    torch.cuda.synchronize(); fd.full[tuple(full.shape)] = full
    stages = sharding._stages(eng, fd, world, queries, bv, bb, k, buf, "alltoall", False, "sharded", 0, None)
    names = ["traverse_slice", "bins_allgather(copy)", "tables+resolve+rerank", "permute+alltoall(copy)", "merge_slice", "merged_allgather(copy)", "out_copies"]
    for _ in range(3):
        for f in stages: f()
    torch.cuda.synchronize()
    acc = np.zeros(len(stages)); host = 0.0; reps = 10
    for _ in range(reps):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(stages) + 1)]
        t0 = time.perf_counter()
        ev[0].record(st)
        for i, f in enumerate(stages):
            f(); ev[i + 1].record(st)
        host += time.perf_counter() - t0
        torch.cuda.synchronize()
        acc += np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(len(stages))])
    res["%d_%d" % (bv, bb)] = {"stage_ms": dict(zip(names, (acc / reps).round(4).tolist())), "sum_ms": round(float(acc.sum() / reps), 4), "host_issue_ms": round(host / reps * 1e3, 4),
                               "library_stage_ms": None}
    sh.set_option("stage_timing", 1)
    for f in stages: f()
    torch.cuda.synchronize()
    res["%d_%d" % (bv, bb)]["library_stage_ms"] = dict(zip(bench.STAGES, sh.stage_ms_history(1)[-1].round(4).tolist()))
    sh.set_option("stage_timing", 0)
print(json.dumps(res, indent=1))
