#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
run() { PQT_BENCH_NO_PIPELINE=1 python bench.py --workload sift1m --steps 40 --warmup 5 --no-cpu --no-hbm-leg --no-gt --timing-period 9 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dbg=$PQT_DBG args=$*', round(d['value']/1e6,3),'M q/s', round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['config']['stage_ms'].items() if v})"; }
run --option overlap=0
PQT_DBG=1048576 run --option overlap=0
PQT_DBG=1048576 run --option overlap=2
python - <<PY
import importlib, sys, torch
sys.path.insert(0, '.')
import bench
pkg = importlib.import_module("product-quantization-tree_amd")
w = bench.WORKLOADS["sift1m"]
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
idx, base, meta = bench.build_index(pkg, w, 0)
idx.build_heuristic(4096)
q = bench.sift_like(w["qn"], w["D"], 0xC0DE03, torch.device("cuda", 0))
k = 100
oi = torch.empty((w["qn"], k), dtype=torch.int32, device="cuda"); od = torch.empty((w["qn"], k), dtype=torch.float32, device="cuda"); oc = torch.empty(w["qn"], dtype=torch.int32, device="cuda")
idx.set_option("overlap", 0)
idx.query_dev(q, 20000, 500, k, oi, od, oc, stream=st.cuda_stream, sync=True)
print(idx.stats())
PY
