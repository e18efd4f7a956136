#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_bench_sharded.py -m gpu -q -x -k "sharded or shard or multi or bins or eight or overlapped" 2>&1 | tail -3 | cut -c1-300
for wl in ${WLS:-synth10m}; do
PQT_SHARD_WORKLOAD=$wl PQT_SHARD_OVERLAP=1 PQT_SHARDS_MEASURED=1 python scripts/r03_shard8_one_device.py 2>/dev/null > gpurun_out/r03/shard8_ov_$wl.json
python - <<PY
import json
d=json.load(open("gpurun_out/r03/shard8_ov_$wl.json"))
for k,v in d['knobs'].items():
    p=v['per_shard'][0]
    print("$wl",k,'unsharded',v['unsharded']['step_ms'],'timed per-rank',p['query_sharded']['per_rank_ms'],'trav slice',p['query_sharded']['traverse_slice_ms'])
    for n,u in p['without_stage_events'].items(): print('   ',n,u)
PY
done
