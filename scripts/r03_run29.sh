#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_bench_sharded.py -m gpu -q -x -k "sharded or shard or multi or bins or eight" 2>&1 | tail -3 | cut -c1-300
PQT_SHARD_WORKLOAD=synth100m PQT_SHARDS_MEASURED=1 python scripts/r03_shard8_one_device.py 2>/dev/null > gpurun_out/r03/shard8_100m_29.json
python - <<PY
import json
d=json.load(open("gpurun_out/r03/shard8_100m_29.json"))
for k,v in d['knobs'].items():
    print(k,'unsharded',v['unsharded']['step_ms'],'sharded',v['per_rank_ms_query_sharded'],v['speedup_query_sharded'], v['per_shard'][0]['query_sharded'], v['bin_lists'])
PY
