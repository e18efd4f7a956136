#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_bench_sharded.py -m gpu -q -x -k "sharded or shard or multi or bins or eight" > gpurun_out/r03/pytest10.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r03/pytest10.log | cut -c1-300
for dbg in 0; do
  PQT_DBG=$dbg PQT_SHARDS_MEASURED=2 python scripts/r03_shard8_one_device.py 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d['knobs'].items(): print('shard8 10m PQT_DBG=$dbg', k, 'unsharded', v['unsharded']['step_ms'], 'replicated', v['per_shard'][0]['replicated'], 'sharded', {x:v['per_shard'][0]['query_sharded'][x] for x in ('traverse_slice_ms','tables_resolve_ms','rerank_select_ms','per_rank_ms','identical_to_replicated')})"
done
