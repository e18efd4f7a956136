#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for b in 1 2; do
PQT_BALANCE=$b PQT_SHARD_WORKLOAD=synth10m PQT_SHARDS_MEASURED=1 python scripts/r03_shard8_one_device.py 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d['knobs'].items(): print('balance=$b',k,'unsharded',v['unsharded']['step_ms'],'per-rank',v['per_rank_ms_query_sharded'],v['per_shard'][0]['query_sharded']['rerank_select_ms'])"
done
