#!/bin/bash
# per-rank work of the 8-way layout with the final binary: 10 M (all 8 shards) and 100 M (2 shards measured), then the phase clocks at 10 M
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
PQT_SHARD_WORKLOAD=synth10m python scripts/r03_shard8_one_device.py 2>/dev/null > gpurun_out/r03/r03_shard8_one_device_synth10m.json
PQT_SHARD_WORKLOAD=synth100m PQT_SHARDS_MEASURED=2 python scripts/r03_shard8_one_device.py 2>/dev/null > gpurun_out/r03/r03_shard8_one_device_synth100m.json
PQT_SHARD_WORKLOAD=synth10m PQT_SHARDS_MEASURED=1 PQT_TSTAMP=1 python scripts/r03_shard8_one_device.py 2>/dev/null > gpurun_out/r03/r03_shard8_one_device_synth10m_phase_clocks.json
python - <<PY
import json
for wl in ("synth10m","synth100m"):
    d=json.load(open("gpurun_out/r03/r03_shard8_one_device_%s.json"%wl))
    for k,v in d['knobs'].items():
        print(wl,k,'unsharded',v['unsharded'],'replicated',v['per_rank_ms_replicated'],v['speedup_replicated'],'sharded',v['per_rank_ms_query_sharded'],v['speedup_query_sharded'])
        print('   shard0',v['per_shard'][0]['query_sharded'], v['per_shard'][0]['local_candidates_per_query'])
PY
