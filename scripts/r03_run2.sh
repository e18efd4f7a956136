#!/bin/bash
# round 3, GPU run 2: -m gpu suite with the query-sharded traversal + pqt_multi, 8-way one-device shard measurement (with phase clocks)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
make -C product-quantization-tree_amd/host > gpurun_out/r03/host_make.log 2>&1 || tail -20 gpurun_out/r03/host_make.log
timeout 2400 python -m pytest tests -m gpu -q --maxfail=12 --durations=12 -x > gpurun_out/r03/pytest2.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03/pytest2.log
tail -40 gpurun_out/r03/pytest2.log
timeout 600 python scripts/r03_shard8_one_device.py > gpurun_out/r03/shard8_10m.json 2> gpurun_out/r03/shard8_10m.log; echo "shard8 rc $?"
python - <<'PY'
import json
d = json.load(open('gpurun_out/r03/shard8_10m.json'))
for k, v in d['knobs'].items():
    print(k, {x: v[x] for x in v if x not in ('per_shard',)}, v['per_shard'][0])
PY
PQT_TSTAMP=1 PQT_SHARDS_MEASURED=1 timeout 600 python scripts/r03_shard8_one_device.py > gpurun_out/r03/shard8_10m_tstamp.json 2> gpurun_out/r03/shard8_10m_tstamp.log; echo "shard8 tstamp rc $?"
grep -A10 shard0_rerank gpurun_out/r03/shard8_10m_tstamp.json
