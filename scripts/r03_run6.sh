#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
for sc in "synth10m 64 1000 16384 20000 500" "synth10m 64 1000 0 20000 500" "sift1m 64 129 16384 4096 4096" "sift1m 64 129 0 4096 4096" "sift1m 2000 129 16384 4096 4096" "sift1m 2000 129 0 4096 4096" "sift1m 10000 4096 0 4096 4096"; do
  echo "=== $sc"
  timeout 90 python scripts/r03_dbg_small.py $sc 2>&1 | grep -v amdgpu.ids | tail -12
  echo "rc $?"
done
PQT_SHARD_WORKLOAD=synth100m PQT_SHARDS_MEASURED=2 timeout 600 python scripts/r03_shard8_one_device.py > gpurun_out/r03/shard8_100m.json 2> gpurun_out/r03/shard8_100m.log; echo "shard8 100m rc $?"
python - <<'PY'
import json
d = json.load(open('gpurun_out/r03/shard8_100m.json'))
for k, v in d['knobs'].items():
    print(k, {x: v[x] for x in v if x not in ('per_shard',)}, v['per_shard'][0])
PY
