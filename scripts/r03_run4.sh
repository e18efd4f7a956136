#!/bin/bash
# round 3, GPU run 4: k = 4096 debug (per-step timeouts), -m gpu suite, early-abandon study, shard8 (+ phase clocks)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
make -C product-quantization-tree_amd/host > gpurun_out/r03/host_make.log 2>&1 || tail -20 gpurun_out/r03/host_make.log

QN=10000 timeout 120 python scripts/r03_dbg_k4096.py > gpurun_out/r03/dbg_k4096.log 2>&1; echo "dbg10000 rc $?"; tail -30 gpurun_out/r03/dbg_k4096.log
timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 --durations=8 > gpurun_out/r03/pytest4.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03/pytest4.log
tail -30 gpurun_out/r03/pytest4.log | cut -c1-300
timeout 300 python scripts/r03_abandon_study.py > gpurun_out/r03/abandon_10m.json 2> gpurun_out/r03/abandon_10m.log; echo "study rc $?"; cat gpurun_out/r03/abandon_10m.json
timeout 300 python scripts/r03_shard8_one_device.py > gpurun_out/r03/shard8_10m.json 2> gpurun_out/r03/shard8_10m.log; echo "shard8 rc $?"
python - <<'PY'
import json
d = json.load(open('gpurun_out/r03/shard8_10m.json'))
for k, v in d['knobs'].items():
    print(k, {x: v[x] for x in v if x not in ('per_shard',)}, v['per_shard'][0])
PY
PQT_TSTAMP=1 PQT_SHARDS_MEASURED=1 timeout 300 python scripts/r03_shard8_one_device.py > gpurun_out/r03/shard8_10m_tstamp.json 2> gpurun_out/r03/shard8_10m_tstamp.log; echo "shard8 tstamp rc $?"
grep -A10 shard0_rerank gpurun_out/r03/shard8_10m_tstamp.json
