#!/bin/bash
# round 3, GPU run 3: -m gpu suite (small-list kernel, coalesced l1virt), early-abandon study, 8-way shard measurement, extras leg (k = 4096)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
make -C product-quantization-tree_amd/host > gpurun_out/r03/host_make.log 2>&1 || tail -20 gpurun_out/r03/host_make.log
timeout 2400 python -m pytest tests -m gpu -q --maxfail=12 --durations=12 > gpurun_out/r03/pytest3.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03/pytest3.log
tail -60 gpurun_out/r03/pytest3.log | cut -c1-400
timeout 600 python scripts/r03_abandon_study.py > gpurun_out/r03/abandon_10m.json 2> gpurun_out/r03/abandon_10m.log; echo "study rc $?"; cat gpurun_out/r03/abandon_10m.json
timeout 600 python scripts/r03_shard8_one_device.py > gpurun_out/r03/shard8_10m.json 2> gpurun_out/r03/shard8_10m.log; echo "shard8 rc $?"
python - <<'PY'
import json
d = json.load(open('gpurun_out/r03/shard8_10m.json'))
for k, v in d['knobs'].items():
    print(k, {x: v[x] for x in v if x not in ('per_shard',)}, v['per_shard'][0])
PY
PQT_BENCH_NO_PIPELINE=1 timeout 600 python bench.py --extras --no-cpu --no-hbm-leg > gpurun_out/r03/bench_extras.json 2> gpurun_out/r03/bench_extras.log; echo "extras rc $?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r03/bench_extras.json').read().strip().splitlines()[-1])
print(d['value'], d['config']['stage_ms'])
for k in ('knobs_4096_4096', 'knobs_4096_4096_k4096', 'knobs_4096_4096_k4096_staged'):
    e = d['config'].get(k); print(k, e and {x: e[x] for x in ('queries_per_sec', 'ms_per_step', 'stage_ms', 'kernel_path', 'mean_candidates') if x in e})
PY
