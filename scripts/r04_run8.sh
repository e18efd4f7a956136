#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_tools.py -m gpu -x -q 2>&1 | tail -15
timeout 600 python bench.py --extras --no-hbm-leg --no-cpu 2> gpurun_out/r04/extras.log | grep '^{"metric' > gpurun_out/r04/r04_bench_default_extras.json
python - <<'PY'
import json
d = json.load(open('gpurun_out/r04/r04_bench_default_extras.json')); c = d['config']
print(round(d['value']), c['stage_ms'])
fe = c.get('frontend_queryKNN')
for k, v in fe.items():
    if isinstance(v, dict):
        for kk, e in v.items(): print(k, kk, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in e.items()})
    else: print(k, v)
for kk in ('knobs_4096_4096', 'knobs_4096_4096_k4096'):
    e = c.get(kk)
    if e: print('  ', kk, round(e['queries_per_sec']), {k: round(v, 4) for k, v in e['stage_ms'].items() if v}, e.get('kernel_path'))
PY
