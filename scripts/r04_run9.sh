#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "short_lists or big_k or topk_select or sharded_two_way or fuzz" 2>&1 | tail -8
timeout 600 python bench.py --extras --no-hbm-leg --no-cpu 2> gpurun_out/r04/extras.log | grep '^{"metric' > gpurun_out/r04/r04_bench_default_extras.json
python - <<'PY'
import json
d = json.load(open('gpurun_out/r04/r04_bench_default_extras.json')); c = d['config']
print(round(d['value']), c['stage_ms'])
for kk in ('knobs_4096_4096', 'knobs_4096_4096_k4096', 'knobs_4096_4096_k4096_staged'):
    e = c.get(kk)
    if e: print('  ', kk, round(e['queries_per_sec']), round(e['ms_per_step'], 4), {k: round(v, 4) for k, v in e['stage_ms'].items() if v}, e.get('kernel_path'))
fe = c.get('frontend_queryKNN')
print({k: {kk: round(vv['total_ms'], 3) for kk, vv in v.items()} for k, v in fe.items() if isinstance(v, dict)})
PY
