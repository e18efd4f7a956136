#!/bin/bash
# full -m gpu suite with the shipped library, then the default bench command (headline + hbm leg) and the A/B of development libraries
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -25
timeout 900 python bench.py 2> gpurun_out/r04/default.log | grep '^{"metric' > gpurun_out/r04/r04_bench_default.json
python - <<'PY'
import json
d = json.load(open('gpurun_out/r04/r04_bench_default.json')); c = d['config']
print(round(d['value']), round(d['ms_per_step'], 4), {k: round(v, 4) for k, v in c['stage_ms'].items() if v}, 'frac', round(d['roofline']['frac'], 3), c['kernel_path'], d.get('cpu_baseline'))
print('no_stage_events', {k: (round(v['queries_per_sec']), v['results_identical']) for k, v in (c.get('no_stage_events') or {}).items() if isinstance(v, dict)})
leg = c.get('hbm_roofline_leg') or {}
for kk in ('knobs_20000_500', 'knobs_4096_4096'):
    e = leg.get(kk)
    if e: print('hbm leg', kk, round(e['queries_per_sec']), round(e['ms_per_step'], 3), e['stage_ms'], 'frac', round(e['roofline']['frac'], 3), e['kernel_path'])
PY
