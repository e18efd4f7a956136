#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
rm -f gpurun_out/r04/r04_cfg3_100m_*_tcc.csv
timeout 900 python bench.py 2> gpurun_out/r04/default.log | grep '^{"metric' > gpurun_out/r04/r04_bench_default.json
python - <<'PY'
import json
d = json.load(open('gpurun_out/r04/r04_bench_default.json')); c = d['config']
print(round(d['value']), round(d['ms_per_step'], 4), {k: round(v, 4) for k, v in c['stage_ms'].items() if v}, 'frac', round(d['roofline']['frac'], 3))
print('h2d', c.get('h2d_included'))
leg = c.get('hbm_roofline_leg') or {}
for kk in ('knobs_20000_500', 'knobs_4096_4096'):
    e = leg.get(kk)
    if e: print('hbm leg', kk, round(e['queries_per_sec']), round(e['ms_per_step'], 3), 'frac', round(e['roofline']['frac'], 3)); print(json.dumps(e.get('dram_side'), indent=1))
PY
tail -3 gpurun_out/r04/default.log
bash scripts/r04_pmc_tcc.sh 20000 500
