#!/bin/bash
# the three plain bench lines of the round (default command = headline + hbm_roofline_leg, extras, 10 M with cpu_baseline)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/bench
python bench.py 2> gpurun_out/bench/default.log | grep '^{"metric' > gpurun_out/bench/r03_bench_default.json
python bench.py --extras --no-hbm-leg 2> gpurun_out/bench/extras.log | grep '^{"metric' > gpurun_out/bench/r03_bench_default_extras.json
python bench.py --workload synth10m 2> gpurun_out/bench/s10m.log | grep '^{"metric' > gpurun_out/bench/r03_bench_synth10m.json
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/bench/r03_*.json')):
    d = json.load(open(f)); c = d['config']
    print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'], 4), {k: round(v, 4) for k, v in c['stage_ms'].items() if v}, 'frac', round(d['roofline']['frac'], 3), 'cpu', (d.get('cpu_baseline') or {}).get('value'))
    print('   no_stage_events', {k: (round(v['queries_per_sec']), round(v['ms_per_step'], 4)) for k, v in (c.get('no_stage_events') or {}).items() if isinstance(v, dict)})
    leg = c.get('hbm_roofline_leg')
    if leg:
        for kk in ('knobs_20000_500', 'knobs_4096_4096'):
            e = leg[kk]; print('   hbm leg', kk, round(e['queries_per_sec']), round(e['ms_per_step'], 3), 'frac', round(e['roofline']['frac'], 3))
    for kk in ('knobs_4096_4096', 'knobs_4096_4096_k4096', 'knobs_4096_4096_k4096_staged'):
        e = c.get(kk)
        if e: print('  ', kk, round(e['queries_per_sec']), {k: round(v, 4) for k, v in e['stage_ms'].items() if v}, e.get('kernel_path'))
PY
