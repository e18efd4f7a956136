#!/bin/bash
# round 4, run 32: the plain bench lines of the final tree (default command with live traffic of the hbm leg, extras, 10 M)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/bench
python bench.py --live-traffic-hbm 2> gpurun_out/bench/default.log | grep '^{"metric' > gpurun_out/bench/r04_bench_default.json
python bench.py --extras --no-hbm-leg 2> gpurun_out/bench/extras.log | grep '^{"metric' > gpurun_out/bench/r04_bench_default_extras.json
python bench.py --workload synth10m 2> gpurun_out/bench/s10m.log | grep '^{"metric' > gpurun_out/bench/r04_bench_synth10m.json
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/bench/r04_*.json')):
    d = json.load(open(f)); c = d['config']; r = d['roofline']
    print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'], 4), {k: round(v, 4) for k, v in c['stage_ms'].items() if v}, 'frac', round(r['frac'], 3), 'in flight', round((r.get('in_flight') or {}).get('frac') or 0, 3),
          'one at a time', round((c.get('one_batch_at_a_time') or {}).get('queries_per_sec') or 0), 'path frac', round(c['path_frac_of_hbm_peak'], 3), 'cpu', (d.get('cpu_baseline') or {}).get('value'))
    leg = c.get('hbm_roofline_leg')
    if leg:
        for kk in ('knobs_20000_500', 'knobs_4096_4096'):
            e = leg[kk]; print('   hbm leg', kk, round(e['queries_per_sec']), round(e['ms_per_step'], 3), {k: round(v, 4) for k, v in e['stage_ms'].items() if v}, 'frac', round(e['roofline']['frac'], 3), 'traffic ratio', e['roofline'].get('traffic_ratio'))
    e = c.get('knobs_4096_4096_k4096')
    if e: print('   k4096 call', round(e['queries_per_sec']), {k: round(v, 4) for k, v in e['stage_ms'].items() if v})
PY
