#!/bin/bash
# round 4, run 31: final library: the whole -m gpu suite, smoke(), then the plain bench lines (default command with live traffic of the hbm leg, extras, 10 M)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04 gpurun_out/bench
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r04/run31_tests.txt
cat gpurun_out/r04/run31_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --live-traffic-hbm 2> gpurun_out/bench/default.log | grep '^{"metric' > gpurun_out/bench/r04_bench_default.json
python bench.py --extras --no-hbm-leg 2> gpurun_out/bench/extras.log | grep '^{"metric' > gpurun_out/bench/r04_bench_default_extras.json
python bench.py --workload synth10m 2> gpurun_out/bench/s10m.log | grep '^{"metric' > gpurun_out/bench/r04_bench_synth10m.json
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/bench/r04_*.json')):
    d = json.load(open(f)); c = d['config']; r = d['roofline']
    print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'], 4), {k: round(v, 4) for k, v in c['stage_ms'].items() if v}, 'frac', round(r['frac'], 3), 'in flight', round((r.get('in_flight') or {}).get('frac') or 0, 3),
          'one at a time', round((c.get('one_batch_at_a_time') or {}).get('queries_per_sec') or 0), 'traffic', r.get('traffic'), 'cpu', (d.get('cpu_baseline') or {}).get('value'))
    leg = c.get('hbm_roofline_leg')
    if leg:
        for kk in ('knobs_20000_500', 'knobs_4096_4096'):
            e = leg[kk]; print('   hbm leg', kk, round(e['queries_per_sec']), round(e['ms_per_step'], 3), {k: round(v, 4) for k, v in e['stage_ms'].items() if v}, 'frac', round(e['roofline']['frac'], 3), 'traffic ratio', e['roofline'].get('traffic_ratio'))
    for kk in ('knobs_4096_4096', 'knobs_4096_4096_k4096', 'frontend_queryKNN'):
        e = c.get(kk)
        if e: print('  ', kk, str(e)[:300])
PY
