#!/bin/bash
# round 4, GPU call 1: the whole -m gpu suite, then the bench lines that the round's host-side work added
# (front-end leg in --extras, configs[4]-shape throughput leg, bare 2-rank launch on one device)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r04/pytest_gpu.txt
cat gpurun_out/r04/pytest_gpu.txt
timeout 600 python bench.py --extras --no-hbm-leg 2> gpurun_out/r04/extras.log | grep '^{"metric' > gpurun_out/r04/r04_bench_default_extras.json
timeout 900 python bench.py --workload synth_cfg5 --cpu-seconds 8 2> gpurun_out/r04/cfg5.log | grep '^{"metric' > gpurun_out/r04/r04_bench_synth_cfg5_20000_500.json
timeout 900 python bench.py --workload synth_cfg5 --bv 4096 --bb 4096 --no-cpu 2> gpurun_out/r04/cfg5b.log | grep '^{"metric' > gpurun_out/r04/r04_bench_synth_cfg5_4096_4096.json
tail -3 gpurun_out/r04/extras.log gpurun_out/r04/cfg5.log gpurun_out/r04/cfg5b.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r04/r04_*.json')):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, 'unreadable', e); continue
    c = d['config']
    print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'], 4), {k: round(v, 4) for k, v in c['stage_ms'].items() if v}, 'frac', round(d['roofline']['frac'], 3), c['kernel_path'], 'cand', c['mean_candidates'], 'cpu', (d.get('cpu_baseline') or {}))
    fe = c.get('frontend_queryKNN')
    if fe: print(json.dumps(fe, indent=1))
    for kk in ('knobs_4096_4096', 'knobs_4096_4096_k4096'):
        e = c.get(kk)
        if e: print('  ', kk, round(e['queries_per_sec']), {k: round(v, 4) for k, v in e['stage_ms'].items() if v}, e.get('kernel_path'))
PY
