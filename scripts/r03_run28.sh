#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "short_lists or tie_statistics or kernel_path" 2>&1 | tail -6 | cut -c1-400
cd /tmp
PQT_BENCH_NO_PIPELINE=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof28 -o t -- python $GRAFT_REPO_ROOT/bench.py --workload sift1m --bv 4096 --bb 4096 --k 4096 --steps 10 --warmup 3 --no-cpu --no-hbm-leg --no-gt > /tmp/b28.json 2>/tmp/b28.log
python - <<PY
import csv, json
for r in csv.DictReader(open('/tmp/prof28/t_kernel_stats.csv')):
    if 'pqt_k' in r['Name'] and int(r['Calls']) >= 10: print(r['Name'][:75], r['Calls'], round(float(r['AverageNs'])/1e3,1), 'us')
d=json.loads(open('/tmp/b28.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['stage_ms'], d['config']['kernel_path'], d['config'].get('filter_fallbacks'))
PY
