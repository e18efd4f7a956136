#!/bin/bash
# kernel-level breakdown of the k = 4096 front-end call at (4096, 4096) (extras leg) + candidate count distribution
cd /tmp; export TMPDIR=/tmp
PQT_BENCH_NO_PIPELINE=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof27 -o t -- python $GRAFT_REPO_ROOT/bench.py --workload sift1m --bv 4096 --bb 4096 --k 4096 --steps 10 --warmup 3 --no-cpu --no-hbm-leg --no-gt > /tmp/b27.json 2>/tmp/b27.log
grep "candidates per query" /tmp/b27.log
python - <<PY
import csv, json
for r in csv.DictReader(open('/tmp/prof27/t_kernel_stats.csv')):
    if 'pqt_k' in r['Name'] and int(r['Calls']) >= 10: print(r['Name'][:70], r['Calls'], round(float(r['AverageNs'])/1e3,1), 'us')
d=json.loads(open('/tmp/b27.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['stage_ms'], d['config']['kernel_path'], d['config'].get('filter_fallbacks'))
PY
