#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
run() { PQT_BENCH_NO_PIPELINE=1 python bench.py --workload ${WL:-sift1m} --steps 40 --warmup 5 --no-cpu --no-hbm-leg --no-gt --timing-period 9 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${WL:-sift1m} args=$*', round(d['value']/1e6,3),'M q/s', round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['config']['stage_ms'].items() if v}, d['config']['kernel_path'])"; }
run
run --option bin_runs=1
run
run --option bin_runs=1
