#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
run() { PQT_BENCH_NO_PIPELINE=1 python bench.py --workload ${WL:-sift1m} --steps 20 --warmup 3 --no-cpu --no-hbm-leg --no-gt "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${WL:-sift1m} args=$*', round(d['value']/1e6,3),'M q/s', round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['config']['stage_ms'].items() if v})"; }
WL=synth10m run --option balance=1
WL=synth10m run --option balance=2
WL=synth10m run --option balance=1 --bv 4096 --bb 4096
WL=synth10m run --option balance=2 --bv 4096 --bb 4096
