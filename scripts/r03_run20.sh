#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
run() { PQT_BENCH_NO_PIPELINE=1 python bench.py --workload sift1m --steps 40 --warmup 5 --no-cpu --no-hbm-leg --no-gt --timing-period 9 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('grid=$PQT_RS_GRID args=$*', round(d['value']/1e6,3),'M q/s', round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['config']['stage_ms'].items() if v})"; }
for g in 96 128 160 192 224 256 384 512; do PQT_RS_GRID=$g run --option overlap=0; done
run --option overlap=2 --option balance=2
run --option overlap=0 --option balance=2
PQT_RS_GRID=128 run --option overlap=0 --option balance=2
