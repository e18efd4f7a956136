#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
run() { PQT_BENCH_NO_PIPELINE=1 python bench.py --workload ${WL:-sift1m} --steps 40 --warmup 5 --no-cpu --no-hbm-leg --no-gt --timing-period 9 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${WL:-sift1m} nw=$PQT_FUSED_NW grid=$PQT_RS_GRID args=$*', round(d['value']/1e6,3),'M q/s', round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['config']['stage_ms'].items() if v}, d['config']['kernel_path'])"; }
run --option one_launch=0
run --option one_launch=1
PQT_FUSED_NW=8 run --option one_launch=1
PQT_FUSED_NW=8 PQT_RS_GRID=512 run --option one_launch=1
PQT_RS_GRID=512 run --option one_launch=1
