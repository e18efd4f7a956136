#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -3 | cut -c1-300
run() { PQT_BENCH_NO_PIPELINE=1 python bench.py --workload ${WL:-sift1m} --steps 40 --warmup 5 --no-cpu --no-hbm-leg --no-gt --timing-period 9 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${WL:-sift1m} first=$PQT_OVERLAP_FIRST_PCT args=$*', round(d['value']/1e6,3),'M q/s', round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['config']['stage_ms'].items() if v})"; }
run --option overlap=0
run --option overlap=2
PQT_OVERLAP_FIRST_PCT=50 run --option overlap=2
run --option overlap=0 --option balance=2
WL=synth10m run --option overlap=0
WL=synth100m run --option overlap=0
WL=synth100m run --option overlap=0 --bv 4096 --bb 4096
