#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "overlapped" 2>&1 | tail -3 | cut -c1-300
run() { PQT_BENCH_NO_PIPELINE=1 python bench.py --workload ${WL:-sift1m} --steps 40 --warmup 5 --no-cpu --no-hbm-leg --no-gt --option overlap=$1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${WL:-sift1m} overlap=$1 first=$PQT_OVERLAP_FIRST_PCT', round(d['value']/1e6,3),'M q/s', round(d['ms_per_step'],4), d['config']['kernel_path'])"; }
run 0
run -1
for f in 50 56 62; do PQT_OVERLAP_FIRST_PCT=$f run 2; done
for f in 34 38 44; do PQT_OVERLAP_FIRST_PCT=$f run 3; done
for f in 25 29 35; do PQT_OVERLAP_FIRST_PCT=$f run 4; done
