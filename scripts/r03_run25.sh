#!/bin/bash
# same-box A/B of traversal register budgets (waves per SIMD the allocator leaves room for): PQT_TR_WPS = 4 / 5 (default build) / 6
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
run() { PQT_BENCH_NO_PIPELINE=1 python bench.py --workload ${WL:-sift1m} --steps 30 --warmup 5 --no-cpu --no-hbm-leg --no-gt "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${WL:-sift1m} lib=$PQT_LIB args=$*', round(d['value']/1e6,3),'M q/s', round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['config']['stage_ms'].items() if v})"; }
C=$GRAFT_REPO_ROOT/product-quantization-tree_amd/csrc
for rep in 1 2; do
for lib in "" $C/libpqt_hip_wps4.so $C/libpqt_hip_wps6.so; do
  PQT_LIB=$lib run
  PQT_LIB=$lib run --bv 4096 --bb 4096
  PQT_LIB=$lib WL=synth10m run
done; done
