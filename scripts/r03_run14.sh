#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "overlapped" > gpurun_out/r03/pytest14.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r03/pytest14.log | cut -c1-300
for o in 0 1 -1; do
  PQT_BENCH_NO_PIPELINE=1 python bench.py --workload sift1m --steps 40 --warmup 5 --no-cpu --no-hbm-leg --option overlap=$o 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('overlap=$o', round(d['value']/1e6,2),'M q/s', d['ms_per_step'], d['config']['stage_ms'], d['config']['kernel_path'])"
done
