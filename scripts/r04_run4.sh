#!/bin/bash
# round 4, GPU call 4: issue-rate microbenchmark beyond 4 waves/SIMD; A/B of the development libraries, X-code rows on and off
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate scripts/micro/valu_rate.hip 2>/dev/null && /tmp/valu_rate > gpurun_out/r04/valu_rate.txt 2>&1
grep "workgroups/CU" gpurun_out/r04/valu_rate.txt
for rep in 1 2; do
for f in tune/lib_*.so; do
  for opt in "" "--option xcode=0"; do
  echo "== $f $opt"
  PQT_LIB=$PWD/$f timeout 600 python bench.py --steps 20 --warmup 5 --no-hbm-leg --cpu-seconds 1 $opt 2>gpurun_out/r04/ab.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print(round(d['value']), {k: round(v, 4) for k, v in c['stage_ms'].items() if v}, c['kernel_path'], 'identical', (d.get('cpu_baseline') or {}).get('result_lists_identical_frac'))
" || tail -3 gpurun_out/r04/ab.log
  done
done
done
