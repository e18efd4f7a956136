#!/bin/bash
# same-box A/B of the development libraries tune/lib_*.so on the headline workload (two rounds), each checked against the CPU checker
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
for rep in 1 2; do
for f in tune/lib_*.so; do
  echo "== $f $@"
  PQT_LIB=$PWD/$f timeout 600 python bench.py --steps 20 --warmup 5 --no-hbm-leg --cpu-seconds 1 "$@" 2>gpurun_out/r04/ab.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print(round(d['value']), 'one-at-a-time', round((c.get('one_batch_at_a_time') or {}).get('queries_per_sec') or 0), (c.get('one_batch_at_a_time') or {}).get('stage_ms'), {k: round(v, 4) for k, v in c['stage_ms'].items() if v}, c['kernel_path'], 'identical', (d.get('cpu_baseline') or {}).get('result_lists_identical_frac'))
" || tail -3 gpurun_out/r04/ab.log
done
done
