#!/bin/bash
# round 4, GPU call 3: VALU/LDS issue-cost microbenchmark (compiled on the box), the two tests that failed in call 2 with their output,
# A/B of the development libraries (tune/lib_*.so) on the headline workload
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate scripts/micro/valu_rate.hip 2>/dev/null && /tmp/valu_rate > gpurun_out/r04/valu_rate.txt 2>&1
cat gpurun_out/r04/valu_rate.txt
timeout 900 python -m pytest tests/test_gpu_bench_sharded.py -m gpu -x -q -k "bare_command" 2>&1 | tail -60
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config5" 2>&1 | tail -30
for f in tune/lib_*.so; do
  echo "== $f"
  PQT_LIB=$PWD/$f timeout 600 python bench.py --steps 20 --warmup 5 --no-hbm-leg --cpu-seconds 2 2>gpurun_out/r04/ab_$(basename $f).log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print(round(d['value']), c['stage_ms'], c['kernel_path'], 'identical', (d.get('cpu_baseline') or {}).get('result_lists_identical_frac'), 'no_events', {k:(round(v['queries_per_sec']), v['results_identical']) for k,v in (c.get('no_stage_events') or {}).items() if isinstance(v, dict)})
"
  tail -2 gpurun_out/r04/ab_$(basename $f).log
done
