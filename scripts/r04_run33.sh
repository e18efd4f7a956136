#!/bin/bash
# round 4, run 33: two batches in flight with the persistent rerank launch on fewer workgroups than CUs (PQT_RS_GRID: the freed CUs take the
# other batch's traversal throughout instead of only where a rerank workgroup has ended)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
for g in 0 248 240 224 192; do
  echo "== PQT_RS_GRID=$g"
  PQT_RS_GRID=$g timeout 300 python bench.py --no-hbm-leg --no-cpu --no-live-traffic --steps 40 --warmup 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print(round(d['value']), round(d['ms_per_step'],4), 'one at a time', round(c['one_batch_at_a_time']['queries_per_sec']), {k: round(v,4) for k,v in c['one_batch_at_a_time']['stage_ms'].items() if v})
"
done 2>&1 | tee gpurun_out/r04/run33.txt
