#!/bin/bash
# same-box A/B of two library builds: tune/libpqt_r03a.so (before) vs csrc/libpqt_hip.so (after), alternating
cd $GRAFT_REPO_ROOT
export PQT_BENCH_NO_PIPELINE=1 TMPDIR=/tmp
OLD=${OLD:-tune/libpqt_r03a.so}; NEW=product-quantization-tree_amd/csrc/libpqt_hip.so
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('$1', round(d['value']), round(d['ms_per_step'],4), {k:round(v,4) for k,v in c['stage_ms'].items() if v})"; }
for rep in 1 2 3; do for lib in $OLD $NEW; do
  PQT_LIB=$PWD/$lib python bench.py --workload sift1m --steps 20 --warmup 5 --no-cpu --no-gt --no-hbm-leg 2>/dev/null | line "sift1m $(basename $lib)"
done; done
for rep in 1 2; do for lib in $OLD $NEW; do
  PQT_LIB=$PWD/$lib python bench.py --workload synth10m --steps 20 --warmup 5 --no-cpu --no-gt 2>/dev/null | line "synth10m $(basename $lib)"
done; done
for lib in $OLD $NEW; do
  PQT_LIB=$PWD/$lib python bench.py --workload synth10m --bv 4096 --bb 4096 --steps 20 --warmup 5 --no-cpu --no-gt 2>/dev/null | line "synth10m 4096/4096 $(basename $lib)"
done
for lib in $OLD $NEW; do
  PQT_LIB=$PWD/$lib PQT_SHARDS_MEASURED=2 python scripts/r03_shard8_one_device.py 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d['knobs'].items(): print('shard8 10m $(basename $lib)', k, 'unsharded', v['unsharded'], 'replicated', v['per_shard'][0]['replicated'], 'sharded', {x:v['per_shard'][0]['query_sharded'][x] for x in ('traverse_slice_ms','tables_resolve_ms','rerank_select_ms','per_rank_ms','identical_to_replicated')})"
done
if [ -n "$AB_100M" ]; then for lib in $OLD $NEW; do
  PQT_LIB=$PWD/$lib python bench.py --workload synth100m --steps 10 --warmup 3 --no-cpu --no-gt 2>/dev/null | line "synth100m $(basename $lib)"
done; fi
