# A/B of several library builds on the same box, alternating
export PQT_BENCH_NO_PIPELINE=1
WL=${WL:-sift1m}; KN=${KN:---bv 20000 --bb 500}
for rep in 1 2 3; do
for lib in ${LIBS:-tune/libpqt_base.so product-quantization-tree_amd/csrc/libpqt_hip.so}; do
  PQT_LIB=$PWD/$lib python bench.py --workload $WL --steps 20 --warmup 5 --no-cpu --no-gt --no-ref1 $KN 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('$lib'.split('/')[-1], round(d['value']), round(d['ms_per_step'],4), {k:round(v,4) for k,v in c['stage_ms'].items() if v})"
done
done
