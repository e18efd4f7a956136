timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
export PQT_BENCH_NO_PIPELINE=1
for lib in tune/libpqt_base.so product-quantization-tree_amd/csrc/libpqt_hip.so; do
PQT_LIB=$PWD/$lib python bench.py --workload synth100m --steps 10 --warmup 3 --no-cpu --no-gt --no-ref1 --bv 20000 --bb 500 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('$lib'.split('/')[-1], round(d['value']), round(d['ms_per_step'],4), {k:round(v,4) for k,v in c['stage_ms'].items() if v}, d['roofline']['frac'])"
done
