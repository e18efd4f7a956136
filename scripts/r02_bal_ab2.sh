export PQT_BENCH_NO_PIPELINE=1
for rep in 1 2 3; do
for bal in 1 2; do
python bench.py --workload sift1m --bv 4096 --bb 4096 --option balance=$bal --no-cpu --no-gt --no-ref1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('sift1m 4096/4096 balance=$bal', round(d['value']), round(d['ms_per_step'],4), {k:round(v,4) for k,v in c['stage_ms'].items() if v})"
done; done
python scripts/r02_timing_cost.py 2>&1 | grep stage_timing | tail -2
PQT_BALANCE=1 python scripts/r02_timing_cost.py 2>&1 | grep stage_timing | tail -2
