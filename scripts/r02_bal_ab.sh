export PQT_BENCH_NO_PIPELINE=1
for wl in sift1m synth10m; do
for rep in 1 2 3; do
for bal in 1 2; do
python bench.py --workload $wl --option balance=$bal --no-cpu --no-gt --no-ref1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('$wl balance=$bal', round(d['value']), round(d['ms_per_step'],4), {k:round(v,4) for k,v in c['stage_ms'].items() if v})"
done; done; done
