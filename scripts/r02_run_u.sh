timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for p in 4 1 4 1; do
python bench.py --timing-period $p --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('period=$p', round(d['value']), round(d['ms_per_step'],4), {k:round(v,4) for k,v in c['stage_ms'].items() if v}, round(d['roofline']['frac'],3), c['kernel_timing'][:60])"
done
python bench.py --workload synth10m --no-cpu 2>/dev/null | cut -c1-300
