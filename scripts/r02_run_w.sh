timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -3
export PQT_BENCH_NO_PIPELINE=1
for rep in 1 2 3; do
python bench.py --no-cpu --no-gt --no-ref1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print(round(d['value']), round(d['ms_per_step'],4), {k:round(v,4) for k,v in c['stage_ms'].items() if v})"
done
bash scripts/r02_tstamp1.sh 2>&1 | grep "phase cycles"
