export PQT_BENCH_NO_PIPELINE=1
python bench.py --steps 10 --warmup 3 --no-cpu --extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print(round(d['value']), d['ms_per_step'], c['stage_ms'])
for k in ('knobs_4096_4096','knobs_4096_4096_k4096','knobs_4096_4096_k4096_staged','exact_rerank_of_topk'):
    print(k, json.dumps(c.get(k)))"
