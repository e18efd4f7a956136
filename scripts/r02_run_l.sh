export PQT_BENCH_NO_PIPELINE=1
for opt in "" "--option bin_runs=0"; do
  echo "== synth100m $opt"
  python bench.py --workload synth100m --steps 10 --warmup 3 --no-cpu $opt 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print(round(d['value']), round(d['ms_per_step'],4), c['stage_ms'], c['recall@1'], c['mean_candidates'], 'frac', round(d['roofline']['frac'],3), c['build_s'])"
done
