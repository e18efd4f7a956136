export PQT_BENCH_NO_PIPELINE=1
for wl in synth10m synth100m; do
  for opt in "" "--option exact_filter=0" "--option adc_bias=1"; do
  echo "== $wl $opt"
  python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu $opt 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print(round(d['value']), round(d['ms_per_step'],3), c['stage_ms'], c['recall@1'], c['recall@100'], c['mean_candidates'], 'fb', c['filter_fallbacks'], 'frac', round(d['roofline']['frac'],3), c['build_s'])"
  done
done
