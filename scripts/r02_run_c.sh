python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "adc_bias" 2>&1 | tail -6
export PQT_BENCH_NO_PIPELINE=1
for wl in synth10m sift1m; do
  for opt in "" "--option adc_bias=1"; do
  echo "== $wl $opt"
  python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu $opt 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print(round(d['value']), d['ms_per_step'], c['stage_ms'], c['recall@1'], c['recall@100'], c['mean_candidates'], d['roofline']['frac'])"
  done
done
