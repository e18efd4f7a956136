python -m pytest tests -m gpu -x -q 2>&1 | tail -6
export PQT_BENCH_NO_PIPELINE=1
for wl in sift1m synth10m; do
  echo "== $wl"
  python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print(round(d['value']), d['ms_per_step'], c['stage_ms'], c['recall@1'], c['mean_candidates'])"
  PQT_TSTAMP=1 python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu 2>&1 >/dev/null | grep tstamp | head -2
done
