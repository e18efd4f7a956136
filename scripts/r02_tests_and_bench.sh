python -m pytest tests -m gpu -x -q 2>&1 | tail -4
export PQT_BENCH_NO_PIPELINE=1
for wl in sift1m synth10m; do
  for knobs in "--bv 20000 --bb 500" "--bv 4096 --bb 4096"; do
  echo "== $wl $knobs"
  python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu $knobs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print(round(d['value']), round(d['ms_per_step'],4), c['stage_ms'], c['recall@1'], c['mean_candidates'])"
  done
done
