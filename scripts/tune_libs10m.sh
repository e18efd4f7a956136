# usage: bash scripts/tune_libs10m.sh -- bench synth10m once per tune/lib_*.so
for f in tune/lib_*.so; do
  echo -n "$f: "
  PQT_LIB=$PWD/$f python bench.py --workload synth10m --steps 5 --warmup 2 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print(round(d['value']), c['stage_ms'])
"
done
