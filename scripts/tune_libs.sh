# usage: bash scripts/tune_libs.sh [bench args]   -- runs bench.py once per tune/lib_*.so (PQT_LIB override)
for f in tune/lib_*.so; do
  echo -n "$f: "
  PQT_LIB=$PWD/$f python bench.py --steps 5 --warmup 2 --no-cpu "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print(round(d['value']), c['stage_ms'])
"
done
