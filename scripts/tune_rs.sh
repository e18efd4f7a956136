for f in tune/lib_rs_*.so; do
  echo -n "$f: "
  PQT_LIB=$PWD/$f python bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print(round(d['value']), c['stage_ms'])
"
done
