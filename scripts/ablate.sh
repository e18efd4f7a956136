for d in 0 8 16 1 17; do
  echo -n "dbg $d: "
  PQT_DBG=$d python bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print(c['stage_ms'])
"
done
