# stage times of the default bench with parts of pqt_k_rerank_select switched off (results are wrong for dbg != 0)
#   1 = no final sort, 2 = no candidates (fixed per-query overhead only), 4 = no coarse copy to LDS, 8 = no ADC arithmetic,
#   16 = cache-resident rows
for d in 0 8 9 24 25 2; do
  echo -n "dbg $d: "
  PQT_DBG=$d python bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print(c['stage_ms'])
"
done
