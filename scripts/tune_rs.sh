# rerank_select stage time per tune/lib_*.so, with/without the balancing order and with cache-resident rows (dbg 16)
for f in tune/lib_*.so; do
  for env in "PQT_BALANCE=1 PQT_DBG=0" "PQT_BALANCE=0 PQT_DBG=0" "PQT_BALANCE=1 PQT_DBG=16" "PQT_BALANCE=1 PQT_DBG=17"; do
    echo -n "$f $env: "
    env $env PQT_LIB=$PWD/$f python bench.py --steps 5 --warmup 2 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']['stage_ms']
print(round(c['rerank_select'],4), round(c['gap'],4))
"
  done
done
