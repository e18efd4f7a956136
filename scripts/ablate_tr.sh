# traverse stage time with parts of pqt_k_traverse switched off (PQT_DBG bits 32 = order all rows, 64 = no bin-table probes,
# 128 = no cb2 reads, 256 = no cb1 reads; results are wrong for the last three)
for d in 0 32 64 128 256 448; do
  echo -n "dbg $d: "
  PQT_DBG=$d python bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print(c['stage_ms'], c['mean_candidates'])
"
done
