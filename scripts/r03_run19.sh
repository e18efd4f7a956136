#!/bin/bash
# timeline of the two overlapped pieces (kernel trace with start/end stamps) + one-piece rerank on half the workgroup slots
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
run() { PQT_BENCH_NO_PIPELINE=1 python bench.py --workload sift1m --steps 40 --warmup 5 --no-cpu --no-hbm-leg --no-gt --timing-period 44 --option overlap=$1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('overlap=$1 first=$PQT_OVERLAP_FIRST_PCT', round(d['value']/1e6,3),'M q/s', round(d['ms_per_step'],4))"; }
run 0
run 2
PQT_OVERLAP_FIRST_PCT=99 run 2
PQT_OVERLAP_FIRST_PCT=1 run 2
cd /tmp
PQT_BENCH_NO_PIPELINE=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof19 -o t -- python $GRAFT_REPO_ROOT/bench.py --workload sift1m --steps 8 --warmup 2 --no-cpu --no-hbm-leg --no-gt --timing-period 9 --option overlap=2 > /dev/null 2>&1
f=$(find /tmp/prof19 -name '*kernel_trace.csv' | head -1)
python - <<PY
import csv
rows=[r for r in csv.DictReader(open("$f")) if 'pqt_k_traverse' in r['Kernel_Name'] or 'pqt_k_rerank_select' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
rows=rows[-16:]
t0=int(rows[0]['Start_Timestamp'])
for r in rows:
    print(r['Kernel_Name'][:40].ljust(40), 'grid', r.get('Grid_Size_X', r.get('Grid_Size','?')), 'stream', r.get('Stream_Id', r.get('Queue_Id','?')), 'start %8.1f us  end %8.1f us  dur %6.1f' % ((int(r['Start_Timestamp'])-t0)/1e3,(int(r['End_Timestamp'])-t0)/1e3,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3))
PY
