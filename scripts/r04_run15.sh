#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r04
( time python bench.py 2> gpurun_out/r04/live.log | grep '^{"metric' > gpurun_out/r04/live_default.json ) 2>&1 | grep real
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04/live_default.json')); r=d['roofline']
print(d['value'], r['frac'], r['traffic'], r['traffic_committed_profile'], r['traffic_ratio']); print(r['traffic_source'])
for kn, e in d['config']['hbm_roofline_leg'].items():
    if kn.startswith('knobs'):
        r = e['roofline']; print(kn, e['queries_per_sec'], r['frac'], r['traffic'], r.get('traffic_committed_profile'), r['traffic_ratio'], r['traffic_source'][:60])
print(d['config']['hbm_roofline_leg']['leg_seconds'])
PY
grep -i "live traffic" gpurun_out/r04/live.log | head
