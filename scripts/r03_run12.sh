#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
cd /tmp
PQT_SHARDS_MEASURED=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof12 -o p12 -- python $GRAFT_REPO_ROOT/scripts/r03_shard8_one_device.py > /tmp/p12.log 2>&1
tail -3 /tmp/p12.log; f=$(find /tmp/prof12 -name '*kernel_stats.csv' | head -1); echo stats=$f
cp $f $GRAFT_REPO_ROOT/gpurun_out/r03/shard8_10m_kernel_stats12.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
for r in rows[:14]:
    print(r['Name'][:70], r['Calls'], r['AverageNs'], r['Percentage'])
PY
