#!/bin/bash
cd /tmp; export TMPDIR=/tmp
for dbg in ${PQT_DBGS:-0 131072 262144}; do
PQT_DBG=$dbg PQT_SHARDS_MEASURED=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof13_$dbg -o p -- python $GRAFT_REPO_ROOT/scripts/r03_shard8_one_device.py > /tmp/p13.log 2>&1
grep tables_resolve /tmp/prof13_$dbg/p_kernel_stats.csv | cut -c1-40,100-200 | sed "s/^/dbg=$dbg /"
done
