#!/bin/bash
# round 4, run 22: what the final ordering costs at all (debug bit 1 = sorting networks of the rerank skipped, results wrong), and the
# per-phase clocks of the development libraries
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
for f in tune/lib_*.so; do
  echo "== $f"
  PQT_LIB=$PWD/$f PQT_TSTAMP=1 PQT_DBG_SWEEP=0,1 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu --no-hbm-leg 2>&1 >/dev/null | grep "tstamp\|dbg-sweep"
  PQT_LIB=$PWD/$f PQT_DBG_SWEEP=0,1,0,1 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu --no-hbm-leg 2>&1 >/dev/null | grep "dbg-sweep"
done 2>&1 | tee gpurun_out/r04/run22.txt
