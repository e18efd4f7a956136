#!/bin/bash
# round 4, run 21: per-phase clocks of the rerank (PQT_TSTAMP=1) for the development libraries tune/lib_*.so, then the plain A/B
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
for f in tune/lib_*.so; do
  echo "== $f"
  PQT_LIB=$PWD/$f PQT_TSTAMP=1 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu --no-hbm-leg 2>&1 >/dev/null | grep tstamp
done 2>&1 | tee gpurun_out/r04/run21_tstamp.txt
bash scripts/r04_ab.sh 2>&1 | tee gpurun_out/r04/run21_ab.txt
