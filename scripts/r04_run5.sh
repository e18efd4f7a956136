#!/bin/bash
# per-phase clocks of the two headline kernels (instrumented run: PQT_TSTAMP=1) with a development library
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for f in "$@"; do
  echo "== $f"
  PQT_TSTAMP=1 PQT_LIB=$PWD/tune/lib_$f.so timeout 600 python bench.py --steps 6 --warmup 2 --no-hbm-leg --no-cpu --timing-period 1 2>&1 >/dev/null | grep -E "tstamp|stage|path"
done
