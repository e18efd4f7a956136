#!/bin/bash
# A/B of development libraries of the shared-row pass on one box: bash scripts/r05_variants.sh "<tag> <tag> ..." [args of r05_shared_ab.py]
tags=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/var
for t in $tags; do
  echo "=== $t"
  PQT_LIB=$GRAFT_REPO_ROOT/tune/lib_$t.so timeout 300 python scripts/r05_shared_ab.py --out gpurun_out/var/$t.json "$@" 2>&1 < /dev/null | grep "^\[" 
done
