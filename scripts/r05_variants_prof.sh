#!/bin/bash
# per-variant kernel statistics: bash scripts/r05_variants_prof.sh "<tag> ..." [args of r05_shared_ab.py]; prints the pass's kernels per variant
tags=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/var
for t in $tags; do
  echo "=== $t"
  rm -rf /tmp/prof_$t
  PQT_LIB=$GRAFT_REPO_ROOT/tune/lib_$t.so timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$t -o $t -- python scripts/r05_shared_ab.py --out gpurun_out/var/$t.json "$@" > gpurun_out/var/$t.log 2>&1 < /dev/null
  grep "^\[" gpurun_out/var/$t.log | grep -v identical | cut -c1-120
  grep "pqt_k_sr_adc\|pqt_k_rerank_select<" /tmp/prof_$t/${t}_kernel_stats.csv | cut -c1-200
done
