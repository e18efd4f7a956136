#!/bin/bash
# round 4, run 19: one-flush fast path of the rerank (scan compaction + 32-bit final network) and the early exit of the radix select:
# rerank-side parity tests with the new library, then a same-box A/B of tune/lib_{old,mid,new}.so on the headline workload
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "xcode or topk or select or short or fused or candidates or edge or ties or primitives or self_checks or big_k" 2>&1 | tail -5 > gpurun_out/r04/run19_tests.txt
cat gpurun_out/r04/run19_tests.txt
bash scripts/r04_ab.sh 2>&1 | tee gpurun_out/r04/run19_ab.txt
