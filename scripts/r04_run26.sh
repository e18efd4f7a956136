#!/bin/bash
# round 4, run 26: 32-bit arg-min of the W nearest cells + 32-bit ordering of the populated rows in the traversal: traversal-side parity
# tests with the new library, then the same-box A/B of tune/lib_{base,wb,new}.so
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "stage or candidates or edge or ties or wrap or tables or primitives or fullsize or sharded or heuristic" 2>&1 | tail -5 > gpurun_out/r04/run26_tests.txt
cat gpurun_out/r04/run26_tests.txt
bash scripts/r04_ab.sh 2>&1 | tee gpurun_out/r04/run26_ab.txt
