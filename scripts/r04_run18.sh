#!/bin/bash
# round 4, run 18: the four part sorts of the traversal in one pass (pqt_row_sort64_u32): primitive self-test + the whole -m gpu suite with the
# new library, then a same-box A/B of tune/lib_old.so (-DPQT_NO_ROW_SORT) against tune/lib_new.so on the headline workload
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r04/run18_tests.txt
cat gpurun_out/r04/run18_tests.txt
bash scripts/r04_ab.sh 2>&1 | tee gpurun_out/r04/run18_ab.txt
