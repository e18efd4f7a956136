#!/bin/bash
# round 4, run 29: the whole -m gpu suite + smoke() with the final library
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r04/run29_tests.txt
cat gpurun_out/r04/run29_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
