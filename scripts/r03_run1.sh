#!/bin/bash
# round 3, GPU run 1: the whole -m gpu suite (incl. the new 100 M module), the default bench line (now with the hbm_roofline_leg), cfg5 build number
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r03/pytest1.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03/pytest1.log
tail -25 gpurun_out/r03/pytest1.log
timeout 900 python bench.py > gpurun_out/r03/bench_default.json 2> gpurun_out/r03/bench_default.log; echo "bench rc $?"
tail -5 gpurun_out/r03/bench_default.log
cut -c1-3000 gpurun_out/r03/bench_default.json
timeout 300 python scripts/r03_cfg5_build.py > gpurun_out/r03/cfg5_build.json 2> gpurun_out/r03/cfg5_build.log; echo "cfg5 rc $?"
cat gpurun_out/r03/cfg5_build.json
