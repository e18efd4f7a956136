#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k4096 -o k4096 -- python scripts/r04_k4096.py "$@" 2>/dev/null | tail -1
cp /tmp/prof_k4096/k4096_kernel_stats.csv gpurun_out/r04/r04_k4096_kernel_stats.csv
grep -E "pqt_k" gpurun_out/r04/r04_k4096_kernel_stats.csv | cut -c1-200
