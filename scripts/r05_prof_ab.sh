#!/bin/bash
# kernel statistics (rocprofv3 --kernel-trace --stats) of scripts/r05_shared_ab.py: every kernel of both modes, averaged by name
#   bash scripts/r05_prof_ab.sh <tag> [args of r05_shared_ab.py]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
rm -rf /tmp/prof_$tag
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o $tag -- python scripts/r05_shared_ab.py --out gpurun_out/prof/${tag}_ab.json "$@" > gpurun_out/prof/${tag}_ab.log 2>&1 < /dev/null
cp /tmp/prof_$tag/${tag}_kernel_stats.csv gpurun_out/prof/ 2>/dev/null
grep -v "^W2026\|amdgpu.ids" gpurun_out/prof/${tag}_ab.log | tail -12
if [ -f gpurun_out/prof/${tag}_kernel_stats.csv ]; then grep "pqt_k_" gpurun_out/prof/${tag}_kernel_stats.csv | grep -v "assign_encode\|reorder\|group_major\|adc_bias\|coarse" | cut -c1-330 | head -20; fi
