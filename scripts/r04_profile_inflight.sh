#!/bin/bash
# kernel statistics of the DEFAULT headline command (two whole batches in flight on one device): every launch of the trace shares the device
# with the other batch's launches, so the averages are the stretched durations the line's `roofline` block is computed from
#   bash scripts/r04_profile_inflight.sh <tag>
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PQT_BENCH_NO_PIPELINE=1   # headline launches only (no one-batch-at-a-time leg, no side legs)
mkdir -p gpurun_out/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o $tag -- python bench.py --no-cpu --no-hbm-leg --no-live-traffic --steps 20 --warmup 3 "$@" > gpurun_out/prof/${tag}_bench_under_rocprof.json 2> gpurun_out/prof/${tag}_bench.log
cp /tmp/prof_$tag/${tag}_kernel_stats.csv gpurun_out/prof/ 2>/dev/null
grep pqt_k gpurun_out/prof/${tag}_kernel_stats.csv | cut -c1-220
cut -c1-400 gpurun_out/prof/${tag}_bench_under_rocprof.json
