#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "cross_lane or stage_tables or tie_fixture or candidates_and_full" 2>&1 | tail -4
bash scripts/r04_ab.sh
