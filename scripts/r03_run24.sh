#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "tie_statistics or overlapped" 2>&1 | tail -12 | cut -c1-400
