#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests/test_gpu_tools.py -x -q -m gpu -k "class_" 2>&1 | tail -15
