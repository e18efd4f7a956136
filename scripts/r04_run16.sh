#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "2d_anisotropic or cuda_style or candidates_and_full or view_handle or multi" 2>&1 | tail -15
