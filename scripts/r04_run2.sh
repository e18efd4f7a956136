#!/bin/bash
# round 4, GPU call 2: micro-benchmarks (VALU / LDS issue cost per instruction; host fill bandwidth), the -m gpu suite, the bare 2-rank bench
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
scripts/micro/valu_rate > gpurun_out/r04/valu_rate.txt 2>&1
scripts/micro/hostfill > gpurun_out/r04/hostfill.txt 2>&1
cat gpurun_out/r04/hostfill.txt
lscpu | grep -E "Model name|^CPU\(s\)|Thread|Socket" 
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/r04/pytest_gpu.txt
cat gpurun_out/r04/pytest_gpu.txt
cat gpurun_out/r04/valu_rate.txt
