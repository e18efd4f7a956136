#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
PQT_SHARD_WORKLOAD=synth10m timeout 900 python scripts/r04_pipeline_one_device.py > gpurun_out/r04/r04_pipeline_one_device_synth10m.json 2> gpurun_out/r04/pipe10m.log || tail -20 gpurun_out/r04/pipe10m.log
cat gpurun_out/r04/r04_pipeline_one_device_synth10m.json
PQT_SHARD_WORKLOAD=synth100m timeout 1500 python scripts/r04_pipeline_one_device.py > gpurun_out/r04/r04_pipeline_one_device_synth100m.json 2> gpurun_out/r04/pipe100m.log || tail -20 gpurun_out/r04/pipe100m.log
cat gpurun_out/r04/r04_pipeline_one_device_synth100m.json
