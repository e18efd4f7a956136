#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
python bench.py > gpurun_out/r03/bench_default16.json 2> gpurun_out/r03/bench_default16.log; tail -3 gpurun_out/r03/bench_default16.log
python - <<PY
import json
d=json.loads(open("gpurun_out/r03/bench_default16.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["config"]["stage_ms"], d["config"]["kernel_path"])
print(json.dumps(d["config"].get("overlap"), indent=1))
h=d["config"]["hbm_roofline_leg"]
print(json.dumps(h)[:1500])
print(d["roofline"]["frac"], d["cpu_baseline"])
PY
