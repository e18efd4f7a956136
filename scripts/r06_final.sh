#!/bin/bash
# final validation of round 5 on one box: GPU suite, the default bench line (with the hbm leg and live counters), the extras line, the device-side
# sanitizer pass when tune/lib_asan.so is there.  Outputs -> gpurun_out/final/
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/final/gpu_suite.log 2>&1 < /dev/null; tail -3 gpurun_out/final/gpu_suite.log
PQT_TEST_EIGHT_RANKS=1 timeout 1600 python -m pytest tests/test_gpu_bench_sharded.py -q -k eight_ranks > gpurun_out/final/eight_ranks.log 2>&1 < /dev/null; tail -3 gpurun_out/final/eight_ranks.log
timeout 900 python bench.py > gpurun_out/final/r06_bench_default.json 2> gpurun_out/final/bench_default.log < /dev/null; echo "bench rc=$?"
timeout 600 python bench.py --extras --no-hbm-leg --no-live-traffic > gpurun_out/final/r06_bench_extras.json 2> gpurun_out/final/bench_extras.log < /dev/null; echo "extras rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/final/r06_bench_default.json").read().strip().splitlines()[-1])
print("value", round(d["value"]), "one at a time", round(d.get("value_one_batch_at_a_time", 0)), "frac", round(d["roofline"]["frac"], 3), "issue", (d["roofline"].get("issue") or {}).get("frac_of_issue_ceiling"))
for k, v in d["config"].get("hbm_roofline_leg", {}).items():
    if isinstance(v, dict) and "queries_per_sec" in v:
        print(" ", k, round(v["queries_per_sec"]), round(v["ms_per_step"], 3), v["kernel_path"], {a: round(b, 3) for a, b in v["stage_ms"].items() if b})
PY
if [ -f tune/lib_asan.so ]; then
  RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
  HSA_XNACK=1 ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 LD_PRELOAD=$RT PQT_LIB=$GRAFT_REPO_ROOT/tune/lib_asan.so timeout 500 python -m pytest tests/test_gpu_parity.py -x -q \
    -k "(cfg2_small or ties or wrap) and fused and (test_candidates_and_full_sorted_list or test_topk_select_path or test_edge_bounds)" > gpurun_out/final/device_asan.log 2>&1 < /dev/null
  echo "device asan rc=$?"; tail -4 gpurun_out/final/device_asan.log | cut -c1-240
fi
