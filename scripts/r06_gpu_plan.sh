#!/bin/bash
# Round 6 measurement plan, in priority order, for ONE gpurun call each (the pool was closed to this repository when the round began:
# everything below is queued for the moment it opens).  Every step writes under gpurun_out/r06/ and has its own timeout.
#   bash scripts/r06_gpu_plan.sh suite | bench | srab | shard100m | shard1b | wide | prof
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r06; mkdir -p $O
case "$1" in
suite)   # HEAD on a GPU first: the whole -m gpu suite + the guarded 8-rank bare command
  timeout 1700 python -m pytest tests -m gpu -q --deselect tests/test_gpu_zz_round6_optin.py > $O/gpu_suite.log 2>&1 < /dev/null; tail -5 $O/gpu_suite.log
  timeout 1200 python -m pytest tests/test_gpu_zz_round6_optin.py -q > $O/gpu_optin.log 2>&1 < /dev/null; tail -5 $O/gpu_optin.log   # the round's opt-in kernels, apart (first run on a device)
  PQT_TEST_EIGHT_RANKS=1 timeout 1200 python -m pytest tests/test_gpu_bench_sharded.py -q -k eight_ranks > $O/eight_ranks.log 2>&1 < /dev/null; tail -3 $O/eight_ranks.log ;;
bench)   # the driver's exact command, then the extras line
  timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_bench_default.json 2> $O/bench_default.log < /dev/null; echo "bench rc=$?"; tail -c 600 $O/r06_bench_default.json
  timeout 700 python bench.py --extras --no-hbm-leg --no-live-traffic > $O/r06_bench_extras.json 2> $O/bench_extras.log < /dev/null; echo "extras rc=$?" ;;
srab)    # evaluating kernel 1 vs 2, scan variants, at 100 M, byte comparison
  timeout 1500 python scripts/r06_sr_ab.py --workload synth100m --steps 10 --out $O/r06_sr_ab.json > $O/sr_ab.log 2>&1 < /dev/null; grep -v "^built" $O/sr_ab.log | tail -40 ;;
shard100m)  # one rank of eight at 100 M: wave-per-query filter kernel vs the cooperative scan vs the pass with kernel 2
  for opt in "" "coop_rerank=1" "shared_rows=1" "shared_rows=1,sr_kernel=2"; do
    tag=$(echo "${opt:-default}" | tr ',=' '__')
    PQT_SHARD_WORKLOAD=synth100m PQT_SKIP_UNSHARDED=1 PQT_SHARD_OPTIONS="$opt" timeout 900 python scripts/r05_pipeline_one_device.py > $O/pipe100m_$tag.json 2> $O/pipe100m_$tag.log < /dev/null
    echo "== $tag rc=$?"; python - <<PY
import json
d = json.load(open("$O/pipe100m_$tag.json"))
for kn, r in d["knobs"].items():
    print(kn, {k: v.get("two_whole_batches_in_flight_ms") for k, v in r.items() if isinstance(v, dict)}, {k: v.get("one_batch_ms") for k, v in r.items() if isinstance(v, dict) and k == "delay_0us"})
PY
  done
  for opt in "" "coop_rerank=1"; do
    tag=$(echo "${opt:-default}" | tr ',=' '__')
    PQT_SHARD_WORKLOAD=synth10m PQT_SKIP_UNSHARDED=1 PQT_SHARD_OPTIONS="$opt" timeout 600 python scripts/r05_pipeline_one_device.py > $O/pipe10m_$tag.json 2> $O/pipe10m_$tag.log < /dev/null; echo "== 10m $tag rc=$?"
  done ;;
shard1b)    # one rank of eight at 1 B (125 M vectors): reuse statistic of the pass on the shard, step with the pass off / on
  for sr in 0 1; do
    PQT_SHARD_WORKLOAD=synth1b PQT_SHARED_ROWS=$sr PQT_SHARD_OPTIONS="sr_stats=1" timeout 1500 python scripts/r05_pipeline_one_device.py > $O/pipe1b_sr$sr.json 2> $O/pipe1b_sr$sr.log < /dev/null; echo "== 1b sr=$sr rc=$?"; tail -c 1500 $O/pipe1b_sr$sr.json
  done ;;
wide)
  timeout 600 python scripts/r05_wide_ab.py sift1m > $O/wide_ab_sift1m.txt 2>&1 < /dev/null; tail -25 $O/wide_ab_sift1m.txt
  timeout 900 python scripts/r05_wide_ab.py synth100m > $O/wide_ab_synth100m.txt 2>&1 < /dev/null; tail -25 $O/wide_ab_synth100m.txt ;;
prof)    # kernel statistics + counters of the default line and of the 100 M leg (rocprofv3; --pmc passes separate from the trace)
  bash scripts/r05_profile.sh > $O/profile.log 2>&1; tail -5 $O/profile.log
  bash scripts/r05_profile_100m.sh > $O/profile_100m.log 2>&1; tail -5 $O/profile_100m.log ;;
variants)  # development libraries of the pass (tune/lib_<tag>.so, scripts/r05_devlib_sr.sh: tile of 1024 / 4096 rows, 8-wavefront scan workgroups) against the
         # build in csrc/, evaluating kernel 1 and 2 each, (20000, 500) only
  for t in base t1024 t4096 sel8; do
    lib=$PWD/tune/lib_$t.so; [ $t = base ] && lib=$PWD/product-quantization-tree_amd/csrc/libpqt_hip.so
    echo "=== $t"; PQT_LIB=$lib timeout 500 python scripts/r06_sr_ab.py --workload synth100m --steps 10 --knobs "20000,500" --variants "sr_kernel=1;sr_kernel=2;sr_kernel=2,sr_scan_split=4,sr_scan_depth=8" --out $O/var_$t.json 2>&1 < /dev/null | grep "^\[\|identical"
  done ;;
asan)    # device-side AddressSanitizer (tune/lib_asan.so rebuilt from the round's sources by /tmp/build_asan.sh's recipe: hipcc -O1 -g
         # --offload-arch=gfx950:xnack+ -fsanitize=address -shared-libsan): default-path fixtures + the round's opt-in kernels
  RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
  export HSA_XNACK=1 ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 LD_PRELOAD=$RT PQT_LIB=$PWD/tune/lib_asan.so
  timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "(cfg2_small or ties or wrap) and fused and (test_candidates_and_full_sorted_list or test_topk_select_path or test_edge_bounds)" > $O/device_asan.log 2>&1 < /dev/null; echo "device asan (default path) rc=$?"; tail -3 $O/device_asan.log | cut -c1-240
  timeout 900 python -m pytest tests/test_gpu_zz_round6_optin.py -x -q -k "agree_bit_for_bit or hands_back or cooperative or compacted" > $O/device_asan_optin.log 2>&1 < /dev/null; echo "device asan (opt-in kernels) rc=$?"; tail -3 $O/device_asan_optin.log | cut -c1-240 ;;
*) echo "usage: $0 suite|bench|srab|variants|shard100m|shard1b|wide|prof|asan" ;;
esac
