#!/bin/bash
# one sanitizer pass.  (1) host layer + csrc/pqt_multi.cpp under ASan+UBSan and under TSan (host/Makefile: make SAN=..),
# driven by host/san_driver.cpp (packed hand-over, HostPool, async slots, two-shard object) and host/test_classes.cpp; (2) the device library
# built with -fsanitize=address (gfx950:xnack+, tune/lib_asan.so, built beforehand) on three parity fixtures.  Logs -> gpurun_out/san/.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/san; rm -f gpurun_out/san/summary.txt
W=/tmp/san_work; rm -rf $W; mkdir -p $W
python - <<PY
import sys, os, numpy as np
sys.path.insert(0, "tests")
from common import fixture
f = fixture("tools_default")
f.oracle.save_tree("$W/o.tree"); f.oracle.save_bins("$W/o.bins")
f.queries[:12].astype(np.float32).tofile("$W/q.raw")
open("$W/cfg", "w").write("%d %d %d %d" % (f.cfg["D"], f.cfg["P"], f.cfg["LP"], f.cfg["W"]))
PY
read D P LP WW < $W/cfg
H=product-quantization-tree_amd/host
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/product-quantization-tree_amd/csrc:/opt/rocm/lib:$LD_LIBRARY_PATH
export PQT_FRONTEND_PACK_MIN_BYTES=0
for san in address thread; do
  if [ $san = address ]; then export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 UBSAN_OPTIONS=print_stacktrace=1; else export TSAN_OPTIONS="report_signal_unsafe=0 history_size=4 suppressions=$GRAFT_REPO_ROOT/scripts/tsan.supp"; fi
  RUN=""; if [ $san = thread ]; then RUN="setarch $(uname -m) -R"; fi  # (TSan: fixed address-space layout, else 'unexpected memory mapping' on this kernel)
  timeout 280 $RUN $H/san/san_driver_$san $D $P $LP $WW $W/o.tree $W/o.bins $W/q.raw 12 > gpurun_out/san/san_driver_$san.log 2>&1 < /dev/null; echo "san_driver_$san rc=$?" | tee -a gpurun_out/san/summary.txt
  timeout 280 $RUN $H/san/test_classes_$san $D $P $LP $WW $W/o.tree $W/o.bins $W/q.raw 12 1500 400 $W/res_$san.bin > gpurun_out/san/test_classes_$san.log 2>&1 < /dev/null; echo "test_classes_$san rc=$?" | tee -a gpurun_out/san/summary.txt
  grep -c "ERROR: AddressSanitizer\|runtime error:\|WARNING: ThreadSanitizer" gpurun_out/san/san_driver_$san.log gpurun_out/san/test_classes_$san.log | tee -a gpurun_out/san/summary.txt
  tail -2 gpurun_out/san/san_driver_$san.log | cut -c1-200
done
unset ASAN_OPTIONS TSAN_OPTIONS
if [ -f tune/lib_asan.so ]; then
  RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
  HSA_XNACK=1 ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 LD_PRELOAD=$RT PQT_LIB=$GRAFT_REPO_ROOT/tune/lib_asan.so timeout 500 python -m pytest tests/test_gpu_parity.py -x -q -k "(cfg2_small or ties or wrap) and fused and (test_candidates_and_full_sorted_list or test_topk_select_path or test_edge_bounds)" > gpurun_out/san/device_asan.log 2>&1 < /dev/null
  echo "device asan rc=$?" | tee -a gpurun_out/san/summary.txt
  tail -5 gpurun_out/san/device_asan.log | cut -c1-300
fi
cat gpurun_out/san/summary.txt
