#!/bin/bash
# Counter-to-bytes calibration of FETCH_SIZE for the rerank kernels' access shape (random 64-B / 128-B row gathers,
# 16 B per lane per load) on a 4 GiB / 8 GiB table, per MI355X_MICROARCH.md "calibrate on a known byte count".
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
for rb in 64 128; do
  cat > /tmp/calib.py <<PY
import ctypes, importlib, sys
sys.path.insert(0, '.')
pkg = importlib.import_module('product-quantization-tree_amd')
L = pkg.lib()
L.pqt_debug_calibrate_gather.argtypes = [ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint64, ctypes.POINTER(ctypes.c_float)]
ms = ctypes.c_float()
assert L.pqt_debug_calibrate_gather(0, 26, $rb, 1 << 24, ctypes.byref(ms)) == 0, L.pqt_last_error()
print('ms', ms.value)
PY
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/calib_$rb -o c -- python /tmp/calib.py > gpurun_out/prof/calib_$rb.log 2>&1
  python - <<PY
import csv, glob, json
known = (1 << 24) * $rb
v = [float(r['Counter_Value']) for f in glob.glob('/tmp/calib_$rb/*counter_collection.csv') for r in csv.DictReader(open(f)) if 'calib_gather' in r['Kernel_Name'] and r['Counter_Name'] == 'FETCH_SIZE']
ms = [l.strip() for l in open('gpurun_out/prof/calib_$rb.log') if l.startswith('ms')]
print(json.dumps({'row_bytes': $rb, 'gathers': 1 << 24, 'known_bytes': known, 'FETCH_SIZE_KiB': v, 'counter_bytes_over_known': [x * 1024 / known for x in v], 'kernel_ms': ms}))
PY
done
