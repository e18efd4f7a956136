timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -3
LIBS="tune/libpqt_prev.so product-quantization-tree_amd/csrc/libpqt_hip.so" bash scripts/r02_ab3.sh
