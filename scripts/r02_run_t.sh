timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
KN="--bv 4096 --bb 4096" LIBS="tune/libpqt_prev.so product-quantization-tree_amd/csrc/libpqt_hip.so" bash scripts/r02_ab3.sh
WL=synth10m KN="--bv 4096 --bb 4096" LIBS="tune/libpqt_prev.so product-quantization-tree_amd/csrc/libpqt_hip.so" bash scripts/r02_ab3.sh
