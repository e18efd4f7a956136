timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -4
bash scripts/r02_ab.sh
bash scripts/r02_tstamp1.sh
