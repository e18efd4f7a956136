python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "big_k or topk_select" 2>&1 | tail -3
bash scripts/r02_run_d.sh
