python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python bench.py --workload synth10m --steps 3 --warmup 1 --no-cpu --no-gt 2>&1 >/dev/null | grep "index:"
python bench.py --workload sift1m --steps 3 --warmup 1 --no-cpu --no-gt 2>&1 >/dev/null | grep "index:"
