export PQT_BENCH_NO_PIPELINE=1
PQT_TSTAMP=1 python bench.py --workload ${WL:-sift1m} --steps 5 --warmup 2 --no-cpu --no-gt --no-ref1 2>&1 >/dev/null | grep tstamp
python scripts/r02_tstamp_wg.py gpurun_out/tstamps.npy
