export PQT_BENCH_NO_PIPELINE=1
for lib in product-quantization-tree_amd/csrc/libpqt_hip.so; do
for bal in ${BALS:-1 2}; do
echo "== $lib balance=$bal"
PQT_BALANCE=$bal PQT_LIB=$PWD/$lib PQT_TSTAMP=1 python bench.py --workload ${WL:-sift1m} --steps 5 --warmup 2 --no-cpu --no-gt --no-ref1 2>&1 >/dev/null | grep "tstamp. rerank"
python scripts/r02_tstamp_wg.py gpurun_out/tstamps.npy
done; done
