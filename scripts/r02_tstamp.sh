export PQT_BENCH_NO_PIPELINE=1
for wl in sift1m synth10m; do
  echo "== $wl"
  PQT_TSTAMP=1 python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu 2>&1 >/dev/null | grep -v amdgpu.ids
done
