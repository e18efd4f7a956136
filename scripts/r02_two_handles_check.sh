for bal in 1 2; do
python bench.py --option balance=$bal --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('balance=$bal', round(d['value']), round(d['ms_per_step'],4), c.get('two_handles_two_streams'))"
done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PQT_BENCH_NO_PIPELINE=1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_h -o r02_default -- python bench.py --no-cpu > gpurun_out/prof/r02_cfg2_sift1m_bench_under_rocprof.json 2> gpurun_out/prof/r02_default_bench.log
cp /tmp/prof_h/r02_default_kernel_stats.csv gpurun_out/prof/r02_cfg2_sift1m_kernel_stats.csv
grep "pqt_k_\(rerank_select\|traverse\)" gpurun_out/prof/r02_cfg2_sift1m_kernel_stats.csv | cut -c1-160
grep '^{"metric' gpurun_out/prof/r02_cfg2_sift1m_bench_under_rocprof.json | cut -c1-250
