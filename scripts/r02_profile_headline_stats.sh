# kernel statistics of the default bench command (no PMC passes) + the plain default lines
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof gpurun_out/bench
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_h -o r02_default -- python bench.py --no-cpu > gpurun_out/prof/r02_default_bench_under_rocprof.json 2> gpurun_out/prof/r02_default_bench.log
cp /tmp/prof_h/r02_default_kernel_stats.csv gpurun_out/prof/
grep "pqt_k_\(rerank_select\|traverse\)" gpurun_out/prof/r02_default_kernel_stats.csv | cut -c1-160
python bench.py 2> gpurun_out/bench/default.log | grep '^{"metric' > gpurun_out/bench/r02_bench_default.json
python bench.py --extras 2> gpurun_out/bench/extras.log | grep '^{"metric' > gpurun_out/bench/r02_bench_default_extras.json
for f in gpurun_out/prof/r02_default_bench_under_rocprof.json gpurun_out/bench/r02_bench_default.json gpurun_out/bench/r02_bench_default_extras.json; do grep '^{"metric' $f | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],4), {k:round(v,4) for k,v in c['stage_ms'].items() if v}, 'frac', round(d['roofline']['frac'],3), 'cpu', (d.get('cpu_baseline') or {}).get('value'), c.get('two_handles_two_streams',{}).get('queries_per_sec'))"; done
