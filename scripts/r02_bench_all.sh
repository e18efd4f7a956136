# the round's plain bench lines (no profiler): default command, extras, 100 M at both knob sets, 10 M
mkdir -p gpurun_out/bench
python bench.py 2> gpurun_out/bench/default.log | grep '^{"metric' > gpurun_out/bench/r02_bench_default.json
python bench.py --extras 2> gpurun_out/bench/extras.log | grep '^{"metric' > gpurun_out/bench/r02_bench_default_extras.json
python bench.py --workload synth100m 2> gpurun_out/bench/s100m.log | grep '^{"metric' > gpurun_out/bench/r02_bench_synth100m.json
python bench.py --workload synth100m --bv 4096 --bb 4096 --no-cpu 2> gpurun_out/bench/s100m_4096.log | grep '^{"metric' > gpurun_out/bench/r02_bench_synth100m_4096_4096.json
python bench.py --workload synth10m 2> gpurun_out/bench/s10m.log | grep '^{"metric' > gpurun_out/bench/r02_bench_synth10m.json
for f in gpurun_out/bench/*.json; do python -c "
import json,sys
d=json.load(open('$f')); c=d['config']
print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],4), {k:round(v,4) for k,v in c['stage_ms'].items() if v}, 'frac', round(d['roofline']['frac'],3), 'ratio', d['roofline'].get('traffic_ratio'), 'cpu', (d.get('cpu_baseline') or {}).get('value'))"; done
