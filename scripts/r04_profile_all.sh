#!/bin/bash
# round 3 profile refresh (final binary): kernel statistics + FETCH/WRITE PMC passes of the headline and of the 100 M configuration at
# both knob sets, then the plain bench lines (default command = headline + hbm_roofline_leg, extras, 10 M with cpu_baseline)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof gpurun_out/bench
bash scripts/r04_profile.sh r04_cfg2_sift1m 1.0 sift1m 20000 500 100 > gpurun_out/prof_a.log 2>&1
bash scripts/r04_profile.sh r04_cfg3_100m_20000_500 2.0 synth100m 20000 500 100 > gpurun_out/prof_b.log 2>&1
bash scripts/r04_profile.sh r04_cfg3_100m_4096_4096 2.0 synth100m 4096 4096 100 > gpurun_out/prof_c.log 2>&1
tail -4 gpurun_out/prof_a.log gpurun_out/prof_b.log gpurun_out/prof_c.log | cut -c1-300
# the default headline (two whole batches in flight on one device): kernel statistics of overlapped launches only
bash scripts/r04_profile_inflight.sh r04_cfg2_sift1m_inflight > gpurun_out/prof_d.log 2>&1
tail -3 gpurun_out/prof_d.log | cut -c1-300
# the default command under the kernel trace (both the SIFT1M-shape and the 100 M launches appear, as different instantiations);
# overlap=0: one-piece calls only, see r04_profile.sh
cd /tmp && PQT_BENCH_NO_PIPELINE=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_default -o r04_default -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --option overlap=0 --pipeline 1 > $GRAFT_REPO_ROOT/gpurun_out/prof/r04_default_bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof/r04_default_bench.log; cd $GRAFT_REPO_ROOT
cp /tmp/prof_default/r04_default_kernel_stats.csv gpurun_out/prof/ 2>/dev/null
grep pqt_k gpurun_out/prof/r04_default_kernel_stats.csv | cut -c1-200
python bench.py --live-traffic-hbm 2> gpurun_out/bench/default.log | grep '^{"metric' > gpurun_out/bench/r04_bench_default.json
python bench.py --extras --no-hbm-leg 2> gpurun_out/bench/extras.log | grep '^{"metric' > gpurun_out/bench/r04_bench_default_extras.json
python bench.py --workload synth10m 2> gpurun_out/bench/s10m.log | grep '^{"metric' > gpurun_out/bench/r04_bench_synth10m.json
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/bench/r04_*.json')):
    d = json.load(open(f)); c = d['config']
    print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'], 4), {k: round(v, 4) for k, v in c['stage_ms'].items() if v}, 'frac', round(d['roofline']['frac'], 3),
          'stream', d['roofline'].get('measured_stream_GBps'), 'cpu', (d.get('cpu_baseline') or {}).get('value'))
    leg = c.get('hbm_roofline_leg')
    if leg:
        for kk in ('knobs_20000_500', 'knobs_4096_4096'):
            e = leg[kk]; print('   hbm leg', kk, round(e['queries_per_sec']), round(e['ms_per_step'], 3), {k: round(v, 4) for k, v in e['stage_ms'].items() if v}, 'frac', round(e['roofline']['frac'], 3), 'recall@1', e['recall@1'])
    for kk in ('knobs_4096_4096', 'knobs_4096_4096_k4096', 'knobs_4096_4096_k4096_staged'):
        e = c.get(kk)
        if e: print('  ', kk, round(e['queries_per_sec']), {k: round(v, 4) for k, v in e['stage_ms'].items() if v}, e.get('kernel_path'))
PY
