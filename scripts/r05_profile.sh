#!/bin/bash
# rocprofv3 recipe of round 5 (= round 4 + a TCC_HIT / TCC_MISS pass for every pqt_k_ kernel) (run on the GPU box through gpurun):
#   bash scripts/r05_profile.sh <tag> <fetch_factor> <workload> <bv> <bb> <k> [more bench args]
# kernel trace + stats of the bench command, then FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (MI355X_MICROARCH.md:
# the two do not fit one pass; gfx950 FETCH_SIZE counts a wide coalesced 16 B/lane stream at half its bytes -> fetch_factor 2
# for the group-major / 128-byte-row kernels, 1 for the 64-byte row gathers calibrated in profiles/r01_pmc_calibration.json).
# Summaries land in gpurun_out/prof/<tag>_*; copy what should be judged into profiles/.
tag=$1; ff=$2; wl=$3; bv=$4; bb=$5; k=$6; shift 6
# overlap=0: every call in one piece, so that the kernel statistics average full-size launches only (the library's default splits an
# untimed SIFT1M-shape batch into two half-size launches per kernel; bench.py's own per-kernel numbers come from the one-piece timed calls)
# --pipeline 1: one batch at a time, every kernel alone on the device (the per-kernel durations behind roofline.one_batch_at_a_time; the
# default line's two batches in flight are profiled by scripts/r04_profile_inflight.sh)
args="--workload $wl --bv $bv --bb $bb --k $k --option overlap=0 --pipeline 1 $@"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PQT_BENCH_NO_PIPELINE=1   # only the headline launches in the kernel statistics (no half-batch two-stream leg)
mkdir -p gpurun_out/prof
if [ -z "$PQT_PROFILE_PMC_ONLY" ]; then
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o $tag -- python bench.py --no-cpu --steps 10 --warmup 3 $args > gpurun_out/prof/${tag}_bench_under_rocprof.json 2> gpurun_out/prof/${tag}_bench.log
cp /tmp/prof_$tag/${tag}_kernel_stats.csv gpurun_out/prof/ 2>/dev/null
fi
for c in FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum; do
  # (rocprofv3 --pmc occasionally hangs at process start on this image: bounded, one retry)
  for attempt in 1 2; do
    rm -rf /tmp/prof_${tag}_$c
    timeout ${PQT_PMC_TIMEOUT:-240} rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/prof_${tag}_$c -o $tag -- python bench.py --no-cpu --no-gt --steps 5 --warmup 2 $args > /dev/null 2> gpurun_out/prof/${tag}_pmc_$c.log && break
  done
  python - <<PY
import csv, collections, glob
fn = glob.glob('/tmp/prof_${tag}_$c/*counter_collection.csv')
agg = collections.defaultdict(list)
for f in fn:
    for r in csv.DictReader(open(f)):
        if 'pqt_k_' in r['Kernel_Name'] and r['Counter_Name'] == '$c':
            agg[r['Kernel_Name'].split('(')[0]].append(float(r['Counter_Value']))
with open('gpurun_out/prof/${tag}_pmc_$c.csv', 'w') as o:
    o.write('kernel,dispatches,mean_$c,min,max\n')
    for k, v in sorted(agg.items()):
        # query launches only: the build kernels (assign_encode, reorder, group_major, adc_bias, coarse) are listed too
        o.write('"%s",%d,%.1f,%.1f,%.1f\n' % (k, len(v), sum(v) / len(v), min(v), max(v)))
print(open('gpurun_out/prof/${tag}_pmc_$c.csv').read()[:1500])
PY
done
python - <<PY
import csv, json
k = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE', 'TCC_HIT_sum', 'TCC_MISS_sum', 'TCC_EA0_RDREQ_sum'):
    for r in csv.DictReader(open('gpurun_out/prof/${tag}_pmc_%s.csv' % c)):
        k.setdefault(r['kernel'].replace('void ', '').split('<')[0], {})[c + ('_KiB' if c.endswith('SIZE') else '')] = float(r['mean_' + c])
json.dump({"workload": "$wl", "bv": $bv, "bb": $bb, "k": $k, "extra_args": "$@", "fetch_factor": $ff,
           "source": "scripts/r05_profile.sh $tag: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), mean per dispatch", "kernels": k},
          open('gpurun_out/prof/${tag}_pmc.json', 'w'), indent=1)
PY
grep pqt_k gpurun_out/prof/${tag}_kernel_stats.csv | cut -c1-220
cut -c1-600 gpurun_out/prof/${tag}_bench_under_rocprof.json
