#!/bin/bash
# rocprofv3 recipe (run on the GPU box through gpurun): kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in
# separate PMC passes.  Summaries land in gpurun_out/prof/<tag>_*; copy what should be judged into profiles/.
tag=${1:-r01}
shift
args="$@"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PQT_BENCH_NO_PIPELINE=1   # only the headline launches in the kernel statistics (no half-batch two-stream leg)
mkdir -p gpurun_out/prof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o $tag -- python bench.py --no-cpu $args > gpurun_out/prof/${tag}_bench.json 2> gpurun_out/prof/${tag}_bench.log
cp /tmp/prof_$tag/${tag}_kernel_stats.csv gpurun_out/prof/ 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/prof_${tag}_$c -o $tag -- python bench.py --no-cpu --steps 5 --warmup 2 $args > /dev/null 2> gpurun_out/prof/${tag}_pmc_$c.log
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
        o.write('"%s",%d,%.1f,%.1f,%.1f\n' % (k, len(v), sum(v) / len(v), min(v), max(v)))
print(open('gpurun_out/prof/${tag}_pmc_$c.csv').read())
PY
done
python - <<PY
import csv, json
k = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    for r in csv.DictReader(open('gpurun_out/prof/${tag}_pmc_%s.csv' % c)):
        k.setdefault(r['kernel'].replace('void ', '').split('<')[0], {})[c + '_KiB'] = float(r['mean_' + c])
json.dump({"workload": "sift1m", "bv": 20000, "bb": 500, "k": 100, "fetch_factor": 1.0,
           "source": "scripts/profile.sh ${tag}: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), mean per dispatch", "kernels": k},
          open('gpurun_out/prof/${tag}_pmc_latest.json', 'w'), indent=1)
PY
grep pqt_k gpurun_out/prof/${tag}_kernel_stats.csv | cut -c1-200
cut -c1-1500 gpurun_out/prof/${tag}_bench.json
