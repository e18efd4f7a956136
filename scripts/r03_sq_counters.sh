#!/bin/bash
# SQ counters of the query kernels (one --pmc pass, 8 SQ slots): where the wavefront cycles go (active / waiting), VALU and LDS share
#   bash scripts/r03_sq_counters.sh <tag> <workload> [bench args]
tag=$1; wl=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
export PQT_BENCH_NO_PIPELINE=1
C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS"
for attempt in 1 2; do
  rm -rf /tmp/prof_sq_$tag
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/prof_sq_$tag -o $tag -- python bench.py --workload $wl --no-cpu --no-gt --no-hbm-leg --steps 5 --warmup 2 --option overlap=0 "$@" > /dev/null 2> gpurun_out/prof/${tag}_sq.log && break
done
python - <<PY
import csv, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/prof_sq_$tag/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'pqt_k_traverse' in r['Kernel_Name'] or 'pqt_k_rerank_select' in r['Kernel_Name']:
            agg[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
names = "$C".split()
with open('gpurun_out/prof/${tag}_sq_counters.csv', 'w') as o:
    o.write('kernel,dispatches,' + ','.join(names) + ',active_frac,wait_any_frac,wait_inst_frac,valu_share_of_active,lds_share_of_active\n')
    for k, v in sorted(agg.items()):
        m = {n: (sum(v[n]) / len(v[n]) if v[n] else 0.0) for n in names}
        wc = max(m['SQ_WAVE_CYCLES'], 1.0); ac = max(m['SQ_ACTIVE_INST_ANY'], 1.0)
        o.write('"%s",%d,' % (k, len(v[names[0]])) + ','.join('%.0f' % m[n] for n in names) +
                ',%.3f,%.3f,%.3f,%.3f,%.3f\n' % (m['SQ_ACTIVE_INST_ANY'] / wc, m['SQ_WAIT_ANY'] / wc, m['SQ_WAIT_INST_ANY'] / wc, m['SQ_ACTIVE_INST_VALU'] / ac, m['SQ_ACTIVE_INST_LDS'] / ac))
print(open('gpurun_out/prof/${tag}_sq_counters.csv').read())
PY
