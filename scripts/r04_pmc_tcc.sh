#!/bin/bash
# L2 / memory-side request counters of the 100 M-vector rerank launch: two --pmc passes (4 TCC slots each) with
# --kernel-trace only (no --stats / other trace domains together with --pmc)
#   bash scripts/r04_pmc_tcc.sh <bv> <bb>
bv=$1; bb=$2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
export PQT_BENCH_NO_PIPELINE=1
for pass in "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_REQ_sum TCC_READ_sum TCC_EA0_RDREQ_GMI_32B_sum TCC_EA0_RDREQ_IO_32B_sum"; do
  tag=$(echo $pass | cut -d' ' -f1)
  for attempt in 1 2; do
    rm -rf /tmp/prof_tcc
    timeout 420 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/prof_tcc -o tcc -- python bench.py --workload synth100m --bv $bv --bb $bb --no-cpu --no-gt --no-hbm-leg --steps 5 --warmup 2 > /dev/null 2> gpurun_out/r04/pmc_tcc_$tag.log && break
  done
  python - <<PY
import csv, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/prof_tcc/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'pqt_k_rerank_select' in r['Kernel_Name'] or 'pqt_k_traverse' in r['Kernel_Name']:
            agg[r['Kernel_Name'].split('(')[0].replace('void ', '')][r['Counter_Name']].append(float(r['Counter_Value']))
with open('gpurun_out/r04/r04_cfg3_100m_${bv}_${bb}_tcc.csv', 'a') as o:
    for k, d in sorted(agg.items()):
        for c, v in sorted(d.items()):
            o.write('"%s",%s,%d,%.1f\n' % (k, c, len(v), sum(v) / len(v)))
            print(k[:60], c, len(v), sum(v) / len(v))
PY
done
