#!/bin/bash
# instruction-cache counters of the two headline kernels (hypothesis: the ~60-100 KB kernels thrash the shared 64 KB instruction cache)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
export PQT_BENCH_NO_PIPELINE=1
rocprofv3 -L 2>/dev/null | grep -E "Counter_Name.*(ICACHE|IFETCH|INST_LEVEL|SQC_INST)" | sed "s/^\s*//" | sort -u | head -30
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_SENDMSG" "SQ_INSTS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC"; do
  rm -rf /tmp/prof_ic
  timeout 420 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/prof_ic -o ic -- python bench.py --no-cpu --no-gt --no-hbm-leg --steps 5 --warmup 2 "$@" > /dev/null 2> gpurun_out/r04/pmc_ic.log
  python - <<PY
import csv, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/prof_ic/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'pqt_k_rerank_select' in r['Kernel_Name'] or 'pqt_k_traverse' in r['Kernel_Name']:
            agg[r['Kernel_Name'].split('(')[0].replace('void ', '')][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in sorted(agg.items()):
    print(k[:70], {c: round(sum(v) / len(v)) for c, v in sorted(d.items())})
PY
done
