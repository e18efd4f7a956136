#!/bin/bash
# instruction-cache and SQ counters of the configs[2]-shape filter kernel (pqt_rs_query MODE 2) on one range shard of eight, for
# every tune/lib_*.so: is the ~70-100 KB kernel thrashing the 64 KB instruction cache two CUs share?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
for f in tune/lib_*.so; do
  t=$(basename $f .so)
  for pass in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_INSTS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"; do
    rm -rf /tmp/prof_ic
    PQT_LIB=$PWD/$f PQT_SHARDS_MEASURED=1 timeout 600 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/prof_ic -o ic -- python scripts/r03_shard8_one_device.py > /dev/null 2> gpurun_out/r04/pmc_ic_shard.log
    python - "$t" <<'PY'
import csv, collections, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/prof_ic/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'pqt_k_rerank_select<' in r['Kernel_Name']:
            agg[r['Kernel_Name'].split('(')[0].replace('void ', '') + ' grid=' + r.get('Grid_Size', '?')][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in sorted(agg.items()):
    print(sys.argv[1], k[:90], 'launches', len(next(iter(d.values()))), {c: round(sum(v) / len(v)) for c, v in sorted(d.items())})
PY
  done
done
