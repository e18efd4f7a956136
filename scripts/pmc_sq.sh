#!/bin/bash
# SQ counter passes for the two fused kernels (8 SQ counters per pass)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
run() { # name, counters
  timeout 150 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d /tmp/pmc_$1 -o p -- python bench.py --no-cpu --steps 3 --warmup 1 $BENCH_ARGS > /dev/null 2> gpurun_out/prof/pmc_$1.log
  python - <<PY | tee -a gpurun_out/prof/pmc_summary.txt
import csv, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/pmc_$1/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if "pqt_k_traverse" in r["Kernel_Name"] or "pqt_k_rerank_select" in r["Kernel_Name"]:
            agg[r['Kernel_Name'].split('<')[0].replace('void ','')][r['Counter_Name']].append(float(r['Counter_Value']))
for k in agg:
    print(k, {c: round(sum(v)/len(v)) for c, v in sorted(agg[k].items())})
PY
}
run a "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM"
run b "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY"
run c "TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE"
run d "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
run e "TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"
run f "TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_WAVEFRONTS_sum SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD"
