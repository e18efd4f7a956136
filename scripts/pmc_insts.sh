cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PQT_BENCH_NO_PIPELINE=1
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d /tmp/pmc_x -o p -- python bench.py --no-cpu --steps 3 --warmup 1 > /dev/null 2>/dev/null
python - <<PY
import csv, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/pmc_x/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if "pqt_k_traverse" in r["Kernel_Name"] or "pqt_k_rerank_select" in r["Kernel_Name"]:
            agg[r['Kernel_Name'].split('<')[0].replace('void ','')][r['Counter_Name']].append(float(r['Counter_Value']))
for k in agg:
    print(k, {c: round(sum(v)/len(v)) for c, v in sorted(agg[k].items())})
PY
