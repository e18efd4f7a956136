#!/bin/bash
# final validation of the round: the whole -m gpu suite, the profile refresh (scripts/r04_profile_all.sh), the 8-shard one-device harness
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -6
timeout 3000 bash scripts/r04_profile_all.sh 2>&1 | tail -60
python - <<'PY'
import json
d = json.load(open('gpurun_out/bench/r04_bench_default_extras.json'))
e = d['config'].get('heuristic_2d_512'); print('2d leg', e and (round(e['queries_per_sec']), e['stage_ms'], e['recall@1'], e['recall@100'], e['mean_candidates'], e['kernel_path']))
PY
timeout 900 python scripts/r03_shard8_one_device.py > gpurun_out/r04/r04_shard8_one_device_synth10m.json 2> gpurun_out/r04/shard8.err || tail -5 gpurun_out/r04/shard8.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r04/r04_shard8_one_device_synth10m.json'))
for kn, r in d['knobs'].items():
    print(kn, 'unsharded', r['unsharded'], 'per-rank', r['per_rank_ms_query_sharded'], 'speedup', r['speedup_query_sharded'], 'shard rerank', [p['query_sharded']['rerank_select_ms'] for p in r['per_shard']], r.get('merged_identical_to_unsharded'))
PY
