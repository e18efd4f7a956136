#!/bin/bash
# same-box A/B of tune/lib_*.so on one range shard of eight (configs[2] shape, 10 M vectors) + the unsharded index: stage times,
# per-query phase clocks, and the result bytes of every library compared with the first one
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
W=${PQT_SHARD_WORKLOAD:-synth10m}
for f in tune/lib_*.so; do
  t=$(basename $f .so)
  echo "== $t"
  PQT_LIB=$PWD/$f PQT_SHARD_WORKLOAD=$W PQT_SHARDS_MEASURED=1 PQT_SHARD_DUMP=gpurun_out/r04/sh_$t timeout 900 python scripts/r03_shard8_one_device.py > gpurun_out/r04/sh_$t.json 2> gpurun_out/r04/sh_$t.err || tail -5 gpurun_out/r04/sh_$t.err
  PQT_LIB=$PWD/$f PQT_SHARD_WORKLOAD=$W PQT_SHARDS_MEASURED=1 PQT_TSTAMP=1 timeout 900 python scripts/r03_shard8_one_device.py > gpurun_out/r04/sh_${t}_ts.json 2> gpurun_out/r04/sh_${t}_ts.err || tail -5 gpurun_out/r04/sh_${t}_ts.err
done
python - <<'PY'
import json, glob, numpy as np
tags = sorted(glob.glob('gpurun_out/r04/sh_lib_*[!s].json'))
for f in tags:
    d = json.load(open(f)); t = f.split('sh_')[1][:-5]
    for kn, r in d['knobs'].items():
        p = r['per_shard'][0]
        line = [t, kn, 'unsharded', r['unsharded']['rerank_select_ms'], 'shard', p['query_sharded']['rerank_select_ms'], p['replicated']['rerank_select_ms'], p['query_sharded']['path'], p['query_sharded']['identical_to_replicated']]
        try:
            c = json.load(open(f[:-5] + '_ts.json'))['knobs'][kn]['shard0_rerank_clocks_median']; line.append(c); line.append(json.load(open(f[:-5] + '_ts.json'))['knobs'][kn].get('shard0_rerank_timeline_us'))
        except Exception as e: line.append(str(e))
        print(*line)
import os
base = None
for f in sorted(glob.glob('gpurun_out/r04/sh_lib_*.npz')):
    z = np.load(f); t = os.path.basename(f)
    kn = t.split('_', 3)[-1] if False else '_'.join(t[:-4].split('_')[-2:])
    if base is None or base[0] != kn and not any(b[0] == kn for b in [base]): pass
    key = kn
    globals().setdefault('B', {})
    if key not in B: B[key] = (t, {k: z[k] for k in z.files}); continue
    same = {k: bool(np.array_equal(B[key][1][k], z[k])) for k in z.files}
    print('compare', t, 'vs', B[key][0], same)
PY
rm -f gpurun_out/r04/sh_lib_*.npz
