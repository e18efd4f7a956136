#!/bin/bash
# how the two launches of two batches in flight share the device: rocprofv3 kernel trace of the default headline command, then per timed step
# the time with both kernels running, with one of them alone, and with none (from the dispatches' start / end timestamps)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PQT_BENCH_NO_PIPELINE=1
mkdir -p gpurun_out/prof
rm -rf /tmp/prof_tl
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tl -o tl -- python bench.py --no-cpu --no-hbm-leg --no-live-traffic --no-gt --steps 40 --warmup 4 > /dev/null 2> gpurun_out/prof/r04_overlap_timeline.log
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/prof_tl/**/*kernel_trace.csv', recursive=True)[0]
ev = []
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    kind = 'T' if 'pqt_k_traverse' in n else ('R' if 'pqt_k_rerank_select' in n else None)
    if kind: ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), kind))
ev.sort()
ev = ev[-80:]   # the 40 timed steps (the last 80 query launches of the run)
t0, t1 = ev[8][0], ev[-8][1]   # steady state: a few launches in from both ends
pts = sorted(set([t0, t1] + [x for e in ev for x in e[:2] if t0 <= x <= t1]))
acc = {}
for a, b in zip(pts, pts[1:]):
    m = (a + b) // 2
    live = ''.join(sorted(k for s, e, k in ev if s <= m < e))
    acc[live] = acc.get(live, 0) + (b - a)
tot = t1 - t0
nst = sum(1 for s, e, k in ev if k == 'R' and t0 <= s and e <= t1)
out = ["steady-state window: %.3f ms, %d complete rerank launches (%.4f ms per step)" % (tot / 1e6, nst, tot / 1e6 / max(nst, 1))]
names = {'': 'no kernel running', 'T': 'a traversal alone', 'R': 'a rerank alone', 'RT': 'a traversal and a rerank', 'TT': 'two traversals', 'RR': 'two reranks', 'RTT': 'rerank + two traversals', 'RRT': 'two reranks + a traversal'}
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    out.append("  %-28s %5.1f %%   %.4f ms per step" % (names.get(k, k), 100.0 * v / tot, v / 1e6 / max(nst, 1)))
dur = {}
for s, e, k in ev: dur.setdefault(k, []).append(e - s)
out.append("  mean launch: traversal %.4f ms, rerank %.4f ms" % (sum(dur['T']) / len(dur['T']) / 1e6, sum(dur['R']) / len(dur['R']) / 1e6))
open('gpurun_out/prof/r04_overlap_timeline.txt', 'w').write('\n'.join(out) + '\n')
print('\n'.join(out))
PY
