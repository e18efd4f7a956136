"""Slot timeline of a rerank launch from the debug timestamps (gpurun_out/tstamps.npy written by PQT_TSTAMP=1 bench.py)."""
import sys, numpy as np
ts = np.load(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/tstamps.npy')
w13, w14 = ts[:, 13], ts[:, 14]
start = (w13 >> 32).astype(np.int64); end = (w14 >> 32).astype(np.int64); slot = (w14 & 0xffff).astype(np.int64)
t0 = start.min(); ns = slot.max() + 1
busy = np.zeros(ns); last = np.zeros(ns); cnt = np.zeros(ns)
np.add.at(busy, slot, end - start); np.add.at(cnt, slot, 1); np.maximum.at(last, slot, end - t0)
print("queries %d  slots %d  launch span %.1f us  queries/slot min %d max %d" % (len(ts), ns, (end.max() - t0) / 100, cnt.min(), cnt.max()))
print("slot busy us: mean %.1f min %.1f max %.1f | slot last end us: min %.1f p10 %.1f med %.1f p90 %.1f max %.1f"
      % (busy.mean() / 100, busy.min() / 100, busy.max() / 100, last.min() / 100, np.percentile(last, 10) / 100, np.median(last) / 100, np.percentile(last, 90) / 100, last.max() / 100))
