"""Timeline of pqt_k_rerank_select from the debug timestamps (gpurun_out/tstamps.npy of a PQT_TSTAMP=1 run).
Record words: [9] start (shader clocks), [10] rows wait, [11] ADC + filter, [12] flushes, [13] clocks of the query | wall start << 32,
[14] wave slot | XCC << 16 | wall end << 32 (wall clock: 100 MHz, global; the shader-clock counters differ per XCD)."""
import numpy as np, sys
ts = np.load(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/tstamps.npy")
NW = int(sys.argv[2]) if len(sys.argv) > 2 else 12
r = ts[:, 9:15]
load, adc, flush = [r[:, i].astype(np.int64) for i in (1, 2, 3)]
tot = (r[:, 4] & 0xffffffff).astype(np.int64)
w0 = (r[:, 4] >> 32).astype(np.int64); w1 = (r[:, 5] >> 32).astype(np.int64)
slot = (r[:, 5] & 0xffff).astype(np.int64); xcc = ((r[:, 5] >> 16) & 0xf).astype(np.int64)
w1 = np.where(w1 < w0, w1 + (1 << 32), w1)
t0 = w0.min()
wg = slot // NW
nwg = int(wg.max()) + 1
span = (w1.max() - t0) * 10
print("queries %d workgroups %d | launch span (first start .. last end) %.1f us | per-query clocks median %d mean %d | sum/slots %d"
      % (len(ts), nwg, span / 1000, np.median(tot), tot.mean(), tot.sum() // (nwg * NW)))
wg_end = np.array([(w1[wg == g].max() - t0) * 10 / 1000 for g in range(nwg)])
wg_first = np.array([(w0[wg == g].min() - t0) * 10 / 1000 for g in range(nwg)])
wg_x = np.array([np.bincount(xcc[wg == g]).argmax() for g in range(nwg)])
wg_n = np.array([(wg == g).sum() for g in range(nwg)])
# wave-slot idle at the end: launch end minus the slot's last end
sl_end = np.array([(w1[slot == s_].max() - t0) * 10 / 1000 if (slot == s_).any() else 0 for s_ in range(nwg * NW)])
print("workgroup end (us): min %.1f p10 %.1f median %.1f p90 %.1f max %.1f | first query start: median %.1f max %.1f" % (wg_end.min(), np.percentile(wg_end, 10), np.median(wg_end), np.percentile(wg_end, 90), wg_end.max(), np.median(wg_first), wg_first.max()))
print("wave-slot last end (us): min %.1f p10 %.1f median %.1f mean %.1f max %.1f  -> mean idle before the launch end %.1f us (%.0f %%)"
      % (sl_end.min(), np.percentile(sl_end, 10), np.median(sl_end), sl_end.mean(), sl_end.max(), sl_end.max() - sl_end.mean(), 100 * (1 - sl_end.mean() / sl_end.max())))
for x in sorted(set(wg_x.tolist())):
    m = wg_x == x; mq = xcc == x
    print(" XCC %d: workgroups %3d queries %5d (per wg %d..%d) | per-query clocks median %6d rows %5d | wg end median %.1f max %.1f us"
          % (x, m.sum(), mq.sum(), wg_n[m].min(), wg_n[m].max(), np.median(tot[mq]), np.median(load[mq]), np.median(wg_end[m]), wg_end[m].max()))
