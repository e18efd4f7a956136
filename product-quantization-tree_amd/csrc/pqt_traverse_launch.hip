// pqt_traverse_launch.hip -- instantiations and launcher of the fused traversal kernel (pqt_k_traverse): a translation unit of its
// own for build time, see pqt_internal.h.
#include "pqt_internal.h"

// ---- fused traversal (pqt_k_traverse): LDS plan and launch, shared by pqt_query*, pqt_traverse_bins and pqt_query_shard_bins
int planTraversal(pqt_index* idx, uint32_t He, TravPlan& tp) {
  const PqtDevParams& d = idx->dp;
  int rc;
  // fused traversal (wave per query) when the bin list fits the in-register sorter
  tp.fused = (He <= 4096) && (d.WC <= 256) && !idx->forceUnfused && !idx->heur2d;  // (2-D sequences: per-query row tables, staged kernels)
  tp.wide = tp.fused && He > 512;  // rows enumerated in blocks of 512, populated ones listed (<= 512), overflow -> pqt_k_bins
  const size_t travR0 = (std::max<size_t>(tp.wide ? 2 * 512 * 8 : 512 * 8 + (idx->sharded ? 512 * 4 : 0), 4 * (size_t)(d.LP * d.C1 + d.P * d.WC + d.D)) + 15) & ~(size_t)15;
  tp.perWave = (uint32_t)(travR0 + ((4 * (size_t)(d.P * d.C1 + d.P * d.W + 2 * d.P * d.WC) + 15) & ~(size_t)15));
  tp.lTrav = (size_t)kTravWaves * tp.perWave;
  const size_t lTrav = tp.lTrav;
  auto isP2 = [](uint32_t x) { return x != 0 && (x & (x - 1)) == 0; };
  // all layout strides powers of two (every BASELINE shape): the traversal uses shifts and masks
  tp.p2 = isP2(d.C1) && isP2(d.C2) && isP2(d.W) && isP2(d.LP) && isP2(d.D) && isP2(d.S) && isP2(d.SS) && isP2(d.R) && d.S >= 4;
#ifndef PQT_DEV_SIFT1M_ONLY
  if (tp.fused && lTrav > 64 * 1024) {
    if ((rc = allowLds(pqt_k_traverse<kTravWaves, 1, false, false>, lTrav))) return rc;
    if ((rc = allowLds(pqt_k_traverse<kTravWaves, 2, false, false>, lTrav))) return rc;
    if ((rc = allowLds(pqt_k_traverse<kTravWaves, 4, false, false>, lTrav))) return rc;
    if ((rc = allowLds(pqt_k_traverse<kTravWaves, 1, true, false>, lTrav))) return rc;
    if ((rc = allowLds(pqt_k_traverse<kTravWaves, 2, true, false>, lTrav))) return rc;
    if ((rc = allowLds(pqt_k_traverse<kTravWaves, 4, true, false>, lTrav))) return rc;
    if ((rc = allowLds(pqt_k_traverse<kTravWaves, 1, false, true>, lTrav))) return rc;
    if ((rc = allowLds(pqt_k_traverse<kTravWaves, 2, false, true>, lTrav))) return rc;
    if ((rc = allowLds(pqt_k_traverse<kTravWaves, 4, false, true>, lTrav))) return rc;
    if ((rc = allowLds(pqt_k_traverse<kTravWaves, 1, true, true>, lTrav))) return rc;
    if ((rc = allowLds(pqt_k_traverse<kTravWaves, 2, true, true>, lTrav))) return rc;
    if ((rc = allowLds(pqt_k_traverse<kTravWaves, 4, true, true>, lTrav))) return rc;
    if ((rc = allowLds(pqt_k_traverse<kTravWaves, 1, false, true, 1>, lTrav))) return rc;
    if ((rc = allowLds(pqt_k_traverse<kTravWaves, 1, true, true, 1>, lTrav))) return rc;
    if ((rc = allowLds(pqt_k_traverse<kTravWaves, 1, false, true, 2>, lTrav))) return rc;
    if ((rc = allowLds(pqt_k_traverse<kTravWaves, 1, true, true, 2>, lTrav))) return rc;
  }
#else
  (void)rc;
  if (tp.fused && lTrav > 64 * 1024) return pqtFail(PQT_ERR_LIMIT, "development build: SIFT1M shape only");
#endif
  return PQT_OK;
}
// compile-time-shape instantiation the traversal of this index runs (0: run-time shape): the two BASELINE shapes, two-phase
// enumeration only (packed heuristic rows and the presence bitmap must be there, no modulo hashing, no order-all-rows switch)
int travShape(const pqt_index* idx, const PqtTravArgs& targs) {
  const bool twoOk = targs.heur4 && targs.filter && !idx->dp.hashMod && !((idx->dbg >> 5) & 1u);
  return (idx->noShape || !twoOk) ? 0 : pqt_shape_of(idx->dp);
}
// a1..a6 in one launch, one wavefront per query (`waves` of them); ev0 / ev1 ride on the dispatch when given
void launchFusedTraversal(pqt_index* idx, const PqtTravArgs& targs, const TravPlan& tp, uint32_t waves, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
  const PqtDevParams& d = idx->dp;
  const uint32_t grid = (waves + kTravWaves - 1) / kTravWaves;
  const size_t lTrav = tp.lTrav;
  const uint32_t travPerWave = tp.perWave;
  const bool travP2 = tp.p2;
  idx->lastTravF1 = false;
#define PQT_LAUNCH_TR1(WCR, SH, PP)                                                                                       \
  hipExtLaunchKernelGGL((pqt_k_traverse<kTravWaves, WCR, SH, PP>), dim3(grid), dim3(kTravWaves * 64), (uint32_t)lTrav, st, ev0, ev1, 0u, \
                        targs, travPerWave)
#define PQT_LAUNCH_TR(WCR) do { if (idx->sharded) { if (travP2) PQT_LAUNCH_TR1(WCR, true, true); else PQT_LAUNCH_TR1(WCR, true, false); } \
                                else { if (travP2) PQT_LAUNCH_TR1(WCR, false, true); else PQT_LAUNCH_TR1(WCR, false, false); } } while (0)
  const int shape = travShape(idx, targs);
#ifndef PQT_DEV_SIFT1M_ONLY
  // wide enumeration at the two BASELINE shapes: kF1Waves wavefronts per workgroup behind one LDS copy of the bitmap's first level
  constexpr int kF1Waves = 8;
  if (tp.wide && targs.filter1 && targs.filter && (shape == 1 || shape == 2)) {
    const size_t f1Bytes = (size_t)1 << (targs.filter1Bits - 3);
    const size_t lds = f1Bytes + (size_t)kF1Waves * travPerWave;
    if (lds <= kMaxLds) {
      const uint32_t g1 = (waves + kF1Waves - 1) / kF1Waves;
#define PQT_LAUNCH_F1(SH, SHAPEV)                                                                                              \
      do { auto kern = pqt_k_traverse_f1<kF1Waves, 1, SH, true, SHAPEV>;                                                        \
           if (pqtAllowLds((const void*)kern, lds) == PQT_OK) {                                                                  \
             hipExtLaunchKernelGGL(kern, dim3(g1), dim3(kF1Waves * 64), (uint32_t)lds, st, ev0, ev1, 0u, targs, travPerWave);    \
             idx->lastTravF1 = true; return; } } while (0)
      if (shape == 1) { if (idx->sharded) PQT_LAUNCH_F1(true, 1); else PQT_LAUNCH_F1(false, 1); }
      else { if (idx->sharded) PQT_LAUNCH_F1(true, 2); else PQT_LAUNCH_F1(false, 2); }
#undef PQT_LAUNCH_F1
    }
  }
#endif
#ifdef PQT_DEV_SIFT1M_ONLY
  // development builds (scripts/r04_devlib.sh): only the SIFT1M-shape instantiation; anything else launches nothing
  (void)d; (void)travP2;
  if (shape == 1 && !idx->sharded) hipExtLaunchKernelGGL((pqt_k_traverse<kTravWaves, 1, false, true, 1>), dim3(grid), dim3(kTravWaves * 64), (uint32_t)lTrav, st, ev0, ev1, 0u, targs, travPerWave);
  return;
#else
  if (shape == 1) { if (idx->sharded) hipExtLaunchKernelGGL((pqt_k_traverse<kTravWaves, 1, true, true, 1>), dim3(grid), dim3(kTravWaves * 64), (uint32_t)lTrav, st, ev0, ev1, 0u, targs, travPerWave);
                    else hipExtLaunchKernelGGL((pqt_k_traverse<kTravWaves, 1, false, true, 1>), dim3(grid), dim3(kTravWaves * 64), (uint32_t)lTrav, st, ev0, ev1, 0u, targs, travPerWave); }
  else if (shape == 2) { if (idx->sharded) hipExtLaunchKernelGGL((pqt_k_traverse<kTravWaves, 1, true, true, 2>), dim3(grid), dim3(kTravWaves * 64), (uint32_t)lTrav, st, ev0, ev1, 0u, targs, travPerWave);
                         else hipExtLaunchKernelGGL((pqt_k_traverse<kTravWaves, 1, false, true, 2>), dim3(grid), dim3(kTravWaves * 64), (uint32_t)lTrav, st, ev0, ev1, 0u, targs, travPerWave); }
  else if (d.WC <= 64) PQT_LAUNCH_TR(1); else if (d.WC <= 128) PQT_LAUNCH_TR(2); else PQT_LAUNCH_TR(4);
#endif
#undef PQT_LAUNCH_TR
#undef PQT_LAUNCH_TR1
}
