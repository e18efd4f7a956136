// pqt_rerank_launch.hip -- instantiations and launchers of the fused rerank + select kernels (pqt_k_rerank_select, its list
// variant, pqt_k_rerank_select_wg, pqt_k_rerank_select_big, pqt_k_rerank_sort_small).  A translation unit of its own so that
// these (the bulk of the library's compile time) build in parallel with the rest; see pqt_internal.h.
#include "pqt_internal.h"

#ifdef PQT_DEV_SIFT1M_ONLY
// Development builds only (scripts/r04_devlib.sh -> tune/lib_*.so, never the shipped library): just the kernel the SIFT1M-shape
// headline launches, for edit-compile-measure cycles of seconds instead of minutes.  Everything else reports PQT_ERR_LIMIT.
#ifndef PQT_RS_U16
#define PQT_RS_U16 4
#endif
int launchRerankSelect(pqt_index* idx, bool cl, uint32_t grid, size_t lds, hipStream_t st, const float* qL1virt, const uint32_t* nLocal,
                       uint64_t stride, uint32_t k, uint32_t nq, uint32_t* oI, float* oD, uint32_t* oP) {
  if (!(idx->dp.LP == 16 && cl && !idx->sharded && idx->dp.C1 == 32 && !idx->curRuns)) return pqtFail(PQT_ERR_LIMIT, "development build: SIFT1M shape only");
  auto kern = idx->curXCode ? pqt_k_rerank_select<kXcWaves, 4, 2, true, false, 5, 0, false, true, kXcSlots> : pqt_k_rerank_select<kFusedWaves, 4, PQT_RS_U16, true, false, 5>;
  const uint32_t nwv = idx->curXCode ? (uint32_t)kXcWaves : (uint32_t)kFusedWaves;
  int rc = allowLds(kern, lds);
  if (rc) return rc;
  const PqtRsArgs rargs{idx->curXCode ? idx->d_codesX : idx->d_codesBin, idx->d_ids, qL1virt, idx->d_coarse, idx->d_cand, idx->d_candPos, nLocal, stride, k, nq, idx->dp, oI, oD, oP,
                        idx->ctr, idx->dbg, (nq <= (1u << 16)) ? idx->d_tstamp : nullptr, idx->curDynamic, idx->curZero8,
                        nullptr, 0, nullptr, 0.f, 0.f, nullptr, nullptr, nullptr, nullptr,
                        nullptr, idx->d_runGpos, idx->d_nRuns, idx->curRunCap, idx->curPool, idx->curPoolNext, idx->curPool ? idx->curPool + 16 : nullptr, idx->d_schedList, idx->curSchedCap};
  hipExtLaunchKernelGGL(kern, dim3(grid), dim3(nwv * 64), (uint32_t)lds, st, idx->lev0, idx->lev1, 0u, rargs);
  return PQT_OK;
}
int launchRSBiasAny(pqt_index*, int, bool, uint32_t, size_t, hipStream_t, const float*, const uint32_t*, uint64_t, uint32_t, uint32_t, uint32_t*, float*, uint32_t*) { return pqtFail(PQT_ERR_LIMIT, "development build"); }
int rswgGroup(const PqtDevParams&) { return 0; }
int launchRSWGAny(pqt_index*, int, uint32_t, hipStream_t, const float*, const uint32_t*, uint64_t, uint32_t, uint32_t*, float*, uint32_t*) { return pqtFail(PQT_ERR_LIMIT, "development build"); }
int launchSmallLists(pqt_index*, bool, size_t, uint32_t, hipStream_t, const PqtRsArgs&, hipEvent_t) { return pqtFail(PQT_ERR_LIMIT, "development build"); }
int launchMidLists(pqt_index*, size_t, uint32_t, hipStream_t, const PqtRsArgs&, uint32_t*, uint32_t*) { return pqtFail(PQT_ERR_LIMIT, "development build"); }
int launchBigK(pqt_index*, bool, size_t, uint32_t, hipStream_t, const float*, const uint32_t*, uint64_t, uint32_t, uint32_t, uint32_t, uint32_t*, float*, uint32_t*, const uint32_t*, const uint32_t*, hipEvent_t, hipEvent_t) { return pqtFail(PQT_ERR_LIMIT, "development build"); }
#else

namespace {
#ifndef PQT_RS_U16
#define PQT_RS_U16 4   // candidates per lane in flight when LP = 16 (scaled so that U * LP/4 stays 16 code vectors)
#endif
template <int LPV, bool CL, bool SH>
int launchRS(pqt_index* idx, uint32_t grid, size_t lds, hipStream_t st, const float* qL1virt, const uint32_t* nLocal,
             uint64_t stride, uint32_t k, uint32_t nq, uint32_t* oI, float* oD, uint32_t* oP) {
  constexpr int U0 = (PQT_RS_U16 * 4) / LPV;
  // 64*UV keys are appended per batch behind a best list of up to PQT_RS_BEST keys: they must fit the pending area
  // (U = 8 at LP = 4/8 overran the wave's key slots when > 512 - k candidates of a batch beat tau)
  constexpr int UV = U0 < 1 ? 1 : (U0 > 4 ? 4 : U0);
  static_assert(64 * UV <= PQT_RS_PEND, "a batch of appended keys must fit the pending area");
  const uint32_t c1 = idx->dp.C1;
  const bool p2 = c1 > 1 && (c1 & (c1 - 1)) == 0;
  auto kern = p2 ? pqt_k_rerank_select<kFusedWaves, LPV, UV, CL, SH, 1> : pqt_k_rerank_select<kFusedWaves, LPV, UV, CL, SH, 0>;
  if constexpr (CL) { if (c1 == 32) kern = pqt_k_rerank_select<kFusedWaves, LPV, UV, true, SH, 5>; }  // compile-time C1 only where the table is in LDS
  // X-code rows: 16 wavefronts per workgroup (4 per SIMD) around 384 key slots each, two candidates per lane in flight when the rows are 64 bytes
  constexpr int UX = LPV >= 4 ? 2 : 4;
  static_assert(PQT_RS_BEST + 64 * UX <= kXcSlots, "a batch of appended keys must fit behind the best list");
  uint32_t nwv = (uint32_t)kFusedWaves;
  if constexpr (CL) { if (c1 == 32 && idx->curXCode) { kern = pqt_k_rerank_select<kXcWaves, LPV, UX, true, SH, 5, 0, false, true, kXcSlots>; nwv = (uint32_t)kXcWaves; } }
  if constexpr (CL && LPV == 4) { if (c1 == 32 && idx->curRuns) kern = pqt_k_rerank_select<kFusedWaves, LPV, UV, true, SH, 5, 0, true>; }  // experimental bin-runs variant
  int rc = allowLds(kern, lds);
  if (rc) return rc;
  // start/stop events ride on the dispatch packet itself (no separate event packets on the stream)
  const bool xc = CL && c1 == 32 && idx->curXCode;
  const PqtRsArgs rargs{xc ? idx->d_codesX : idx->d_codesBin, idx->d_ids, qL1virt, idx->d_coarse, idx->d_cand, idx->d_candPos, nLocal, stride, k, nq, idx->dp, oI, oD, oP,
                        idx->ctr, idx->dbg, (nq <= (1u << 16)) ? idx->d_tstamp : nullptr /* buffer holds 65536 query records */, idx->curDynamic, idx->curZero8,
                        nullptr, 0, nullptr, 0.f, 0.f, nullptr, nullptr, nullptr, nullptr,
                        (CL && idx->curRuns) ? idx->d_runs : nullptr, idx->d_runGpos, idx->d_nRuns, idx->curRunCap, idx->curPool, idx->curPoolNext, idx->curPool ? idx->curPool + 16 : nullptr, idx->d_schedList, idx->curSchedCap};
  hipExtLaunchKernelGGL(kern, dim3(grid), dim3(nwv * 64), (uint32_t)lds, st, idx->lev0, idx->lev1, 0u, rargs);
  return PQT_OK;
}
template <int LPV>
int launchRS1(pqt_index* idx, bool cl, uint32_t grid, size_t lds, hipStream_t st, const float* v, const uint32_t* nl,
              uint64_t stride, uint32_t k, uint32_t nq, uint32_t* oI, float* oD, uint32_t* oP) {
  if (cl) return idx->sharded ? launchRS<LPV, true, true>(idx, grid, lds, st, v, nl, stride, k, nq, oI, oD, oP)
                              : launchRS<LPV, true, false>(idx, grid, lds, st, v, nl, stride, k, nq, oI, oD, oP);
  return idx->sharded ? launchRS<LPV, false, true>(idx, grid, lds, st, v, nl, stride, k, nq, oI, oD, oP)
                      : launchRS<LPV, false, false>(idx, grid, lds, st, v, nl, stride, k, nq, oI, oD, oP);
}
}  // namespace
int launchRerankSelect(pqt_index* idx, bool cl, uint32_t grid, size_t lds, hipStream_t st, const float* v, const uint32_t* nl,
                       uint64_t stride, uint32_t k, uint32_t nq, uint32_t* oI, float* oD, uint32_t* oP) {
#ifdef PQT_DEV_CFG3_ONLY
  return pqtFail(PQT_ERR_LIMIT, "development build: configs[2]/[3] filter kernels only");
#else
  switch (idx->dp.LP / 4) {
    case 1: return launchRS1<1>(idx, cl, grid, lds, st, v, nl, stride, k, nq, oI, oD, oP);
    case 2: return launchRS1<2>(idx, cl, grid, lds, st, v, nl, stride, k, nq, oI, oD, oP);
    case 4: return launchRS1<4>(idx, cl, grid, lds, st, v, nl, stride, k, nq, oI, oD, oP);
    default: return launchRS1<8>(idx, cl, grid, lds, st, v, nl, stride, k, nq, oI, oD, oP);
  }
#endif
}

namespace {
// fused rerank+select without the coarse table in LDS (pqt_rs_query MODE 1 = opt-in adc_bias distances, MODE 2 = reference
// distances through the MODE 1 filter): group-major code words, NW wavefronts per workgroup
PqtRsArgs rsArgsFilter(pqt_index* idx, const float* qL1virt, const uint32_t* nLocal, uint64_t stride, uint32_t k, uint32_t nq,
                       uint32_t* oI, float* oD, uint32_t* oP) {
  const double lp = idx->dp.LP;
  const float kappa = (float)(2.02 * (lp * lp + 8.0 * lp + 2.0) / 16777216.0);
  return PqtRsArgs{idx->d_codesBin, idx->d_ids, qL1virt, idx->d_coarse, idx->d_cand, idx->d_candPos, nLocal, stride, k, nq, idx->dp, oI, oD, oP,
                   idx->ctr, idx->dbg, (nq <= (1u << 16)) ? idx->d_tstamp : nullptr, idx->curDynamic, idx->curZero8,
                   (const uint4*)idx->d_codesGrp, (uint64_t)idx->nIds, idx->d_bias, kappa, 20.f * idx->coarseMax, idx->d_fbList, idx->d_fbCount,
                   idx->d_fbList, idx->d_fbCount, idx->curRuns ? idx->d_runs : nullptr, idx->d_runGpos, idx->d_nRuns, idx->curRunCap, idx->curPool, idx->curPoolNext, idx->curPool ? idx->curPool + 16 : nullptr, idx->d_schedList, idx->curSchedCap};
}
template <int NW, int LPV, bool SH, int MODE>
int launchRSBias(pqt_index* idx, uint32_t grid, size_t lds, hipStream_t st, const float* qL1virt, const uint32_t* nLocal,
                 uint64_t stride, uint32_t k, uint32_t nq, uint32_t* oI, float* oD, uint32_t* oP) {
  constexpr int UV = LPV >= 8 ? 2 : 4;
  const uint32_t c1 = idx->dp.C1;
#ifdef PQT_DEV_CFG3_ONLY
  if (c1 != 64) return pqtFail(PQT_ERR_LIMIT, "development build: C1 = 64 only");
  auto kern = pqt_k_rerank_select<NW, LPV, UV, false, SH, 6, MODE>;
#else
  auto kern = c1 == 64 ? pqt_k_rerank_select<NW, LPV, UV, false, SH, 6, MODE> : c1 == 32 ? pqt_k_rerank_select<NW, LPV, UV, false, SH, 5, MODE>
                                                                                         : pqt_k_rerank_select<NW, LPV, UV, false, SH, 1, MODE>;
#endif
  constexpr bool kRunsVariant = NW == 12 && LPV == 8;  // bin runs: BASELINE configs[2]/[3] shape only
  if constexpr (kRunsVariant) { if (idx->curRuns) kern = pqt_k_rerank_select<NW, LPV, UV, false, SH, 6, MODE, true>; }
  int rc = allowLds(kern, lds);
  if (rc) return rc;
  const PqtRsArgs rargs = rsArgsFilter(idx, qL1virt, nLocal, stride, k, nq, oI, oD, oP);
  if (MODE == 2) HIPCHK(hipMemsetAsync(idx->d_fbCount, 0, 4, st));
  hipExtLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), (uint32_t)lds, st, idx->lev0, idx->lev1, 0u, rargs);
  if (MODE == 2 && kRunsVariant && c1 == 64 && idx->candCap) {
    // queries whose near-tie band overflowed the wave's list (normally none, a few per FRESH batch at configs[2]): exact distances by whole
    // workgroups, then the exact selection over them (pqt_shared_launch.hip) -- not one wavefront per 23 k-candidate query
    if ((rc = launchHandedBack(idx, st, rargs))) return rc;
  } else
  if (MODE == 2) {
    // queries whose near-tie band overflowed the wave's list (normally none): plain exact kernel on that list
    constexpr int LNW = 4;
    auto lk = pqt_k_rerank_select_list<LNW, LPV, UV, SH, 1>;
    size_t llds = (size_t)LNW * ((PQT_RS_BEST + PQT_RS_PEND) * 8 + (size_t)idx->dp.LP * idx->dp.C1 * 4);
    if constexpr (kRunsVariant) { if (idx->curRuns) { lk = pqt_k_rerank_select_list<LNW, LPV, UV, SH, 1, true>; llds = ((llds + 15) & ~(size_t)15) + (size_t)LNW * idx->curRunCap * 12; } }
    if ((rc = allowLds(lk, llds))) return rc;
    PqtRsArgs largs = rargs;
    largs.tstamp = nullptr; largs.dynamic = 0; largs.zero8 = nullptr; largs.pool = nullptr; largs.poolNext = nullptr; largs.schedCnt = nullptr;
    hipLaunchKernelGGL(lk, dim3(std::min<uint32_t>((nq + LNW - 1) / LNW, (uint32_t)idx->numCUs * 2)), dim3(LNW * 64), llds, st, largs);
  }
  return PQT_OK;
}

// ---- workgroup-per-query rerank+select for coarse tables that do not fit LDS (pqt_k_rerank_select_wg) ----------
template <int G>
int launchRSWG(pqt_index* idx, uint32_t nq, hipStream_t st, const float* v, const uint32_t* nl, uint64_t stride, uint32_t k,
               uint32_t* oI, float* oD, uint32_t* oP) {
  const PqtDevParams& d = idx->dp;
  int rc0 = ensureGroupMajor(idx, G);
  if (rc0) return rc0;
  const size_t lds = (size_t)G * d.C1 * d.C1 * 4 + (size_t)d.LP * d.C1 * 4 + (size_t)PQT_RS2_NW * PQT_RS2_KEYS * 8;
  const bool p2 = (d.C1 & (d.C1 - 1)) == 0;
  const uint32_t c1v = idx->dp.C1;
  auto kern = idx->sharded ? (c1v == 64 ? pqt_k_rerank_select_wg<G, true, 6> : c1v == 128 ? pqt_k_rerank_select_wg<G, true, 7>
                              : p2 ? pqt_k_rerank_select_wg<G, true, 1> : pqt_k_rerank_select_wg<G, true, 0>)
                           : (c1v == 64 ? pqt_k_rerank_select_wg<G, false, 6> : c1v == 128 ? pqt_k_rerank_select_wg<G, false, 7>
                              : p2 ? pqt_k_rerank_select_wg<G, false, 1> : pqt_k_rerank_select_wg<G, false, 0>);
  int rc = allowLds(kern, lds);
  if (rc) return rc;
  hipExtLaunchKernelGGL(kern, dim3(nq), dim3(PQT_RS2_NW * 64), (uint32_t)lds, st, idx->lev0, idx->lev1, 0u, idx->d_codesGrp, (uint64_t)idx->nIds, idx->d_ids, v,
                        idx->d_coarse, idx->d_cand, idx->d_candPos, nl, stride, k, d, oI, oD, oP, idx->ctr, idx->dbg);
  return PQT_OK;
}
#ifndef PQT_RSWG_SLICE_KB
#define PQT_RSWG_SLICE_KB 32   // 32 KB slices let two workgroups share a CU (one stages while the other computes)
#endif
constexpr size_t kRswgSlice = (size_t)PQT_RSWG_SLICE_KB * 1024;
}  // namespace
// line parts per staged group: the largest of 4, 2, 1 whose table slice fits the budget and divides LP; falls back to a
// 64 KB slice (C1 = 128); 0 = unsupported
int rswgGroup(const PqtDevParams& d) {
  for (int g : {4, 2, 1})
    if ((size_t)g * d.C1 * d.C1 * 4 <= kRswgSlice && d.LP % g == 0 && (d.C1 * d.C1) % 4 == 0) return g;
  for (int g : {4, 2, 1})
    if ((size_t)g * d.C1 * d.C1 * 4 <= 64 * 1024 && d.LP % g == 0 && (d.C1 * d.C1) % 4 == 0) return g;
  return 0;
}


// ---- entry points with run-time dispatch ------------------------------------------------------------------------------------
int launchRSBiasAny(pqt_index* idx, int nw, bool filter, uint32_t grid, size_t lds, hipStream_t st, const float* v, const uint32_t* nl,
                    uint64_t stride, uint32_t k, uint32_t nq, uint32_t* oI, float* oD, uint32_t* oP) {
#define PQT_LAUNCH_BIAS1(NWV, LPVV, MD) (idx->sharded ? launchRSBias<NWV, LPVV, true, MD>(idx, grid, lds, st, v, nl, stride, k, nq, oI, oD, oP) \
                                                      : launchRSBias<NWV, LPVV, false, MD>(idx, grid, lds, st, v, nl, stride, k, nq, oI, oD, oP))
#define PQT_LAUNCH_BIAS(NWV, LPVV) (filter ? PQT_LAUNCH_BIAS1(NWV, LPVV, 2) : PQT_LAUNCH_BIAS1(NWV, LPVV, 1))
#ifdef PQT_DEV_CFG3_ONLY
  if (!(idx->dp.LP == 32 && nw == 12 && filter)) return pqtFail(PQT_ERR_LIMIT, "development build: LP = 32, 12 wavefronts, exact filter only");
  return PQT_LAUNCH_BIAS1(12, 8, 2);
#else
  return idx->dp.LP == 16 ? (nw == 12 ? PQT_LAUNCH_BIAS(12, 4) : PQT_LAUNCH_BIAS(6, 4)) : (nw == 12 ? PQT_LAUNCH_BIAS(12, 8) : PQT_LAUNCH_BIAS(6, 8));
#endif
#undef PQT_LAUNCH_BIAS1
#undef PQT_LAUNCH_BIAS
}

int launchRSWGAny(pqt_index* idx, int G, uint32_t nq, hipStream_t st, const float* v, const uint32_t* nl, uint64_t stride, uint32_t k,
                  uint32_t* oI, float* oD, uint32_t* oP) {
#ifdef PQT_DEV_CFG3_ONLY
  return pqtFail(PQT_ERR_LIMIT, "development build");
#else
  return G == 4 ? launchRSWG<4>(idx, nq, st, v, nl, stride, k, oI, oD, oP)
       : G == 2 ? launchRSWG<2>(idx, nq, st, v, nl, stride, k, oI, oD, oP)
                : launchRSWG<1>(idx, nq, st, v, nl, stride, k, oI, oD, oP);
#endif
}

int launchSmallLists(pqt_index* idx, bool cl, size_t lds, uint32_t grid, hipStream_t st, const PqtRsArgs& sa, hipEvent_t ev0) {
#ifdef PQT_DEV_CFG3_ONLY
  return pqtFail(PQT_ERR_LIMIT, "development build");
#else
  constexpr int SNW = kSmallWaves;
  int rc;
#define PQT_LAUNCH_SMALL(LPVV, CL)                                                                                            \
  do { auto kern = pqt_k_rerank_sort_small<SNW, LPVV, CL>;                                                                     \
       if ((rc = allowLds(kern, lds))) return rc;                                                                              \
       hipExtLaunchKernelGGL(kern, dim3(grid), dim3(SNW * 64), (uint32_t)lds, st, ev0, nullptr, 0u, sa, idx->d_fbList, idx->d_fbCount); } while (0)
  if (idx->dp.LP == 16) { if (cl) PQT_LAUNCH_SMALL(4, true); else PQT_LAUNCH_SMALL(4, false); }
  else { if (cl) PQT_LAUNCH_SMALL(8, true); else PQT_LAUNCH_SMALL(8, false); }
#undef PQT_LAUNCH_SMALL
  return PQT_OK;
#endif
}

// second pass of the short-list path: lists of 1025..2048 candidates (SIFT1M shape with the coarse table in LDS only)
int launchMidLists(pqt_index* idx, size_t lds, uint32_t grid, hipStream_t st, const PqtRsArgs& sa, uint32_t* outList, uint32_t* outCount) {
#ifdef PQT_DEV_CFG3_ONLY
  return pqtFail(PQT_ERR_LIMIT, "development build");
#else
  auto kern = pqt_k_rerank_sort_small<kMidWaves, 4, true, 2048, true>;
  int rc = allowLds(kern, lds);
  if (rc) return rc;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kMidWaves * 64), (uint32_t)lds, st, sa, outList, outCount);
  return PQT_OK;
#endif
}

int launchBigK(pqt_index* idx, bool cl, size_t lBig, uint32_t nq, hipStream_t st, const float* v, const uint32_t* nl, uint64_t stride, uint32_t k,
               uint32_t kP2, uint32_t kcap, uint32_t* oI, float* oD, uint32_t* oP, const uint32_t* qlist, const uint32_t* qcount, hipEvent_t ev0, hipEvent_t ev1) {
#ifdef PQT_DEV_CFG3_ONLY
  return pqtFail(PQT_ERR_LIMIT, "development build");
#else
  const PqtDevParams& d = idx->dp;
  int rc;
#define PQT_LAUNCH_BIG(CL, SH, VEC)                                                                                         \
  do { auto kern = pqt_k_rerank_select_big<CL, SH, VEC>;                                                                     \
       if ((rc = allowLds(kern, lBig))) return rc;                                                                           \
       const uint32_t wgPerCu = (uint32_t)std::max<size_t>(1, std::min<size_t>(4, kMaxLds / lBig));                          \
       hipExtLaunchKernelGGL(kern, dim3(std::min<uint32_t>(nq, (uint32_t)idx->numCUs * wgPerCu)), dim3(PQT_RSB_NT), (uint32_t)lBig, st, ev0, ev1, 0u, \
                             idx->d_codesBin, idx->d_ids, v, idx->d_coarse, idx->d_cand, idx->d_candPos, nl, stride, k, kP2, kcap, nq, d, \
                             oI, oD, oP, idx->ctr, qlist, qcount); } while (0)
  if (d.LP % 4 == 0) {
    if (cl) { if (idx->sharded) PQT_LAUNCH_BIG(true, true, 4); else PQT_LAUNCH_BIG(true, false, 4); }
    else { if (idx->sharded) PQT_LAUNCH_BIG(false, true, 4); else PQT_LAUNCH_BIG(false, false, 4); }
  } else {
    if (cl) { if (idx->sharded) PQT_LAUNCH_BIG(true, true, 1); else PQT_LAUNCH_BIG(true, false, 1); }
    else { if (idx->sharded) PQT_LAUNCH_BIG(false, true, 1); else PQT_LAUNCH_BIG(false, false, 1); }
  }
#undef PQT_LAUNCH_BIG
  return PQT_OK;
#endif
}

#endif  // PQT_DEV_SIFT1M_ONLY
