// pqt_shared_launch.hip -- the shared-row pass of the filtered rerank (pqt_shared_rows.h): scratch, launches, and the instantiations of the
// selection kernel that reads its distances (pqt_k_rerank_select<.., PRE = true>).  A translation unit of its own (build time).
#include "pqt_internal.h"
#include "pqt_shared_rows.h"

namespace {
constexpr int kSrWaves = 16;   // wavefronts per workgroup of pqt_k_sr_adc: 16 x 8 KB of L1virt copies
constexpr int kSrU = 2;        // rows per lane in flight
template <class T>
int growArr(T** p, uint64_t* cap, uint64_t need) {
  if (need <= *cap) return PQT_OK;
  int rc = devAlloc(p, (size_t)need);
  if (rc) { *cap = 0; return rc; }
  *cap = need;
  return PQT_OK;
}
}  // namespace

bool sharedRowsShape(const pqt_index* idx) { return idx->dp.LP == 32 && idx->dp.C1 == 64; }

// steps 1-4 of pqt_shared_rows.h for the chunk whose traversal has just been enqueued on st: fills d_candDist and d_srPreOk
int launchSharedRows(pqt_index* idx, hipStream_t st, const float* qL1virt, const uint32_t* nLocal, uint64_t stride, uint32_t nq, hipEvent_t ev0) {
  if (!sharedRowsShape(idx)) return pqtFail(PQT_ERR_LIMIT, "shared-row pass: 32 line parts, C1 = 64 only");
  int rc;
  uint32_t bits = 12;
  while ((1ull << bits) < (uint64_t)nq * 16 && bits < 24) ++bits;
  const uint64_t slots = 1ull << bits;
  const uint64_t itemCap = (uint64_t)nq * (stride / PQT_SR_TILE + 64);
  if ((rc = growArr(&idx->d_srTable, &idx->srTableCap, slots * 4))) return rc;     // keys | cnt | len | base
  if ((rc = growArr(&idx->d_srPairs, &idx->srPairCap, (uint64_t)nq * 64 * 2 + nq))) return rc;  // pairSlot | pairIdx | preOk
  if ((rc = growArr(&idx->d_srBlocks, &idx->srBlockCap, slots / 1024 + 16))) return rc;
  if ((rc = growArr(&idx->d_srItems, &idx->srItemCap, itemCap))) return rc;
  PqtSrArgs a{};
  a.runs = idx->d_runs; a.nRuns = idx->d_nRuns; a.nLocal = nLocal; a.qn = nq;
  a.keys = idx->d_srTable; a.cnt = a.keys + slots; a.len = a.cnt + slots; a.base = a.len + slots; a.slotBits = bits;
  a.pairSlot = idx->d_srPairs; a.pairIdx = a.pairSlot + (size_t)nq * 64; a.preOk = a.pairIdx + (size_t)nq * 64;
  a.blockSum = idx->d_srBlocks; a.nBlocks = (uint32_t)(slots / 1024); a.total = idx->d_srBlocks + a.nBlocks;
  a.items = idx->d_srItems; a.itemCap = itemCap;
  a.codesGrp4 = (const uint4*)idx->d_codesGrp; a.nIds = idx->nIds; a.bias = idx->d_bias; a.qL1virt = qL1virt; a.dist = idx->d_candDist; a.stride = stride;
  idx->curPreOk = a.preOk;
  HIPCHK(hipMemsetAsync(a.keys, 0xff, slots * 4, st));
  HIPCHK(hipMemsetAsync(a.cnt, 0, slots * 4, st));
  hipExtLaunchKernelGGL(pqt_k_sr_visits, dim3((nq + 3) / 4), dim3(256), 0, st, ev0, nullptr, 0u, a);
  hipLaunchKernelGGL(pqt_k_sr_scan, dim3(a.nBlocks), dim3(1024), 0, st, a);
  hipLaunchKernelGGL(pqt_k_sr_scan2, dim3(1), dim3(1024), 0, st, a);
  hipLaunchKernelGGL(pqt_k_sr_items, dim3((uint32_t)(((uint64_t)nq * 64 + 255) / 256)), dim3(256), 0, st, a);
  auto kern = pqt_k_sr_adc<kSrWaves, 8, 6, kSrU>;
  const size_t lds = (size_t)kSrWaves * idx->dp.LP * idx->dp.C1 * 4;
  if ((rc = allowLds(kern, lds))) return rc;
  hipLaunchKernelGGL(kern, dim3((uint32_t)idx->numCUs), dim3(kSrWaves * 64), lds, st, a);
  return PQT_OK;
}

// step 5: the 12-wavefront filtered selection with bin runs (launchRSBias<12, 8, SH, 2> of pqt_rerank_launch.hip) reading the pass's distances
template <bool SH>
static int launchSel(pqt_index* idx, uint32_t grid, size_t lds, hipStream_t st, const float* qL1virt, const uint32_t* nLocal,
                     uint64_t stride, uint32_t k, uint32_t nq, uint32_t* oI, float* oD, uint32_t* oP) {
  constexpr int NW = 12, LPV = 8, UV = 2;
  auto kern = pqt_k_rerank_select<NW, LPV, UV, false, SH, 6, 2, true, false, PQT_RS_BEST + PQT_RS_PEND, true>;
  int rc = allowLds(kern, lds);
  if (rc) return rc;
  const double lp = idx->dp.LP;
  const float kappa = (float)(2.02 * (lp * lp + 8.0 * lp + 2.0) / 16777216.0);
  PqtRsArgs rargs{idx->d_codesBin, idx->d_ids, qL1virt, idx->d_coarse, idx->d_cand, idx->d_candPos, nLocal, stride, k, nq, idx->dp, oI, oD, oP,
                  idx->ctr, idx->dbg, (nq <= (1u << 16)) ? idx->d_tstamp : nullptr, idx->curDynamic, idx->curZero8,
                  (const uint4*)idx->d_codesGrp, (uint64_t)idx->nIds, idx->d_bias, kappa, 20.f * idx->coarseMax, idx->d_fbList, idx->d_fbCount,
                  idx->d_fbList, idx->d_fbCount, idx->d_runs, idx->d_runGpos, idx->d_nRuns, idx->curRunCap, idx->curPool, idx->curPoolNext, idx->curPool ? idx->curPool + 16 : nullptr, idx->d_schedList, idx->curSchedCap};
  rargs.preDist = idx->d_candDist; rargs.preOk = idx->curPreOk;
  HIPCHK(hipMemsetAsync(idx->d_fbCount, 0, 4, st));
  hipExtLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), (uint32_t)lds, st, nullptr, idx->lev1, 0u, rargs);
  // queries whose near-tie band overflowed the wave's list (normally none): plain exact kernel on that list
  constexpr int LNW = 4;
  auto lk = pqt_k_rerank_select_list<LNW, LPV, UV, SH, 1, true>;
  const size_t llds = ((((size_t)LNW * ((PQT_RS_BEST + PQT_RS_PEND) * 8 + (size_t)idx->dp.LP * idx->dp.C1 * 4)) + 15) & ~(size_t)15) + (size_t)LNW * idx->curRunCap * 12;
  if ((rc = allowLds(lk, llds))) return rc;
  PqtRsArgs largs = rargs;
  largs.tstamp = nullptr; largs.dynamic = 0; largs.zero8 = nullptr; largs.pool = nullptr; largs.poolNext = nullptr; largs.schedCnt = nullptr;
  hipLaunchKernelGGL(lk, dim3(std::min<uint32_t>((nq + LNW - 1) / LNW, (uint32_t)idx->numCUs * 2)), dim3(LNW * 64), llds, st, largs);
  return PQT_OK;
}
int launchSharedSelect(pqt_index* idx, uint32_t grid, size_t lds, hipStream_t st, const float* v, const uint32_t* nl,
                       uint64_t stride, uint32_t k, uint32_t nq, uint32_t* oI, float* oD, uint32_t* oP) {
  return idx->sharded ? launchSel<true>(idx, grid, lds, st, v, nl, stride, k, nq, oI, oD, oP) : launchSel<false>(idx, grid, lds, st, v, nl, stride, k, nq, oI, oD, oP);
}
