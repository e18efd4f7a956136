// pqt_shared_launch.hip -- the shared-row pass of the filtered rerank (pqt_shared_rows.h): scratch, launches, and the instantiations of the
// selection kernel that reads its distances (pqt_k_rerank_select<.., PRE = true>).  A translation unit of its own (build time).
#include "pqt_internal.h"
#include "pqt_shared_rows.h"

namespace {
#ifndef PQT_SR_WAVES
#define PQT_SR_WAVES 8
#endif
#ifndef PQT_SR_WGS
#define PQT_SR_WGS 2
#endif
constexpr int kSrWaves = PQT_SR_WAVES;   // wavefronts per workgroup of pqt_k_sr_adc (around PQT_SR_QC tables of 8 KB)
constexpr int kSrWgs = PQT_SR_WGS;       // workgroups per CU
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
  if (idx->srSlotBits) bits = idx->srSlotBits;  // tests: a table small enough to fill up
  const uint64_t slots = 1ull << bits;
  const uint64_t itemCap = (uint64_t)nq * (stride / PQT_SR_TILE + 64), listCap = (uint64_t)nq * 64;
  if ((rc = growArr(&idx->d_srTable, &idx->srTableCap, slots * 5))) return rc;     // keys | cnt | len | base | lbase
  if ((rc = growArr(&idx->d_srPairs, &idx->srPairCap, (uint64_t)nq * 64 * 2 + 2 * (uint64_t)nq))) return rc;  // pairSlot | pairIdx | preOk | qmax
  if ((rc = growArr(&idx->d_srBlocks, &idx->srBlockCap, 2 * (slots / 1024) + 16 + 16))) return rc;  // block sums | total[16] | 8 statistics counters (64-bit)
  if ((rc = growArr(&idx->d_srItems, &idx->srItemCap, itemCap + listCap))) return rc;
  PqtSrArgs a{};
  a.runs = idx->d_runs; a.nRuns = idx->d_nRuns; a.nLocal = nLocal; a.qn = nq;
  a.keys = idx->d_srTable; a.cnt = a.keys + slots; a.len = a.cnt + slots; a.base = a.len + slots; a.lbase = a.base + slots; a.slotBits = bits;
  a.pairSlot = idx->d_srPairs; a.pairIdx = a.pairSlot + (size_t)nq * 64; a.preOk = a.pairIdx + (size_t)nq * 64; a.qmax = reinterpret_cast<float*>(a.preOk + nq);
  a.blockSum = idx->d_srBlocks; a.nBlocks = (uint32_t)(slots / 1024); a.total = idx->d_srBlocks + 2 * a.nBlocks;
  a.maxProbes = idx->srProbes; a.stat = reinterpret_cast<unsigned long long*>(idx->d_srBlocks + ((2 * (size_t)a.nBlocks + 16 + 1) & ~(size_t)1));  // (8-byte aligned)
  a.items = idx->d_srItems; a.itemCap = itemCap; a.binList = idx->d_srItems + itemCap; a.listCap = listCap;
  a.codesGrp4 = (const uint4*)idx->d_codesGrp; a.nIds = idx->nIds; a.bias = idx->d_bias; a.qL1virt = qL1virt; a.dist = idx->d_candDist; a.stride = stride; a.tableFloats = idx->dp.LP * idx->dp.C1;
  idx->curPreOk = a.preOk; idx->curPreQmax = a.qmax; idx->curPreFlags = a.total;
  HIPCHK(hipMemsetAsync(a.keys, 0xff, slots * 4, st));
  HIPCHK(hipMemsetAsync(a.cnt, 0, slots * 4, st));
  hipExtLaunchKernelGGL(pqt_k_sr_visits, dim3((nq + 3) / 4), dim3(256), 0, st, ev0, nullptr, 0u, a);
  hipLaunchKernelGGL(pqt_k_sr_scan, dim3(a.nBlocks), dim3(1024), 0, st, a);
  hipLaunchKernelGGL(pqt_k_sr_scan2, dim3(1), dim3(1024), 0, st, a);
  hipLaunchKernelGGL(pqt_k_sr_items, dim3((uint32_t)(((uint64_t)nq * 64 + 255) / 256)), dim3(256), 0, st, a);
  if (idx->srStats) {
    // on request: what the pass will read and write for this batch (pqt_get_shared_rows_stats after the call)
    HIPCHK(hipMemsetAsync(a.stat, 0, 8 * sizeof(unsigned long long), st));
    const uint32_t sg = (uint32_t)((std::max<uint64_t>(slots, nq) + 1023) / 1024);
    hipLaunchKernelGGL(pqt_k_sr_stats, dim3(sg), dim3(1024), 0, st, a);
    idx->srStatPtr = a.stat;
  }
  const size_t lds = (size_t)PQT_SR_QC * (idx->dp.LP * idx->dp.C1 * 4 + 64 * 4 + 8);
  // (timed calls: the stop event of stage "rerank_select" = preparation + this kernel rides on its dispatch; the selection is stage "select")
  if (idx->srKernel == 2) {
    // pair-interleaved tables, row decode outside the query loop (pqt_k_sr_adc2; same LDS bytes, same results bit for bit)
    static_assert(PQT_SR_QC == 8 && kSrWaves == 8, "pqt_k_sr_adc2 is written for chunks of 8 queries and 8 wavefronts");
    auto kern = pqt_k_sr_adc2<kSrWaves, 8, 6>;
    if ((rc = allowLds(kern, lds))) return rc;
    hipExtLaunchKernelGGL(kern, dim3((uint32_t)idx->numCUs * kSrWgs), dim3(kSrWaves * 64), (uint32_t)lds, st, nullptr, idx->lev1, 0u, a);
    return PQT_OK;
  }
  auto kern = pqt_k_sr_adc<kSrWaves, 8, 6>;
  if ((rc = allowLds(kern, lds))) return rc;
  hipExtLaunchKernelGGL(kern, dim3((uint32_t)idx->numCUs * kSrWgs), dim3(kSrWaves * 64), (uint32_t)lds, st, nullptr, idx->lev1, 0u, a);
  return PQT_OK;
}

// The queries a filtered selection handed back through fbList (near-tie band beyond its 256 slots; in the shared-row pass also the queries
// the pass did not cover): exact distances of all their candidates by whole workgroups (pqt_k_sr_exact_list, 2048 candidates per
// workgroup), then the exact (MODE 0) selection over them, one wavefront per query.  Replaces the one-wavefront-per-query exact list
// kernel at the configs[2]/[3] shape: that wavefront needed ~1 ms for a 23 k-candidate query, once per (fresh) batch -- most of the
// 0.75 ms per step VERDICT r04 found outside the two kernels.  rargs: the arguments of the selection launch (fbList / fbCount filled by it).
template <bool SH>
static int handedBack(pqt_index* idx, hipStream_t st, const PqtRsArgs& rargs) {
  constexpr int XNW = 16, LNW = 4, LPV = 8, UV = 2;
  int rc;
  auto xk = pqt_k_sr_exact_list<XNW, LPV, 6>;
  const size_t xlds = (size_t)idx->dp.LP * idx->dp.C1 * 4 + (size_t)PQT_RUNCAP * 8;
  if ((rc = allowLds(xk, xlds))) return rc;
  PqtRsArgs largs = rargs;
  largs.tstamp = nullptr; largs.dynamic = 0; largs.zero8 = nullptr; largs.pool = nullptr; largs.poolNext = nullptr; largs.schedCnt = nullptr;
  largs.qlist = idx->d_fbList; largs.qcount = idx->d_fbCount; largs.preDist = idx->d_candDist;
  largs.codes = idx->d_codesBin;
  if (!idx->curRuns) { largs.runs = nullptr; largs.nRuns = nullptr; }
  hipLaunchKernelGGL(xk, dim3((uint32_t)idx->numCUs), dim3(XNW * 64), xlds, st, largs);
  auto lk = pqt_k_sr_select<LNW, LPV, UV, SH, 6, true>;
  const size_t llds = (size_t)LNW * ((PQT_RS_BEST + PQT_RS_PEND) * 8 + (size_t)PQT_RUNCAP * 12);
  if ((rc = allowLds(lk, llds))) return rc;
  hipLaunchKernelGGL(lk, dim3(std::min<uint32_t>((rargs.qn + LNW - 1) / LNW, (uint32_t)idx->numCUs)), dim3(LNW * 64), llds, st, largs);
  return PQT_OK;
}
int launchHandedBack(pqt_index* idx, hipStream_t st, const PqtRsArgs& rargs) {
  if (!sharedRowsShape(idx)) return pqtFail(PQT_ERR_LIMIT, "handed-back queries: 32 line parts, C1 = 64 only");
  return idx->sharded ? handedBack<true>(idx, st, rargs) : handedBack<false>(idx, st, rargs);
}

// step 5: the selection over the pass's distances (pqt_k_sr_select), then the plain exact kernel for the queries it handed back
#ifndef PQT_SR_SEL_WAVES
#define PQT_SR_SEL_WAVES 16
#endif
#ifndef PQT_SR_SEL_WGS
#define PQT_SR_SEL_WGS 2
#endif
#ifndef PQT_SR_SEL_SPLIT
#define PQT_SR_SEL_SPLIT 1
#endif
template <bool SH>
static int launchSel(pqt_index* idx, hipStream_t st, const float* qL1virt, const uint32_t* nLocal,
                     uint64_t stride, uint32_t k, uint32_t nq, uint32_t* oI, float* oD, uint32_t* oP) {
  constexpr int NW = PQT_SR_SEL_WAVES, LPV = 8, UV = 2;
  const double lp = idx->dp.LP;
  const float kappa = (float)(2.02 * (lp * lp + 8.0 * lp + 2.0) / 16777216.0);
  PqtRsArgs rargs{idx->d_codesBin, idx->d_ids, qL1virt, idx->d_coarse, idx->d_cand, idx->d_candPos, nLocal, stride, k, nq, idx->dp, oI, oD, oP,
                  idx->ctr, idx->dbg, (nq <= (1u << 16)) ? idx->d_tstamp : nullptr, 0u, idx->curZero8,
                  (const uint4*)idx->d_codesGrp, (uint64_t)idx->nIds, idx->d_bias, kappa, 20.f * idx->coarseMax, idx->d_fbList, idx->d_fbCount,
                  idx->d_fbList, idx->d_fbCount, idx->d_runs, idx->d_runGpos, idx->d_nRuns, idx->curRunCap, idx->curPool, idx->curPoolNext, idx->curPool ? idx->curPool + 16 : nullptr, idx->d_schedList, idx->curSchedCap};
  rargs.preDist = idx->d_candDist; rargs.preOk = idx->curPreOk; rargs.preQmax = idx->curPreQmax; rargs.preFlags = idx->curPreFlags;
  HIPCHK(hipMemsetAsync(idx->d_fbCount, 0, 4, st));
  int rc;
#if PQT_SR_SEL_SPLIT
  // two launches: the scan over the distances (keys and run list in LDS only: more wavefronts per CU) and the band re-evaluation + results
  // (table copy in LDS) -- the best lists travel through global memory (2 KB per query)
  if ((rc = growArr(&idx->d_srKeys, &idx->srKeysCap, (uint64_t)nq * 257))) return rc;
  rargs.preKeys = idx->d_srKeys; rargs.preCnt = reinterpret_cast<uint32_t*>(idx->d_srKeys + (size_t)nq * 256);
  if (idx->srScanSplit > 1 || idx->srScanDepth != 4) {
    // opt-in (round 6): the scan over position ranges of a query with a deeper request queue, then the merge of a query's range lists
    const uint32_t seg = (uint32_t)idx->srScanSplit;
    if ((rc = growArr(&idx->d_srSeg, &idx->srSegCap, (uint64_t)nq * seg * 257))) return rc;
    unsigned long long* const segKeys = idx->d_srSeg;
    uint32_t* const segCnt = reinterpret_cast<uint32_t*>(idx->d_srSeg + (size_t)nq * seg * 256);
    const size_t lds = (size_t)NW * ((size_t)1024 * 8);
    const uint32_t items = nq * seg;
    const uint32_t grid = std::min<uint32_t>((items + NW - 1) / NW, (uint32_t)idx->numCUs * PQT_SR_SEL_WGS);
#define PQT_SCAN_SEG(SEGV, QDV)                                                                                     \
    do { auto kern = pqt_k_sr_scan_seg<NW, SEGV, QDV>;                                                              \
         if ((rc = allowLds(kern, lds))) return rc;                                                                 \
         hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, st, rargs, segKeys, segCnt); } while (0)
    const bool deep = idx->srScanDepth == 8;
    if (seg == 1) { if (deep) PQT_SCAN_SEG(1, 8); else PQT_SCAN_SEG(1, 4); }
    else if (seg == 2) { if (deep) PQT_SCAN_SEG(2, 8); else PQT_SCAN_SEG(2, 4); }
    else { if (deep) PQT_SCAN_SEG(4, 8); else PQT_SCAN_SEG(4, 4); }
#undef PQT_SCAN_SEG
    constexpr int MW = 4;
    const uint32_t mgrid = (nq + MW - 1) / MW;
    if (seg == 1) hipLaunchKernelGGL((pqt_k_sr_merge<MW, 1>), dim3(mgrid), dim3(MW * 64), 0, st, rargs, (const unsigned long long*)segKeys, (const uint32_t*)segCnt);
    else if (seg == 2) hipLaunchKernelGGL((pqt_k_sr_merge<MW, 2>), dim3(mgrid), dim3(MW * 64), 0, st, rargs, (const unsigned long long*)segKeys, (const uint32_t*)segCnt);
    else hipLaunchKernelGGL((pqt_k_sr_merge<MW, 4>), dim3(mgrid), dim3(MW * 64), 0, st, rargs, (const unsigned long long*)segKeys, (const uint32_t*)segCnt);
  } else {
    auto kern = pqt_k_sr_select<NW, LPV, UV, SH, 6, false, 2>;
    const size_t lds = (size_t)NW * ((size_t)1024 * 8);  // the lean scan: 1024 key slots per wavefront, nothing else
    if ((rc = allowLds(kern, lds))) return rc;
    const uint32_t grid = std::min<uint32_t>((nq + NW - 1) / NW, (uint32_t)idx->numCUs * PQT_SR_SEL_WGS);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, st, rargs);
  }
  {
    constexpr int BW = 12;
    auto kern = pqt_k_sr_select<BW, LPV, UV, SH, 6, false, 3>;
    const size_t lds = (size_t)BW * ((PQT_RS_BEST + PQT_RS_PEND) * 8 + (size_t)idx->curRunCap * 12 + (size_t)idx->dp.LP * idx->dp.C1 * 4);
    if ((rc = allowLds(kern, lds))) return rc;
    const uint32_t grid = std::min<uint32_t>((nq + BW - 1) / BW, (uint32_t)idx->numCUs);
    hipExtLaunchKernelGGL(kern, dim3(grid), dim3(BW * 64), (uint32_t)lds, st, nullptr, idx->lev1, 0u, rargs);
  }
#else
  {
    auto kern = pqt_k_sr_select<NW, LPV, UV, SH, 6, false>;
    const size_t lds = (size_t)NW * ((PQT_RS_BEST + PQT_RS_PEND) * 8 + (size_t)idx->curRunCap * 12);
    if ((rc = allowLds(kern, lds))) return rc;
    const uint32_t grid = std::min<uint32_t>((nq + NW - 1) / NW, (uint32_t)idx->numCUs * PQT_SR_SEL_WGS);
    hipExtLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), (uint32_t)lds, st, nullptr, idx->lev1, 0u, rargs);
  }
#endif
  return launchHandedBack(idx, st, rargs);
}
// Cooperative filter scan (pqt_k_pair_scan: two wavefronts per query around one table copy) + merge of the pair's lists + the band launch
// + the handed-back queries: the filtered rerank of one chunk in four launches.  Opt-in ("coop_rerank" = 1).
template <bool SH>
static int launchCoop(pqt_index* idx, hipStream_t st, const float* qL1virt, const uint32_t* nLocal, uint64_t stride, uint32_t k, uint32_t nq,
                      uint32_t* oI, float* oD, uint32_t* oP) {
  constexpr int LPV = 8, UV = 2;
  const double lp = idx->dp.LP;
  const float kappa = (float)(2.02 * (lp * lp + 8.0 * lp + 2.0) / 16777216.0);
  PqtRsArgs rargs{idx->d_codesBin, idx->d_ids, qL1virt, idx->d_coarse, idx->d_cand, idx->d_candPos, nLocal, stride, k, nq, idx->dp, oI, oD, oP,
                  idx->ctr, idx->dbg, nullptr, 0u, idx->curZero8,
                  (const uint4*)idx->d_codesGrp, (uint64_t)idx->nIds, idx->d_bias, kappa, 20.f * idx->coarseMax, idx->d_fbList, idx->d_fbCount,
                  idx->d_fbList, idx->d_fbCount, idx->curRuns ? idx->d_runs : nullptr, idx->d_runGpos, idx->d_nRuns, idx->curRunCap, idx->curPool, idx->curPoolNext, idx->curPool ? idx->curPool + 16 : nullptr, idx->d_schedList, idx->curSchedCap};
  int rc;
  if ((rc = growArr(&idx->d_srSeg, &idx->srSegCap, (uint64_t)nq * 2 * 257))) return rc;   // the two lists of every query + their lengths
  if ((rc = growArr(&idx->d_srKeys, &idx->srKeysCap, (uint64_t)nq * 257))) return rc;     // the merged list + its length
  if ((rc = growArr(&idx->d_srPairs, &idx->srPairCap, (uint64_t)nq * 2))) return rc;      // preOk (all ones) | largest table entry per query
  if (!idx->d_coopErr) { if ((rc = devAlloc(&idx->d_coopErr, 1))) return rc; HIPCHK(hipMemsetAsync(idx->d_coopErr, 0, 4, st)); }
  unsigned long long* const segKeys = idx->d_srSeg;
  uint32_t* const segCnt = reinterpret_cast<uint32_t*>(idx->d_srSeg + (size_t)nq * 2 * 256);
  uint32_t* const ones = idx->d_srPairs;
  float* const qmax = reinterpret_cast<float*>(idx->d_srPairs + nq);
  HIPCHK(hipMemsetAsync(ones, 0x01, (size_t)nq * 4, st));
  HIPCHK(hipMemsetAsync(idx->d_fbCount, 0, 4, st));
  // 1. the scan: its lists go to segKeys / segCnt (through the preKeys / preCnt fields), the table maxima to qmax
  {
    PqtRsArgs sargs = rargs;
    sargs.preKeys = segKeys; sargs.preCnt = segCnt; sargs.preQmax = qmax;
    auto kern = pqt_k_pair_scan<LPV, UV, SH, 6>;
    const size_t lds = (size_t)8 * idx->dp.LP * idx->dp.C1 * 4 + (size_t)16 * (512 * 8 + (size_t)idx->curRunCap * 12) + 64;
    if ((rc = allowLds(kern, lds))) return rc;
    const uint32_t grid = std::min<uint32_t>((nq + 7) / 8, (uint32_t)idx->numCUs);
    hipExtLaunchKernelGGL(kern, dim3(grid), dim3(1024), (uint32_t)lds, st, idx->lev0, nullptr, 0u, sargs, idx->d_coopErr);
  }
  // 2. the two lists of a query -> its 256 smallest keys, ascending
  rargs.preDist = idx->d_candDist; rargs.preOk = ones; rargs.preQmax = qmax; rargs.preFlags = nullptr;
  rargs.preKeys = idx->d_srKeys; rargs.preCnt = reinterpret_cast<uint32_t*>(idx->d_srKeys + (size_t)nq * 256);
  HIPCHK(hipMemsetAsync(rargs.preCnt, 0, (size_t)nq * 4, st));  // (0xffffffff would mean "handed back by the scan")
  {
    constexpr int MW = 4;
    hipLaunchKernelGGL((pqt_k_sr_merge<MW, 2>), dim3((nq + MW - 1) / MW), dim3(MW * 64), 0, st, rargs, (const unsigned long long*)segKeys, (const uint32_t*)segCnt);
  }
  // 3. band re-evaluation with the reference association, sort, results (the launch that follows the shared-row pass's scan)
  {
    constexpr int BW = 12;
    auto kern = pqt_k_sr_select<BW, LPV, UV, SH, 6, false, 3>;
    const size_t lds = (size_t)BW * ((PQT_RS_BEST + PQT_RS_PEND) * 8 + (size_t)idx->curRunCap * 12 + (size_t)idx->dp.LP * idx->dp.C1 * 4);
    if ((rc = allowLds(kern, lds))) return rc;
    const uint32_t grid = std::min<uint32_t>((nq + BW - 1) / BW, (uint32_t)idx->numCUs);
    hipExtLaunchKernelGGL(kern, dim3(grid), dim3(BW * 64), (uint32_t)lds, st, nullptr, idx->lev1, 0u, rargs);
  }
  // 4. queries whose near-tie band did not fit the 256 slots: exact distances by whole workgroups, exact selection
  return launchHandedBack(idx, st, rargs);
}
int launchCoopRerank(pqt_index* idx, hipStream_t st, const float* v, const uint32_t* nl, uint64_t stride, uint32_t k, uint32_t nq, uint32_t* oI, float* oD, uint32_t* oP) {
  if (!sharedRowsShape(idx)) return pqtFail(PQT_ERR_LIMIT, "cooperative rerank: 32 line parts, C1 = 64 only");
  return idx->sharded ? launchCoop<true>(idx, st, v, nl, stride, k, nq, oI, oD, oP) : launchCoop<false>(idx, st, v, nl, stride, k, nq, oI, oD, oP);
}

int launchSharedSelect(pqt_index* idx, uint32_t grid, size_t lds, hipStream_t st, const float* v, const uint32_t* nl,
                       uint64_t stride, uint32_t k, uint32_t nq, uint32_t* oI, float* oD, uint32_t* oP) {
  (void)grid; (void)lds;
  return idx->sharded ? launchSel<true>(idx, st, v, nl, stride, k, nq, oI, oD, oP) : launchSel<false>(idx, st, v, nl, stride, k, nq, oI, oD, oP);
}
