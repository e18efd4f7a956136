// pqt_hip.hip -- libpqt_hip.so: index management + launch logic behind the C-ABI of include/pqt_hip.h.
// gfx950 (MI355X) only.  Build: see csrc/Makefile (hipcc --offload-arch=gfx950 -ffp-contract=off).
#define PQT_MAIN_TU 1   // the non-template kernels of pqt_kernels.h are compiled here only
#include "pqt_internal.h"

namespace {
thread_local std::string g_err;
}  // namespace
int pqtFail(int code, const std::string& msg) { g_err = msg; return code; }

namespace {
std::mutex g_ldsMu;
std::map<std::pair<int, const void*>, size_t> g_ldsSet;
}  // namespace
int pqtAllowLds(const void* kernel, size_t bytes) {
  if (bytes > kMaxLds) return fail(PQT_ERR_LIMIT, "request needs more than 160 KiB of LDS per workgroup");
  if (bytes <= 64 * 1024) return PQT_OK;
  int dev = 0;
  HIPCHK(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_ldsMu);
  size_t& have = g_ldsSet[{dev, kernel}];
  if (bytes > have) {
    HIPCHK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    have = bytes;
  }
  return PQT_OK;
}

namespace {

int ensureQueryScratch(pqt_index* idx, uint32_t qn) {
  if (qn <= idx->qCap) return PQT_OK;
  const PqtDevParams& d = idx->dp;
  int rc;
  if ((rc = devAlloc(&idx->d_qL1virt, (size_t)qn * d.LP * d.C1))) return rc;
  if ((rc = devAlloc(&idx->d_segD, (size_t)qn * d.P * d.WC))) return rc;
  if ((rc = devAlloc(&idx->d_segBin, (size_t)qn * d.P * d.WC))) return rc;
  if ((rc = devAlloc(&idx->d_nCand, (size_t)qn))) return rc;
  if ((rc = devAlloc(&idx->d_nLocal, (size_t)qn))) return rc;
  if ((rc = devAlloc(&idx->d_nIncl, (size_t)qn))) return rc;
  if ((rc = devAlloc(&idx->d_ovList, (size_t)qn))) return rc;
  if (!idx->d_ovCount && (rc = devAlloc(&idx->d_ovCount, (size_t)2))) return rc;
  if ((rc = devAlloc(&idx->d_fbList, (size_t)2 * qn))) return rc;  // second half: the list the mid-size pass of the k > 128 path leaves for the block-wide kernel
  if (!idx->d_fbCount && (rc = devAlloc(&idx->d_fbCount, (size_t)2))) return rc;
  if ((rc = devAlloc(&idx->d_tvList, (size_t)qn))) return rc;
  if (!idx->d_tvCount && (rc = devAlloc(&idx->d_tvCount, (size_t)2))) return rc;
  idx->qCap = qn;
  return PQT_OK;
}

int ensureCandScratch(pqt_index* idx, uint64_t slots) {
  if (slots <= idx->candCap) return PQT_OK;
  int rc;
  if ((rc = devAlloc(&idx->d_cand, slots))) return rc;
  if ((rc = devAlloc(&idx->d_candDist, slots))) return rc;
  if (idx->sharded) { if ((rc = devAlloc(&idx->d_candPos, slots))) return rc; }
  idx->candCap = slots;
  return PQT_OK;
}

// host-side construction of the open-addressing bin table
struct BinDesc { uint32_t key, gcount, lstart, lcount, lower; };
int uploadBins(pqt_index* idx, const std::vector<BinDesc>& bins, const std::vector<uint32_t>& localIds, bool sharded) {
  int rc = setDevice(idx);
  if (rc) return rc;
  // two-choice (cuckoo) table at load factor <= 0.4: every look-up is two independent reads
  uint32_t bits = 4;
  while (((uint64_t)1 << bits) * 2 < 5 * (uint64_t)bins.size() && bits < 31) ++bits;
  if (((uint64_t)1 << bits) < 2 * (uint64_t)bins.size() + 1) return fail(PQT_ERR_LIMIT, "too many bins for the table");
  const size_t tsz = (size_t)1 << bits;
  std::vector<PqtBinEntry> table;
  std::vector<uint32_t> lower;
  uint32_t maxBin = 0;
  uint32_t seed = 0x5bd1e995u, usedSeed = 0;
  bool built = false;
  for (int attempt = 0; attempt < 16 && !built; ++attempt, seed = seed * 1664525u + 1013904223u) {
    usedSeed = seed;
    table.assign(tsz, PqtBinEntry{0, 0, 0, 0});
    if (sharded) lower.assign(tsz, 0);
    built = true;
    maxBin = 0;
    for (const BinDesc& b0 : bins) {
      if (b0.gcount == 0) continue;
      maxBin = std::max(maxBin, b0.gcount);
      PqtBinEntry cur{b0.key, b0.gcount, b0.lstart, b0.lcount};
      uint32_t curLower = b0.lower;
      // duplicate check against both homes of the new key
      {
        const uint32_t a = pqt_hash1(cur.key, bits), c = pqt_hash2(cur.key, bits, seed);
        if ((table[a].gcount && table[a].key == cur.key) || (table[c].gcount && table[c].key == cur.key))
          return fail(PQT_ERR_INVALID, "duplicate bin id in bin list");
      }
      uint32_t slot = pqt_hash1(cur.key, bits);
      bool placed = false;
      for (int kick = 0; kick < 512; ++kick) {
        if (table[slot].gcount == 0) { table[slot] = cur; if (sharded) lower[slot] = curLower; placed = true; break; }
        // evict the resident, move it to its other home
        std::swap(cur, table[slot]);
        if (sharded) std::swap(curLower, lower[slot]);
        const uint32_t a = pqt_hash1(cur.key, bits), c = pqt_hash2(cur.key, bits, seed);
        slot = (slot == a) ? c : a;
      }
      if (!placed) { built = false; break; }
    }
  }
  if (!built) return fail(PQT_ERR_LIMIT, "could not build the bin table (cuckoo insertion failed for 16 seeds)");
  idx->dp.tableSeed = usedSeed;
  if ((rc = devAlloc(&idx->d_table, tsz))) return rc;
  HIPCHK(hipMemcpy(idx->d_table, table.data(), tsz * sizeof(PqtBinEntry), hipMemcpyHostToDevice));
  if (idx->d_lower) { (void)hipFree(idx->d_lower); idx->d_lower = nullptr; }
  if (sharded) {
    if ((rc = devAlloc(&idx->d_lower, tsz))) return rc;
    HIPCHK(hipMemcpy(idx->d_lower, lower.data(), tsz * 4, hipMemcpyHostToDevice));
  }
  {
    // presence filter: >= 64 bits per bin (<= 1.6 % set), between 2^16 and 2^28 bits
    uint32_t fb = 16;
    const uint64_t perBin = getenv("PQT_FILTER_BITS_PER_BIN") ? (uint64_t)atoi(getenv("PQT_FILTER_BITS_PER_BIN")) : 64;
    while (((uint64_t)1 << fb) < perBin * (uint64_t)bins.size() && fb < 28) ++fb;
    std::vector<uint32_t> filt((size_t)1 << (fb - 5), 0u);
    for (const BinDesc& b0 : bins) {
      if (b0.gcount == 0) continue;
      const uint32_t bit = pqt_hash_filter(b0.key, fb);
      filt[bit >> 5] |= 1u << (bit & 31u);
    }
    if ((rc = devAlloc(&idx->d_filter, filt.size()))) return rc;
    HIPCHK(hipMemcpy(idx->d_filter, filt.data(), filt.size() * 4, hipMemcpyHostToDevice));
    idx->filterBits = fb;
    // first level for the wide enumeration (pqt_k_traverse_f1): the bitmap folded to at most 2^19 bits = 64 KB of LDS, bit i = OR of the
    // bits whose index starts with i; kept only while it still rejects most rows (at most ~40 % of its bits set)
    idx->filter1Bits = 0;
    if (idx->d_filter1) { (void)hipFree(idx->d_filter1); idx->d_filter1 = nullptr; }
    {
      const uint32_t f1 = std::min<uint32_t>(fb, 19u);
      size_t nonEmpty = 0;
      for (const BinDesc& b0 : bins) nonEmpty += b0.gcount != 0;
      if (f1 >= 12 && (double)nonEmpty <= 0.5 * (double)((uint64_t)1 << f1)) {
        std::vector<uint32_t> f1w((size_t)1 << (f1 - 5), 0u);
        for (const BinDesc& b0 : bins) {
          if (b0.gcount == 0) continue;
          const uint32_t bit = pqt_hash_filter(b0.key, fb) >> (fb - f1);
          f1w[bit >> 5] |= 1u << (bit & 31u);
        }
        if ((rc = devAlloc(&idx->d_filter1, f1w.size()))) return rc;
        HIPCHK(hipMemcpy(idx->d_filter1, f1w.data(), f1w.size() * 4, hipMemcpyHostToDevice));
        idx->filter1Bits = f1;
      }
    }
  }
  if ((rc = devAlloc(&idx->d_ids, localIds.size()))) return rc;
  if (!localIds.empty()) HIPCHK(hipMemcpy(idx->d_ids, localIds.data(), localIds.size() * 4, hipMemcpyHostToDevice));
  idx->nIds = localIds.size();
  idx->tableBits = bits; idx->maxBin = maxBin; idx->sharded = sharded; idx->haveBins = true;
  // a new bin layout invalidates the bin-ordered line store; if the id-ordered copy was consumed by the reorder, the
  // caller has to hand the line codes over again (the next query reports PQT_ERR_STATE otherwise)
  if (idx->d_codesBin) { (void)hipFree(idx->d_codesBin); idx->d_codesBin = nullptr; }
  idx->binOrdered = false;
  // candidate scratch layout depends on sharded-ness
  idx->candCap = 0;
  return PQT_OK;
}

size_t ldsTables(const PqtDevParams& d) { return (size_t)(d.D + d.LP * d.C1 + d.P * d.C1 + d.P * d.W + d.P * d.WC) * 4; }
size_t ldsBins(const PqtDevParams& d, uint32_t cap, uint32_t capP2, bool sharded) {
  return (size_t)capP2 * 8 + (size_t)cap * 4 * (sharded ? 4 : 2) + (size_t)d.P * d.WC * 8 + (PQT_BLOCK / 64 + 1 + 4) * 4;
}
size_t ldsSelect(uint32_t kP2) { return (size_t)kP2 * 8 + (256 + 8 + PQT_BLOCK / 64 + 1) * 4; }
size_t ldsEncode(const PqtDevParams& d) {
  return (size_t)(d.D + d.LP * d.C1 + d.P * d.C1 + d.P + d.P * d.C2 + ((d.P * d.C2 + d.P) & 1)) * 4 + (size_t)PQT_BLOCK * 8;
}

// permute the line store into bin order (see pqt_k_reorder_lines); afterwards the id-ordered copy is dropped if owned
int reorderLines(pqt_index* idx) {
  if (idx->isView) return fail(PQT_ERR_STATE, "view handle asked to rebuild shared data (line store)");
  int rc = setDevice(idx);
  if (rc) return rc;
  const uint32_t LP = idx->dp.LP;
  if ((rc = devAlloc(&idx->d_codesBin, (size_t)idx->nIds * LP))) return rc;
  unsigned long long* bad = idx->d_counters + 8 * kCtrRing;
  HIPCHK(hipMemsetAsync(bad, 0, 8, idx->stream));
  const uint64_t pieces = idx->nIds * (uint64_t)(LP % 4 == 0 ? LP / 4 : LP);
  const uint64_t maxGrid = 1ull << 30;
  if (pieces > maxGrid * 256) return fail(PQT_ERR_LIMIT, "line store too large for one reorder launch");
  if (pieces) hipLaunchKernelGGL(pqt_k_reorder_lines, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, idx->stream, idx->d_codes,
                                 idx->idBase, idx->nCodes, idx->d_ids, idx->nIds, LP, idx->d_codesBin, bad);
  HIPCHK(hipGetLastError());
  unsigned long long nbad = 0;
  HIPCHK(hipMemcpyAsync(&nbad, bad, 8, hipMemcpyDeviceToHost, idx->stream));
  HIPCHK(hipStreamSynchronize(idx->stream));
  if (nbad) return fail(PQT_ERR_STATE, "bin members reference vector ids outside the line store [id_base, id_base + nvec)");
  if (idx->codesOwned && idx->d_codes) { (void)hipFree(idx->d_codes); idx->d_codes = nullptr; idx->codesOwned = false; idx->linesDropped = true; }
  if (idx->d_codesGrp) { (void)hipFree(idx->d_codesGrp); idx->d_codesGrp = nullptr; idx->grpG = 0; }
  if (idx->d_codesX) { (void)hipFree(idx->d_codesX); idx->d_codesX = nullptr; idx->xcodeShift = 0; }
  idx->biasReady = false;
  idx->binOrdered = true;
  return PQT_OK;
}

}  // namespace
int ensureGroupMajor(pqt_index* idx, int G) {
  if (idx->d_codesGrp && idx->grpG == G) return PQT_OK;
  if (idx->isView) return fail(PQT_ERR_STATE, "view handle asked to rebuild shared data (group-major store)");
  int rc;
  const size_t words = (size_t)idx->nIds * idx->dp.LP;
  if ((rc = devAlloc(&idx->d_codesGrp, words))) return rc;
  if (words) hipLaunchKernelGGL(pqt_k_group_major, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, idx->stream, idx->d_codesBin,
                                (uint64_t)idx->nIds, idx->dp.LP, (uint32_t)G, idx->d_codesGrp);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(idx->stream));
  idx->grpG = G;
  return PQT_OK;
}
int ensureXCode(pqt_index* idx, int c1Shift) {
  if (idx->d_codesX && idx->xcodeShift == c1Shift) return PQT_OK;
  if (idx->isView) return fail(PQT_ERR_STATE, "view handle asked to rebuild shared data (X-code store)");
  int rc;
  const size_t words = (size_t)idx->nIds * idx->dp.LP;
  if ((rc = devAlloc(&idx->d_codesX, words))) return rc;
  if (words) hipLaunchKernelGGL(pqt_k_xcode, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, idx->stream, idx->d_codesBin, (uint64_t)words, (uint32_t)c1Shift, idx->d_codesX);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(idx->stream));
  idx->xcodeShift = c1Shift;
  return PQT_OK;
}
namespace {

// opt-in adc_bias mode: bias[pos] of every row of the bin-ordered store (once per index / line store)
int ensureBias(pqt_index* idx) {
  if (idx->biasReady) return PQT_OK;
  if (idx->isView) return fail(PQT_ERR_STATE, "view handle asked to rebuild shared data (row bias)");
  int rc;
  if ((rc = devAlloc(&idx->d_bias, (size_t)idx->nIds))) return rc;
  if (idx->nIds) hipLaunchKernelGGL(pqt_k_adc_bias, dim3((unsigned)((idx->nIds + 255) / 256)), dim3(256), 0, idx->stream, idx->d_codesBin, (uint64_t)idx->nIds,
                                    idx->d_coarse, idx->dp, idx->d_bias);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(idx->stream));
  idx->biasReady = true;
  return PQT_OK;
}

// per-stage ms of one chunk of one ring slot: the time between two consecutive RECORDED events goes to the stage the later
// one closes ({tables, traversal, order, rerank, select}); returns the last recorded event
int stageMs(const pqt_index* idx, int slot, int ch, float st[5]) {
  int prev = EV_BEGIN;
  if (!(idx->evMask[slot][ch] & 1u)) return -1;
  for (int e = 1; e < EV_COUNT; ++e) {
    if (!(idx->evMask[slot][ch] & (1u << e))) continue;
    float ms = 0;
    if (hipEventElapsedTime(&ms, idx->evRing[slot][ch][prev], idx->evRing[slot][ch][e]) == hipSuccess) st[e - 1] += ms;
    prev = e;
  }
  return prev;
}

// ring slot of the most recent query call that carried per-kernel events (option "stage_timing" = N > 1: not every call
// does), or -1
int lastTimedSlot(const pqt_index* idx) {
  const int have = (int)std::min<unsigned long long>(idx->calls, (unsigned long long)kRing);
  for (int b = 1; b <= have; ++b) {
    const int slot = (int)((idx->calls - b) % kRing);
    if (idx->ringChunks[slot] > 0 && (idx->evMask[slot][0] & 1u)) return slot;
  }
  return -1;
}

// binsIn != null (pqt_query_shard_bins): the traversal of every query ran elsewhere (pqt_traverse_bins) and arrives as
// binsIn[qn][binsCap + 1]; this shard resolves the listed bins against its own table instead of traversing
int queryImpl(pqt_index* idx, const float* q_dev, uint32_t qn, uint32_t Bv, uint32_t Bb, uint32_t k,
              uint32_t* outIdx, float* outDist, uint32_t* outPos, uint32_t* outCount, hipStream_t st, int sync,
              const unsigned long long* binsIn = nullptr, uint32_t binsCap = 0) {
  if (!idx) return fail(PQT_ERR_INVALID, "null index");
  if (!idx->haveTree || !idx->haveBins || !(idx->d_codes || idx->binOrdered) || !idx->d_heur) return fail(PQT_ERR_STATE, "index needs codebooks, heuristic, bins and line codes before querying");
  if (qn == 0) return PQT_OK;
  if (k == 0 || !q_dev || !outIdx || !outDist) return fail(PQT_ERR_INVALID, "bad query arguments");
  if (idx->sharded && !outPos) return fail(PQT_ERR_INVALID, "sharded index: use pqt_query_shard");
  int rc = setDevice(idx);
  if (rc) return rc;
  if (!st) st = idx->stream;
  if (!idx->binOrdered && (rc = reorderLines(idx))) return rc;
  const PqtDevParams& d = idx->dp;
  // number of heuristic rows enumerated (treequantizer.hpp:552)
  uint64_t He64 = std::min<uint64_t>(Bb, idx->maxMultiIndex);
  if (He64 > idx->heurRows) return fail(PQT_ERR_STATE, "bound_bins exceeds the heuristic rows held by the index (build/set a longer prefix)");
  const uint32_t He = (uint32_t)He64;
  const uint32_t HeP2 = np2(std::max<uint32_t>(He, 2));
  const size_t lBins = ldsBins(d, He, HeP2, idx->sharded);
  if (lBins > kMaxLds || He > 8192) return fail(PQT_ERR_LIMIT, "bound_bins too large for the LDS-resident bin sort (limit 8192)");
  const uint32_t cap1 = std::min<uint32_t>(He, 1024), cap1P2 = np2(std::max<uint32_t>(cap1, 2));
  const size_t lBins1 = ldsBins(d, cap1, cap1P2, idx->sharded);
  // candidate list bound: the reference overshoots Bv by at most the bin that crosses it
  uint64_t stride = std::min<uint64_t>((uint64_t)Bv + idx->maxBin + 1, (uint64_t)He * idx->maxBin + 1);
  stride = (stride + 63) & ~(uint64_t)63;
  if (stride > ((uint64_t)1 << 31)) return fail(PQT_ERR_LIMIT, "candidate list bound (bound_vectors + largest bin) exceeds 2^31 entries per query");
  idx->stride = stride;
  const bool fullSort = (k > 4096);
  const uint32_t kP2 = np2(std::max<uint32_t>(k, 2));
  const size_t perQueryBytes = stride * (idx->sharded ? 12 : 8) + (fullSort ? (size_t)np2(stride) * 8 : 0);
  uint32_t qChunk = (uint32_t)std::max<size_t>(1, std::min<size_t>(qn, idx->scratchBudget / perQueryBytes));
  int nChunks = (int)((qn + qChunk - 1) / qChunk);
  if (nChunks > kMaxChunks) { qChunk = (qn + kMaxChunks - 1) / kMaxChunks; nChunks = (int)((qn + qChunk - 1) / qChunk); }
  if ((rc = ensureQueryScratch(idx, qn))) return rc;
  if ((rc = ensureCandScratch(idx, (uint64_t)qChunk * stride))) return rc;
  const uint32_t sortP2 = fullSort ? np2(stride) : 0;
  if (fullSort && (uint64_t)qChunk * sortP2 > idx->sortCap) {
    if ((rc = devAlloc(&idx->d_sortKeys, (size_t)qChunk * sortP2))) return rc;
    idx->sortCap = (uint64_t)qChunk * sortP2;
  }
  if (!idx->evCreated) {
    for (int r = 0; r < kRing; ++r) for (int c = 0; c < kMaxChunks; ++c) for (int e = 0; e < EV_COUNT; ++e) HIPCHK(hipEventCreate(&idx->evRing[r][c][e]));
    idx->evCreated = true;
  }
  // statistics block of this call; the next call's block is zeroed by the rerank kernel of this one (or a memset at the
  // end when that kernel does not run), which saves a launch per call
  idx->ctr = idx->d_counters + 8 * idx->ctrPos;
  unsigned long long* nextCtr = idx->d_counters + 8 * ((idx->ctrPos + 1) % kCtrRing);
  bool nextZeroed = false;

  const size_t lTab = ldsTables(d);
  if ((rc = allowLds(pqt_k_tables, lTab))) return rc;
  if (idx->sharded) { if ((rc = allowLds(pqt_k_bins<true>, lBins))) return rc; } else { if ((rc = allowLds(pqt_k_bins<false>, lBins))) return rc; }
  const size_t lRer = (size_t)d.LP * d.C1 * 4;
  const size_t lSel = ldsSelect(kP2);
  if (!fullSort) { if (idx->sharded) { if ((rc = allowLds(pqt_k_select<true>, lSel))) return rc; } else { if ((rc = allowLds(pqt_k_select<false>, lSel))) return rc; } }

  TravPlan tplan;
  if ((rc = planTraversal(idx, He, tplan))) return rc;
  const bool travFused = tplan.fused, travWide = tplan.wide, travP2 = tplan.p2;
  // the exchanged bin lists are only used together with the fused traversal (which re-traverses the overflowed queries); other
  // shapes / bounds traverse every query here
  if (binsIn && (!travFused || !idx->sharded || binsCap > PQT_GBIN_MAX)) binsIn = nullptr;
  // the short fused traversal writes the caller's candidate counts itself (no device-to-device copy on the stream); the wide
  // one can hand queries to pqt_k_bins, which fills them in later: copy at the end
  const bool countDirect = outCount && travFused && !travWide;
  // fused rerank+select (wave per query) whenever the result list fits the in-register selector
  bool fused = (k <= PQT_RS_BEST) && (d.LP == 4 || d.LP == 8 || d.LP == 16 || d.LP == 32) && !idx->forceUnfused;
  const size_t coarseBytes = (size_t)d.LP * d.C1 * d.C1 * 4;
  const bool coarseLds = coarseBytes <= 64 * 1024;
  const size_t lFused = (coarseLds ? coarseBytes : 0) + (size_t)kFusedWaves * ((PQT_RS_BEST + PQT_RS_PEND) * 8 + (size_t)d.LP * d.C1 * 4) + 16 + 3 * PQT_RS_LIST * 4;
  // adc_bias mode: wave-per-query kernel without the coarse table; NW = 12 or 6 wavefronts around per-wave L1virt copies
  const size_t lBias12 = (size_t)12 * ((PQT_RS_BEST + PQT_RS_PEND) * 8 + (size_t)d.LP * d.C1 * 4) + 16 + 3 * PQT_RS_LIST * 4;
  const size_t lBias6 = (size_t)6 * ((PQT_RS_BEST + PQT_RS_PEND) * 8 + (size_t)d.LP * d.C1 * 4) + 16 + 3 * PQT_RS_LIST * 4;
  const bool biasShape = (d.LP == 16 || d.LP == 32) && (d.C1 & (d.C1 - 1)) == 0 && lBias6 <= kMaxLds;
  // MODE 1: opt-in adc_bias distances.  MODE 2 (default when the coarse table does not fit the LDS): reference distances
  // through the MODE 1 filter -- replaces the workgroup-per-query kernel with its staged table slices
  const bool useFilter = !idx->adcBias && idx->exactFilter && !coarseLds && fused && biasShape && std::isfinite(idx->coarseMax);
  const bool useBias = (idx->adcBias && fused && biasShape) || useFilter;
  const int biasNW = lBias12 <= kMaxLds ? 12 : 6;
  if (useBias) {
    if ((rc = ensureGroupMajor(idx, 4))) return rc;
    if ((rc = ensureBias(idx))) return rc;
  }
  int wgG = (fused && !useBias && !coarseLds && idx->useWgRerank) ? rswgGroup(d) : 0;
  if (wgG && (size_t)wgG * d.C1 * d.C1 * 4 + (size_t)d.LP * d.C1 * 4 + (size_t)PQT_RS2_NW * PQT_RS2_KEYS * 8 > kMaxLds) wgG = 0;
  // a shape whose fused kernel does not fit the LDS (e.g. C1 = 256 with >= 16 line parts) runs the staged rerank/select,
  // which needs LP*C1*4 bytes only
  if (fused && !wgG && !useBias && lFused > kMaxLds) fused = false;  // (useBias implies lBias6 <= kMaxLds: the 6- or 12-wave filter kernel is the one launched)
  // bin runs instead of a candidate list: fused traversal -> MODE 0 rerank with the LDS table (the SIFT1M shapes)
  // bin runs instead of a candidate list.  MODE 0 with the LDS table (SIFT1M shape): only on request (measured a net loss);
  // MODE 2 at the configs[2]/[3] shape (long bins, few runs per query: 64 slots per wave): on unless switched off.
  const size_t lRuns = ((lFused + 15) & ~(size_t)15) + (size_t)kFusedWaves * PQT_RUNCAP * 12;
  const size_t lRunsBig = ((lBias12 + 15) & ~(size_t)15) + (size_t)12 * 64 * 12;
  const bool runsSmall = idx->useRuns == 1 && travFused && fused && coarseLds && !useBias && !wgG && lRuns <= kMaxLds && d.C1 == 32 && d.LP == 16;
  const bool runsBig = idx->useRuns != 0 && travFused && useBias && biasNW == 12 && d.C1 == 64 && d.LP == 32 && lRunsBig <= kMaxLds;
  const bool emitRuns = runsSmall || runsBig;
  idx->curRunCap = runsBig ? 64u : (uint32_t)PQT_RUNCAP;
  if (emitRuns && (uint64_t)qChunk * PQT_RUNCAP > idx->runsCap) {
    if ((rc = devAlloc(&idx->d_runs, (size_t)qChunk * PQT_RUNCAP))) return rc;
    if ((rc = devAlloc(&idx->d_runGpos, (size_t)qChunk * PQT_RUNCAP))) return rc;
    if ((rc = devAlloc(&idx->d_nRuns, (size_t)qChunk))) return rc;
    idx->runsCap = (uint64_t)qChunk * PQT_RUNCAP;
  }
  idx->curRuns = emitRuns;
  // shared-row pass in front of the filtered selection (configs[2]/[3] shape with bin runs).  Automatic choice: where the line store is far
  // beyond the caches (the pass costs eight small launches per chunk); on a range shard only when the vector bound reaches past the first
  // long bin of a query -- the pass wins by the visits of one query to the same (aliased) bin and by bins shared between queries, and a
  // shard's slices of the bins are short.  Measured at 100 M, ms per 10 k-query batch with / without: unsharded (20000, 500) 2.9 / 5.0,
  // (4096, 4096) 2.77 / 2.89; one shard of eight, two batches in flight: (20000, 500) 0.69 / 0.82, (4096, 4096) 0.70 / 0.61
  const bool sharedPass = runsBig && useFilter && sharedRowsShape(idx) && !(idx->dbg & 0xffffu) &&
                          (idx->sharedRows == 1 || (idx->sharedRows < 0 && (size_t)idx->nIds * d.LP * 4 >= ((size_t)1 << 30) &&
                                                    (!idx->sharded || (uint64_t)Bv * 2 >= idx->maxBin)));
  // cooperative filter scan (opt-in, round 6): the same shapes as the pass, where the pass does not run
  const bool coopPass = idx->coopRerank == 1 && !sharedPass && runsBig && useFilter && sharedRowsShape(idx) && !(idx->dbg & 0xffffu);
  // X-code rows for the exact rerank with the LDS table at C1 = 32 (SIFT1M shape): a second copy of the line store with cheaper
  // address arithmetic (pqt_rs_query XC); not for stores beyond 16 GiB (the copy doubles their footprint) and not with bin runs
  bool xcode = idx->useXCode != 0 && fused && !useBias && !wgG && coarseLds && d.C1 == 32 && (d.LP == 4 || d.LP == 8 || d.LP == 16 || d.LP == 32) && !emitRuns &&
                     (size_t)idx->nIds * d.LP * 4 <= ((size_t)16 << 30);
  const size_t lFusedX = coarseBytes + (size_t)kXcWaves * ((size_t)kXcSlots * 8 + (size_t)d.LP * d.C1 * 4) + 16 + 3 * PQT_RS_LIST * 4;
  const bool xcodeFits = lFusedX <= kMaxLds;
  if (xcode && xcodeFits && (rc = ensureXCode(idx, 5))) {
    // automatic mode: a second copy of the line store that cannot be had (out of memory; a view whose owner has not built it yet) is not
    // an error -- the plain bin-ordered rows serve the same kernel family; "xcode" = 1 keeps the failure visible
    if (idx->useXCode > 0) return rc;
    (void)hipGetLastError();
    xcode = false;
  }
  xcode = xcode && xcodeFits;
  idx->curXCode = xcode;
  // 128 < k <= 4096 (queryKNN(.., 4096) of the reference front-end): workgroup-per-query fused rerank+select, distances on chip
  const uint32_t kcap = std::max<uint32_t>(2 * kP2, 1024);
  const size_t lBigBase = (size_t)d.LP * d.C1 * 4 + (size_t)kcap * 8 + 256 * 4 + 4 * 8 + 16;
  bool bigK = !idx->forceUnfused && !fullSort && k > PQT_RS_BEST && (d.LP * d.C1) % 4 == 0 && kcap <= 8192;
  const bool bigCL = coarseLds && coarseBytes + lBigBase <= kMaxLds;
  if (bigK && !bigCL && lBigBase > kMaxLds) bigK = false;
  const size_t lBig = lBigBase + (bigCL ? coarseBytes : 0);
  bool usedSmallFirst = false;
  bool usedOneLaunch = false;
  bool usedMid = false;
  idx->nChunks = nChunks;
  idx->ringPos = (int)(idx->calls % kRing);
  idx->ringChunks[idx->ringPos] = nChunks;
  idx->timedCall = idx->stageTiming > 0 && (idx->timingPhase++ % (unsigned long long)idx->stageTiming) == 0;  // the first call after the option is set is a timed one
  idx->calls++;
  // the two fused launches are bracketed by three events; the staged path keeps one event per stage
  const bool leanEvents = travFused && (fused || bigK);
#define PQT_REC(e) do { if (idx->timedCall) { HIPCHK(hipEventRecord(idx->evRing[idx->ringPos][c][e], st)); idx->evMask[idx->ringPos][c] |= 1u << (e); } } while (0)
  for (int c = 0; c < nChunks; ++c) {
    const uint32_t q0 = (uint32_t)c * qChunk;
    const uint32_t nq = std::min<uint32_t>(qChunk, qn - q0);
    idx->evMask[idx->ringPos][c] = 0;
    idx->lev0 = idx->lev1 = nullptr;
    if (leanEvents && idx->timedCall) {
      // lean timing: start/stop timestamps ride on the two fused dispatches (no event packets between the kernels)
      idx->lev0 = idx->evRing[idx->ringPos][c][EV_BEGIN]; idx->lev1 = idx->evRing[idx->ringPos][c][EV_BINS];
      idx->evMask[idx->ringPos][c] |= (1u << EV_BEGIN) | (1u << EV_BINS);
    } else PQT_REC(EV_BEGIN);
    unsigned long long* const tstamp = (nq <= (1u << 16)) ? idx->d_tstamp : nullptr;  // debug buffer holds 65536 query records
    // rerank schedule of this chunk (decided here: for schedule 2 the traversal kernels register the queries by size class)
    const uint32_t rsNW = useBias ? (uint32_t)biasNW : (xcode ? (uint32_t)kXcWaves : (uint32_t)kFusedWaves);
    static const int envGrid = getenv("PQT_RS_GRID") ? atoi(getenv("PQT_RS_GRID")) : 0;  // experiment: workgroups of the persistent rerank launch
    const uint32_t rsGrid = std::min<uint32_t>((nq + rsNW - 1) / rsNW, envGrid > 0 ? (uint32_t)envGrid : (uint32_t)idx->numCUs);
    const bool severalPerWave = fused && !wgG && nq > rsGrid * rsNW;
    // automatic choice: the global pools pay for their registration atomics and chunk draws when a query brings thousands of
    // candidates from an HBM-resident store (configs[2]/[3]: -8..-10 % on the rerank launch); with a cache-resident store and
    // a few hundred candidates per query the fixed share per workgroup is 1-2 % ahead (SIFT1M shape)
    // (round 3: the filtered rerank also prefers the pools on a cache-resident store -- 160 MB shard of the 8-way 10 M layout: 0.334 -> 0.318 ms)
    const int balance = idx->balance >= 0 ? idx->balance : ((useBias || (size_t)idx->nIds * d.LP * 4 > ((size_t)256 << 20)) ? 2 : 1);
    const bool useSched = severalPerWave && balance == 2;
    uint32_t* const schedCntArg = useSched ? poolBlock(idx, idx->poolPos) + 16 : nullptr;
    if (useSched) {
      // the registration counts of this block are zeroed by the previous rerank launch; a call that failed between its
      // traversal and its rerank left them dirty: clear the block before it is registered into again
      if (idx->poolDirty) HIPCHK(hipMemsetAsync(poolBlock(idx, idx->poolPos), 0, kPoolWords * sizeof(uint32_t), st));
      idx->poolDirty = true;
    }
    idx->curSchedCap = (nq + 7) / 8;  // entries a (pool, class) list can be asked to hold: a pool's queries
    if (useSched && (uint64_t)idx->curSchedCap > idx->schedCapQ) {
      if ((rc = devAlloc(&idx->d_schedList, (size_t)8 * PQT_SCHED_CLASSES * idx->curSchedCap))) return rc;
      idx->schedCapQ = idx->curSchedCap;
    }
    // SIFT1M shape, short traversal, plain exact rerank with the LDS table, unsharded: the whole query in ONE launch (pqt_k_query_fused).
    // Opt-in ("one_launch" = 1): measured 0.191 against 0.167 ms per 10 k queries for the two launches (kernel comment).
    bool oneLaunch = false;
    PqtTravArgs oneLaunchT{};
    if (travFused) {
      // a1..a6 in one launch, one wavefront per query
      if (travWide) HIPCHK(hipMemsetAsync(idx->d_ovCount, 0, 4, st));
      PqtTravArgs targs{q_dev + (size_t)q0 * d.D, idx->d_cb1, idx->d_cb2, (const float4*)idx->d_cb2T, d, (const uint4*)idx->d_heur8, (idx->dbg & 4096u) ? nullptr : idx->d_heur4, He, Bv,
                        idx->d_table, idx->d_lower, idx->tableBits, idx->d_ids, nq, idx->d_qL1virt + (size_t)q0 * d.LP * d.C1, idx->d_cand,
                        idx->d_candPos, idx->d_nCand + q0, idx->d_nLocal + q0, idx->d_nIncl + q0, stride, idx->ctr, tstamp,
                        idx->d_segD + (size_t)q0 * d.P * d.WC, idx->d_segBin + (size_t)q0 * d.P * d.WC, idx->d_ovList, idx->d_ovCount,
                        (idx->dbg & 2048u) ? nullptr : idx->d_filter, idx->filterBits,
                        emitRuns ? idx->d_runs : nullptr, idx->d_runGpos, idx->d_nRuns, idx->curRunCap,
                        countDirect ? outCount + q0 : nullptr, (idx->dbg >> 5) & 3u,
                        schedCntArg, idx->d_schedList, idx->curSchedCap, nullptr, nullptr, nullptr, 0u,
                        (idx->useFilter1 > 0 && !(idx->dbg & 2048u)) ? idx->d_filter1 : nullptr, idx->filter1Bits, idx->useFilter1 == 2 ? 1u : 0u};
      if (binsIn) {
        // query-sharded traversal, receiving side: distance tables of every query, the exchanged bin lists resolved against this
        // shard's table, and the (normally empty) list of queries whose list overflowed at the sender traversed here
        HIPCHK(hipMemsetAsync(idx->d_tvCount, 0, 4, st));
        const PqtResolveArgs rargs{binsIn + (size_t)q0 * (binsCap + 1u), binsCap, idx->d_table, idx->d_lower, idx->tableBits, d.tableSeed, nq,
                                   idx->d_cand, idx->d_candPos, stride, idx->d_nCand + q0, idx->d_nLocal + q0, idx->d_nIncl + q0,
                                   emitRuns ? idx->d_runs : nullptr, idx->d_runGpos, idx->d_nRuns, idx->curRunCap,
                                   countDirect ? outCount + q0 : nullptr, idx->d_tvList, idx->d_tvCount, schedCntArg, idx->d_schedList, idx->curSchedCap};
        const size_t lTR = (size_t)(PQT_L1V_QB * d.D + d.LP * d.C1) * 4;
        auto trKern = binsCap > 128u ? pqt_k_tables_resolve<4, 4> : pqt_k_tables_resolve<4, 2>;  // list entries per lane of the resolving wavefront
        if ((rc = allowLds(trKern, lTR))) return rc;
        const uint32_t nResolve = (nq + 3) / 4, nTab = (nq + PQT_L1V_QB - 1) / PQT_L1V_QB;
        hipExtLaunchKernelGGL(trKern, dim3(nResolve + nTab), dim3(PQT_BLOCK), (uint32_t)lTR, st, idx->lev0, nullptr, 0u, rargs, nTab,
                              q_dev + (size_t)q0 * d.D, idx->d_cb1, idx->d_cb1L, d, idx->d_qL1virt + (size_t)q0 * d.LP * d.C1);
        targs.qlist = idx->d_tvList; targs.qcount = idx->d_tvCount;
        launchFusedTraversal(idx, targs, tplan, nq, st, nullptr, idx->lev1);
      } else {
        oneLaunch = idx->oneLaunch > 0 && !travWide && fused && !wgG && !useBias && coarseLds && !emitRuns && !idx->sharded && !useSched && !tstamp &&
                    travShape(idx, targs) == 1 && queryFusedShape(idx) && !(idx->dbg & 0xffffu) &&
                    coarseBytes + (size_t)kFusedWaves * queryFusedPerWave(idx, tplan) + 16 <= kMaxLds;
        if (oneLaunch) oneLaunchT = targs;  // launched below, where the rerank's arguments are known
        else launchFusedTraversal(idx, targs, tplan, nq, st, idx->lev0, idx->lev1);
      }
      if (travWide) {
        // queries with more than 512 populated rows queued themselves: workgroup-per-query kernel with an He-sized arena
        // on that (usually empty) list, from the sorted lists the traversal left in segD/segBin
        if (idx->sharded)
          hipLaunchKernelGGL(pqt_k_bins<true>, dim3(nq), dim3(PQT_BLOCK), lBins, st, idx->d_segD + (size_t)q0 * d.P * d.WC,
                             idx->d_segBin + (size_t)q0 * d.P * d.WC, idx->d_heur8, He, He, HeP2, Bv, d, idx->d_table, idx->d_lower,
                             idx->tableBits, idx->d_cand, idx->d_candPos, idx->d_nCand + q0, idx->d_nLocal + q0, idx->d_nIncl + q0,
                             stride, idx->d_ovList, idx->d_ovCount, idx->d_ovList, idx->d_ovCount, idx->ctr, schedCntArg, idx->d_schedList, idx->curSchedCap, (uint64_t)0);
        else
          hipLaunchKernelGGL(pqt_k_bins<false>, dim3(nq), dim3(PQT_BLOCK), lBins, st, idx->d_segD + (size_t)q0 * d.P * d.WC,
                             idx->d_segBin + (size_t)q0 * d.P * d.WC, idx->d_heur8, He, He, HeP2, Bv, d, idx->d_table, idx->d_lower,
                             idx->tableBits, idx->d_cand, idx->d_candPos, idx->d_nCand + q0, idx->d_nLocal + q0, idx->d_nIncl + q0,
                             stride, idx->d_ovList, idx->d_ovCount, idx->d_ovList, idx->d_ovCount, idx->ctr, schedCntArg, idx->d_schedList, idx->curSchedCap, (uint64_t)0);
      }
    } else {
    hipLaunchKernelGGL(pqt_k_tables, dim3(nq), dim3(PQT_BLOCK), lTab, st, q_dev + (size_t)q0 * d.D, idx->d_cb1, idx->d_cb2, d,
                       idx->d_qL1virt + (size_t)q0 * d.LP * d.C1, idx->d_segD + (size_t)q0 * d.P * d.WC,
                       idx->d_segBin + (size_t)q0 * d.P * d.WC, idx->ctr);
    const uint16_t* heurArg = idx->d_heur8;
    uint64_t heurStride = 0;
    if (idx->heur2d) {
      // 2-D anisotropic sequences: this chunk's queries get their own row tables from the sorted part lists
      if ((uint64_t)nq * He > idx->heurQCap) {
        if ((rc = devAlloc(&idx->d_heurQ, (size_t)qChunk * He))) return rc;
        idx->heurQCap = (uint64_t)qChunk * He;
      }
      PqtRows2dArgs ra{idx->d_segD + (size_t)q0 * d.P * d.WC, idx->d_seq2d, idx->d_heurQ, idx->seq2dDc, d.WC, std::min<uint32_t>(64u, d.WC), He, nq, {0}};
      for (int j = 0; j < 9; ++j) ra.thr[j] = idx->slopeThr[j];
      hipLaunchKernelGGL(pqt_k_rows_2d, dim3((nq + 3) / 4), dim3(256), 0, st, ra);
      heurArg = reinterpret_cast<const uint16_t*>(idx->d_heurQ);
      heurStride = He;
    }
    PQT_REC(EV_TABLES);
    // pass 1: small LDS arena (high occupancy); queries with more populated bins than it holds queue themselves for
    // pass 2, which runs the same kernel with a full-size arena on that (usually empty) list
    HIPCHK(hipMemsetAsync(idx->d_ovCount, 0, 4, st));
    for (int pass = 0; pass < 2; ++pass) {
      const uint32_t cap = pass == 0 ? cap1 : He, capP2 = pass == 0 ? cap1P2 : HeP2;
      const size_t lds = pass == 0 ? lBins1 : lBins;
      const uint32_t* ql = pass == 0 ? nullptr : idx->d_ovList;
      if (pass == 1 && cap1 >= He) break;
      if (idx->sharded)
        hipLaunchKernelGGL(pqt_k_bins<true>, dim3(nq), dim3(PQT_BLOCK), lds, st, idx->d_segD + (size_t)q0 * d.P * d.WC,
                           idx->d_segBin + (size_t)q0 * d.P * d.WC, heurArg, He, cap, capP2, Bv, d, idx->d_table, idx->d_lower,
                           idx->tableBits, idx->d_cand, idx->d_candPos, idx->d_nCand + q0, idx->d_nLocal + q0, idx->d_nIncl + q0,
                           stride, ql, idx->d_ovCount, idx->d_ovList, idx->d_ovCount, idx->ctr, schedCntArg, idx->d_schedList, idx->curSchedCap, heurStride);
      else
        hipLaunchKernelGGL(pqt_k_bins<false>, dim3(nq), dim3(PQT_BLOCK), lds, st, idx->d_segD + (size_t)q0 * d.P * d.WC,
                           idx->d_segBin + (size_t)q0 * d.P * d.WC, heurArg, He, cap, capP2, Bv, d, idx->d_table, idx->d_lower,
                           idx->tableBits, idx->d_cand, idx->d_candPos, idx->d_nCand + q0, idx->d_nLocal + q0, idx->d_nIncl + q0,
                           stride, ql, idx->d_ovCount, idx->d_ovList, idx->d_ovCount, idx->ctr, schedCntArg, idx->d_schedList, idx->curSchedCap, heurStride);
    }
    }
    // wave-per-query rerank: workgroup-local dynamic schedule when a wavefront slot gets more than one query
    idx->curDynamic = (severalPerWave && balance) ? (useSched ? 2u : 1u) : 0u;
    idx->curZero8 = nullptr;
    // draw counters of this launch (zeroed by the previous one) and the block the launch zeroes for the next
    idx->curPool = poolBlock(idx, idx->poolPos); idx->curPoolNext = poolBlock(idx, idx->poolPos + 1);
    if (fused && !wgG) idx->poolPos++;
    idx->lev0 = idx->lev1 = nullptr;
    if (leanEvents && idx->timedCall) {
      idx->lev0 = idx->evRing[idx->ringPos][c][EV_ORDER]; idx->lev1 = idx->evRing[idx->ringPos][c][EV_RERANK];
      idx->evMask[idx->ringPos][c] |= (1u << EV_ORDER) | (1u << EV_RERANK);
    } else PQT_REC(EV_BINS);
    uint32_t* oI = outIdx + (size_t)q0 * k; float* oD = outDist + (size_t)q0 * k;
    uint32_t* oP = outPos ? outPos + (size_t)q0 * k : nullptr;
    if (fused) {
      // a7 + a8 in one launch, one wavefront per query (distances stay on chip)
      const uint32_t grid = rsGrid;
      if (useBias) {
        idx->curZero8 = nextCtr;
        nextZeroed = true;
        const float* v = idx->d_qL1virt + (size_t)q0 * d.LP * d.C1;
        const uint32_t* nl = idx->d_nLocal + q0;
        if (sharedPass) {
          // rows of a bin read once for all the queries (and all the visits of a query) that include it: pqt_shared_rows.h
          if ((rc = launchSharedRows(idx, st, v, nl, stride, nq, idx->lev0))) return rc;
          if (idx->lev1) {  // timed call: stage "rerank_select" ends with pqt_k_sr_adc, the selection over its distances is stage "select"
            idx->lev1 = idx->evRing[idx->ringPos][c][EV_SELECT];
            idx->evMask[idx->ringPos][c] |= 1u << EV_SELECT;
          }
          rc = launchSharedSelect(idx, grid, lRunsBig, st, v, nl, stride, k, nq, oI, oD, oP);
        } else if (coopPass) {
          // two wavefronts per query around one table copy (pqt_k_pair_scan) + merge + band launches: pqt_shared_rows.h
          rc = launchCoopRerank(idx, st, v, nl, stride, k, nq, oI, oD, oP);
        } else
        rc = launchRSBiasAny(idx, biasNW, useFilter, grid, biasNW == 12 ? (runsBig ? lRunsBig : lBias12) : lBias6, st, v, nl, stride, k, nq, oI, oD, oP);
        if (rc) return rc;
        idx->poolDirty = false;  // the launch consumes this chunk's registrations and zeroes the next block
      } else if (wgG) {
        const float* v = idx->d_qL1virt + (size_t)q0 * d.LP * d.C1;
        rc = launchRSWGAny(idx, wgG, nq, st, v, idx->d_nLocal + q0, stride, k, oI, oD, oP);
        if (rc) return rc;
      } else {
        idx->curZero8 = nextCtr;  // the kernel also zeroes the statistics block of the next call
        nextZeroed = true;
        if (oneLaunch) {
          hipEvent_t e0 = idx->lev0;
          if (leanEvents && idx->timedCall) {  // one dispatch: its start is the call's BEGIN, the traversal's stop / the rerank's start do not exist
            e0 = idx->evRing[idx->ringPos][c][EV_BEGIN];
            idx->evMask[idx->ringPos][c] &= ~((1u << EV_BINS) | (1u << EV_ORDER));
          }
          if ((rc = launchQueryFused(idx, oneLaunchT, tplan, grid, st, idx->d_qL1virt + (size_t)q0 * d.LP * d.C1, idx->d_nLocal + q0, stride, k, nq, oI, oD,
                                     e0, idx->lev1))) return rc;
          usedOneLaunch = true;
        } else
        if ((rc = launchRerankSelect(idx, coarseLds, grid, emitRuns ? lRuns : (xcode ? lFusedX : lFused), st, idx->d_qL1virt + (size_t)q0 * d.LP * d.C1, idx->d_nLocal + q0,
                                     stride, k, nq, oI, oD, oP))) return rc;
        idx->poolDirty = false;
      }
      if (!leanEvents) PQT_REC(EV_RERANK);
    } else if (bigK) {
      const float* v = idx->d_qL1virt + (size_t)q0 * d.LP * d.C1;
      // short lists (n <= 1024) first: wave-per-query evaluate + sort (pqt_k_rerank_sort_small); it hands the queries with longer
      // lists to the block-wide select kernel through fbList
      constexpr int SNW = kSmallWaves;
      const bool smallCL = coarseLds && coarseBytes + (size_t)SNW * (d.LP * d.C1 * 4 + PQT_RSS_MAXN * 4) + 16 <= kMaxLds;
      // (the wavefronts' L1virt copies and key slots must fit the LDS: not at C1 = 128 with 32 line parts, BASELINE configs[4])
      const bool smallFirst = idx->smallLists && (d.LP == 16 || d.LP == 32) && (smallCL ? coarseBytes : 0) + (size_t)SNW * (d.LP * d.C1 * 4 + PQT_RSS_MAXN * 4) + 16 <= kMaxLds;
      usedSmallFirst = smallFirst;
      const uint32_t* bigQl = nullptr; const uint32_t* bigQc = nullptr;
      hipEvent_t bigEv0 = idx->lev0;
      bool padSideUsed = false;
      if (smallFirst) {
        const size_t lSmall = (smallCL ? coarseBytes : 0) + (size_t)SNW * (d.LP * d.C1 * 4 + PQT_RSS_MAXN * 4) + 16;
        HIPCHK(hipMemsetAsync(idx->d_fbCount, 0, 8, st));
        // the padding of the rows (k - n slots each: most of a 4096-slot row) is written on a side stream beside the sort kernels
        const bool padSide = idx->padSide;
        padSideUsed = padSide;
        if (padSide) {
          if (!idx->padStream) {
            HIPCHK(hipStreamCreateWithFlags(&idx->padStream, hipStreamNonBlocking));
            HIPCHK(hipEventCreateWithFlags(&idx->evPadFork, hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&idx->evPadJoin, hipEventDisableTiming));
          }
          HIPCHK(hipEventRecord(idx->evPadFork, st));  // the traversal's candidate counts are final
          HIPCHK(hipStreamWaitEvent(idx->padStream, idx->evPadFork, 0));
          hipLaunchKernelGGL(pqt_k_pad_rows, dim3(nq), dim3(256), 0, idx->padStream, idx->d_nLocal + q0, nq, k, oI, oD, oP);
          HIPCHK(hipEventRecord(idx->evPadJoin, idx->padStream));
        }
        PqtRsArgs sa{};
        sa.padDone = padSide ? 1u : 0u;
        sa.codes = idx->d_codesBin; sa.ids = idx->d_ids; sa.qL1virt = v; sa.coarse = idx->d_coarse; sa.cand = idx->d_cand; sa.candPos = idx->d_candPos;
        sa.nLocal = idx->d_nLocal + q0; sa.stride = stride; sa.k = k; sa.qn = nq; sa.prm = d; sa.outIdx = oI; sa.outDist = oD; sa.outPos = oP; sa.counters = idx->ctr; sa.dbg = idx->dbg & 15u; sa.nIds = idx->nIds;
        const uint32_t sgrid = std::min<uint32_t>((nq + SNW - 1) / SNW, (uint32_t)idx->numCUs);
        if (!(idx->dbg & 32768u)) { if ((rc = launchSmallLists(idx, smallCL, lSmall, sgrid, st, sa, idx->lev0))) return rc; }  // (debug bit: bisecting)
        bigQl = idx->d_fbList; bigQc = idx->d_fbCount;
        bigEv0 = nullptr;
        // lists of 1025..2048 candidates: second wave-per-query pass (8 wavefronts around 8 KB of key slots each), SIFT1M shape
        const size_t lMid = coarseBytes + (size_t)kMidWaves * (d.LP * d.C1 * 4 + 2048 * 4) + 16;
        if (smallCL && d.LP == 16 && lMid <= kMaxLds && idx->smallLists && !(idx->dbg & (32768u | 131072u))) {
          sa.qlist = idx->d_fbList; sa.qcount = idx->d_fbCount;
          if ((rc = launchMidLists(idx, lMid, (uint32_t)idx->numCUs, st, sa, idx->d_fbList + idx->qCap, idx->d_fbCount + 1))) return rc;
          bigQl = idx->d_fbList + idx->qCap; bigQc = idx->d_fbCount + 1;
          usedMid = true;
        }
      }
      if (!(smallFirst && (idx->dbg & 16384u))) {
      if ((rc = launchBigK(idx, bigCL, lBig, nq, st, v, idx->d_nLocal + q0, stride, k, kP2, kcap, oI, oD, oP, bigQl, bigQc, bigEv0, idx->lev1))) return rc;
      }
      if (padSideUsed) HIPCHK(hipStreamWaitEvent(st, idx->evPadJoin, 0));
      if (!leanEvents) PQT_REC(EV_RERANK);
    } else {
    if (d.LP % 4 == 0)
      hipLaunchKernelGGL(pqt_k_rerank<4>, dim3(nq), dim3(PQT_BLOCK), lRer, st, idx->d_codesBin,
                         idx->d_qL1virt + (size_t)q0 * d.LP * d.C1, idx->d_coarse, idx->d_cand, idx->d_candDist,
                         idx->d_nLocal + q0, stride, d);
    else
      hipLaunchKernelGGL(pqt_k_rerank<1>, dim3(nq), dim3(PQT_BLOCK), lRer, st, idx->d_codesBin,
                         idx->d_qL1virt + (size_t)q0 * d.LP * d.C1, idx->d_coarse, idx->d_cand, idx->d_candDist,
                         idx->d_nLocal + q0, stride, d);
    PQT_REC(EV_RERANK);
    if (fullSort) {
      if (idx->sharded)
        hipLaunchKernelGGL(pqt_k_fullsort<true>, dim3(nq), dim3(PQT_BLOCK), 0, st, idx->d_ids, idx->d_cand, idx->d_candDist, idx->d_candPos,
                           idx->d_nLocal + q0, stride, k, idx->d_sortKeys, sortP2, oI, oD, oP, idx->ctr);
      else
        hipLaunchKernelGGL(pqt_k_fullsort<false>, dim3(nq), dim3(PQT_BLOCK), 0, st, idx->d_ids, idx->d_cand, idx->d_candDist, idx->d_candPos,
                           idx->d_nLocal + q0, stride, k, idx->d_sortKeys, sortP2, oI, oD, oP, idx->ctr);
    } else {
      if (idx->sharded)
        hipLaunchKernelGGL(pqt_k_select<true>, dim3(nq), dim3(PQT_BLOCK), lSel, st, idx->d_ids, idx->d_cand, idx->d_candDist, idx->d_candPos,
                           idx->d_nLocal + q0, stride, k, kP2, oI, oD, oP, idx->ctr);
      else
        hipLaunchKernelGGL(pqt_k_select<false>, dim3(nq), dim3(PQT_BLOCK), lSel, st, idx->d_ids, idx->d_cand, idx->d_candDist, idx->d_candPos,
                           idx->d_nLocal + q0, stride, k, kP2, oI, oD, oP, idx->ctr);
    }
    }
    if (!leanEvents) PQT_REC(EV_SELECT);
  }
#undef PQT_REC
  if (!nextZeroed) HIPCHK(hipMemsetAsync(nextCtr, 0, 8 * sizeof(unsigned long long), st));
  idx->ctrPos = (idx->ctrPos + 1) % kCtrRing;
  if (outCount && !countDirect) HIPCHK(hipMemcpyAsync(outCount, idx->d_nCand, (size_t)qn * 4, hipMemcpyDeviceToDevice, st));
  HIPCHK(hipGetLastError());
  idx->lastQn = qn; idx->lastHe = He; idx->lastPieces = 0;
  idx->lastSegKept = !travFused || travWide;  // the fused traversal keeps the sorted part lists on chip unless He > 512
  idx->lastFilter = useFilter || usedSmallFirst;  // (pqt_stats.filter_fallbacks then counts the queries the short-list kernel handed to the block-wide one)
  idx->lastRuns = emitRuns;
  idx->lastShared = sharedPass;
  idx->lastCoop = coopPass;
  {
    // which kernels ran (pqt_get_last_path): tests assert the path, not only the result
    std::string tp = !travFused ? "traverse=staged" : (travWide ? "traverse=fused-wide" : "traverse=fused");
    if (binsIn) tp = "traverse=bins-resolved+" + tp.substr(9);
    if (travFused) {
      const bool twoOk = idx->d_heur4 && idx->d_filter && !(idx->dbg & (4096u | 2048u | 32u)) && !d.hashMod;
      const int shape = (idx->noShape || !twoOk) ? 0 : pqt_shape_of(d);
      tp += shape ? (shape == 1 ? "-shape1" : "-shape2") : (travP2 ? "-p2" : "-generic");
      if (idx->lastTravF1) tp += idx->useFilter1 == 2 ? "-f1c" : "-f1";  // pqt_k_traverse_f1 really ran (launchFusedTraversal falls back to the plain kernel when the first level does not exist or fit)
    }
    std::string rp;
    if (fused) {
      if (useBias) rp = std::string(useFilter ? "rerank=mode2" : "rerank=mode1") + (biasNW == 12 ? "-nw12" : "-nw6") + (runsBig ? "-runs" : "") + (sharedPass ? "-shared" : "") + (coopPass ? "-coop" : "");
      else if (wgG) rp = "rerank=wg-g" + std::to_string(wgG);
      else rp = std::string(coarseLds ? "rerank=lds-table" : "rerank=l2-table") + (emitRuns ? "-runs" : "") + ((xcode && !usedOneLaunch) ? "-xcode" : "");
    } else if (bigK) rp = std::string(bigCL ? "rerank=big-lds-table" : "rerank=big-l2-table") + (usedSmallFirst ? (usedMid ? "+small-lists+mid-lists" : "+small-lists") : "");
    else rp = fullSort ? "rerank=staged-fullsort" : "rerank=staged-select";
    idx->lastPath = tp + " " + rp + " chunks=" + std::to_string(nChunks) + (usedOneLaunch ? " one-launch" : "");
  }
  idx->lastDistKept = (!fused && !bigK) || sharedPass;  // the fused rerank kernels never write candDist; the shared-row pass leaves its filter distances d1 there
                                                        // (exact distances for the queries handed back): pqt_debug_read lets the tests compare evaluating kernels bit for bit
  if (sync) HIPCHK(hipStreamSynchronize(st));
  return PQT_OK;
}

}  // namespace

// =====================================================================================================
extern "C" {

const char* pqt_last_error(void) { return g_err.c_str(); }

int pqt_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  int ok = 0;
  for (int i = 0; i < n; ++i) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, i) == hipSuccess && strncmp(p.gcnArchName, "gfx950", 6) == 0) ++ok;
  }
  return ok;
}

int pqt_index_create(const pqt_params* prm, int device, pqt_index** out) {
  if (!prm || !out) return fail(PQT_ERR_INVALID, "null argument");
  const pqt_params& p = *prm;
  if (!p.dim || !p.p || !p.c1 || !p.c2 || !p.w || !p.lp || p.dim % p.p || p.dim % p.lp || p.lp % p.p || p.w > p.c1 ||
      p.c1 > 256 || p.c2 > 256 || p.p > PQT_MAXP || (uint64_t)p.w * p.c2 > 65536)
    return fail(PQT_ERR_INVALID, "invalid parameters: need dim%p==0, dim%lp==0, lp%p==0, 1<=w<=c1<=256, c2<=256, p<=8");
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail(PQT_ERR_DEVICE, "no such HIP device");
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(PQT_ERR_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
  pqt_index* idx = new pqt_index();
  idx->prm = p; idx->device = device;
  PqtDevParams& d = idx->dp;
  d.D = p.dim; d.P = p.p; d.C1 = p.c1; d.C2 = p.c2; d.W = p.w; d.LP = p.lp;
  d.S = p.dim / p.p; d.SS = p.dim / p.lp; d.R = p.lp / p.p; d.WC = p.w * p.c2; d.hashMod = 0;
  for (uint32_t i = 0; i < PQT_MAXP; ++i) d.powers[i] = i < p.p ? upow(p.c1 * p.c2, i) : 0;  // treequantizer.hpp:45-49
  idx->maxMultiIndex = upow(d.WC, p.p);  // treequantizer.hpp:40-41 (wraps in uint32 like the reference)
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&idx->stream, hipStreamNonBlocking) != hipSuccess ||
      hipMalloc((void**)&idx->d_counters, (8 * (kCtrRing + 1) * sizeof(unsigned long long) + kPoolRing * kPoolWords * sizeof(uint32_t))) != hipSuccess) {
    delete idx;
    return fail(PQT_ERR_DEVICE, "stream/counter allocation failed");
  }
  (void)hipMemset(idx->d_counters, 0, (8 * (kCtrRing + 1) * sizeof(unsigned long long) + kPoolRing * kPoolWords * sizeof(uint32_t)));
  size_t freeB = 0, totalB = 0;
  idx->numCUs = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  idx->forceUnfused = getenv("PQT_FORCE_UNFUSED") != nullptr;
  if (getenv("PQT_DBG")) idx->dbg = (uint32_t)atoi(getenv("PQT_DBG"));
  if (getenv("PQT_BALANCE")) idx->balance = std::max(-1, std::min(2, atoi(getenv("PQT_BALANCE"))));
  if (getenv("PQT_TSTAMP")) { if (hipMalloc((void**)&idx->d_tstamp, (size_t)(1 << 16) * PQT_TS_WORDS * 8) != hipSuccess) idx->d_tstamp = nullptr; }
  if (hipMemGetInfo(&freeB, &totalB) == hipSuccess) idx->scratchBudget = std::min<size_t>(idx->scratchBudget, totalB / 8);
  *out = idx;
  return PQT_OK;
}

void pqt_index_destroy(pqt_index* idx) {
  if (!idx) return;
  (void)hipSetDevice(idx->device);
  (void)hipDeviceSynchronize();
  for (auto& v : idx->views) if (v) { pqt_index_destroy(v); v = nullptr; }
  {  // user views die with their owner; a view destroyed by the caller leaves the owner's list
    std::vector<pqt_index*> uv;
    uv.swap(idx->userViews);
    for (pqt_index* v : uv) { v->owner = nullptr; pqt_index_destroy(v); }
    if (idx->userView && idx->owner) {
      auto& l = idx->owner->userViews;
      l.erase(std::remove(l.begin(), l.end(), idx), l.end());
    }
  }
  if (idx->evPadFork) (void)hipEventDestroy(idx->evPadFork);
  if (idx->evPadJoin) (void)hipEventDestroy(idx->evPadJoin);
  if (idx->padStream) (void)hipStreamDestroy(idx->padStream);
  if (idx->evFork) (void)hipEventDestroy(idx->evFork);
  for (auto& e : idx->evJoin) if (e) (void)hipEventDestroy(e);
  if (idx->isView) {  // the arrays of the index belong to the owner
    idx->d_cb1 = idx->d_cb2 = idx->d_coarse = idx->d_cb1L = idx->d_cb2T = nullptr; idx->d_heur = idx->d_heur8 = nullptr; idx->d_heur4 = nullptr; idx->d_seq2d = nullptr;
    idx->d_table = nullptr; idx->d_lower = idx->d_ids = idx->d_codes = idx->d_codesBin = idx->d_codesGrp = idx->d_codesX = idx->d_filter = idx->d_filter1 = nullptr; idx->d_bias = nullptr;
  }
  void* ptrs[] = {idx->d_cb1, idx->d_cb1L, idx->d_cb2, idx->d_cb2T, idx->d_coarse, idx->d_heur, idx->d_heur8, idx->d_heur4, idx->d_tstamp, idx->d_table, idx->d_filter, idx->d_lower, idx->d_ids,
                  idx->codesOwned ? idx->d_codes : nullptr, idx->d_codesBin, idx->d_codesGrp, idx->d_codesX, idx->d_bias, idx->d_qL1virt, idx->d_segD, idx->d_segBin, idx->d_cand,
                  idx->d_candDist, idx->d_candPos, idx->d_runs, idx->d_runGpos, idx->d_nRuns, idx->d_fbList, idx->d_fbCount, idx->d_tvList, idx->d_tvCount, idx->h2dQ, idx->h2dI, idx->h2dD, idx->h2dC, idx->d_nCand, idx->d_nLocal, idx->d_nIncl, idx->d_ovList, idx->d_ovCount, idx->d_sortKeys, idx->d_counters, idx->d_schedList, idx->d_seq2d, idx->d_heurQ, idx->d_srTable, idx->d_srPairs, idx->d_srBlocks, idx->d_srItems, idx->d_filter1, idx->d_srKeys, idx->d_srSeg, idx->d_coopErr};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  if (idx->evCreated) for (int r = 0; r < kRing; ++r) for (int c = 0; c < kMaxChunks; ++c) for (int e = 0; e < EV_COUNT; ++e) (void)hipEventDestroy(idx->evRing[r][c][e]);
  if (idx->stream) (void)hipStreamDestroy(idx->stream);
  delete idx;
}

int pqt_index_params(const pqt_index* idx, pqt_params* out) {
  if (!idx || !out) return fail(PQT_ERR_INVALID, "null argument");
  *out = idx->prm;
  return PQT_OK;
}

int pqt_index_create_view(pqt_index* owner, pqt_index** out) {
  if (!owner || !out) return fail(PQT_ERR_INVALID, "null argument");
  if (owner->isView) return fail(PQT_ERR_INVALID, "a view cannot be the owner of another view");
  pqt_index* v = nullptr;
  int rc = pqt_index_create(&owner->prm, owner->device, &v);
  if (rc) return rc;
  v->isView = true; v->userView = true; v->owner = owner;
  v->stageTiming = 0;  // per-kernel events only on request (pqt_index_set_option on the view)
  owner->userViews.push_back(v);
  *out = v;
  return PQT_OK;
}

int pqt_index_set_option(pqt_index* idx, const char* name, int64_t value) {
  if (!idx || !name) return fail(PQT_ERR_INVALID, "null argument");
  if (strcmp(name, "fused") == 0) { idx->forceUnfused = (value == 0); return PQT_OK; }
  if (strcmp(name, "wg_rerank") == 0) { idx->useWgRerank = (value != 0); return PQT_OK; }
  if (strcmp(name, "adc_bias") == 0) {
    // opt-in: ADC distance as sum_p (b + l*(a-b)) + bias[row] (SURVEY App. C "E-alt"): same candidate sets, distances
    // rounded differently from the reference's association (the default keeps the reference's)
    if (value && !(idx->dp.LP == 16 || idx->dp.LP == 32)) return fail(PQT_ERR_LIMIT, "adc_bias needs 16 or 32 line parts");
    if (value && (idx->dp.C1 & (idx->dp.C1 - 1))) return fail(PQT_ERR_LIMIT, "adc_bias needs a power-of-two C1");
    idx->adcBias = (value != 0);
    return PQT_OK;
  }
  // 1: the fused traversal hands the included bins to the MODE 0 rerank as (first position, first row) runs instead of
  // materialising the candidate list.  Measured r02 (SIFT1M shape): traversal 0.066 -> 0.057 ms, but the rerank's expansion
  // of the runs (uniform v_readlane walk or 7-step LDS search per 64 candidates) sits in front of every row request where
  // the prefetched list costs nothing: rerank+select 0.155 -> 0.170 ms.  Net loss, so the default stays 0.
  // 0: the exact rerank with the LDS table reads the plain bin-ordered store instead of its X-code copy (same results; A/B and tests)
  if (strcmp(name, "xcode") == 0) { idx->useXCode = value < 0 ? -1 : (value != 0); return PQT_OK; }
  // shared-row pass of the filtered rerank (pqt_shared_rows.h): -1 automatic (line stores of 1 GiB and more), 0 off, 1 on where the shape allows
  if (strcmp(name, "shared_rows") == 0) { idx->sharedRows = value < 0 ? -1 : (value != 0); return PQT_OK; }
  // evaluating kernel of the shared-row pass: 1 = pqt_k_sr_adc, 2 = pqt_k_sr_adc2 (tables of two queries interleaved, row decode hoisted; same bits)
  // tests / measurement: size of the pass's per-batch bin table (10..24 bits, 0 = automatic), probes before a pair gives up (1..128), and the
  // statistics launch behind the preparation (pqt_get_shared_rows_stats)
  if (strcmp(name, "sr_slot_bits") == 0) { if (value != 0 && (value < 10 || value > 24)) return fail(PQT_ERR_INVALID, "sr_slot_bits: 0 or 10..24"); idx->srSlotBits = (uint32_t)value; return PQT_OK; }
  if (strcmp(name, "sr_probes") == 0) { if (value < 1 || value > 128) return fail(PQT_ERR_INVALID, "sr_probes: 1..128"); idx->srProbes = (uint32_t)value; return PQT_OK; }
  if (strcmp(name, "sr_stats") == 0) { idx->srStats = value != 0; if (!idx->srStats) idx->srStatPtr = nullptr; return PQT_OK; }
  // the selection's scan over the pass's distances: position ranges per query (1 = one wavefront per query, the default kernel; 2 | 4) and
  // 16-byte requests in flight per lane (4 | 8)
  if (strcmp(name, "sr_scan_split") == 0) { if (value != 1 && value != 2 && value != 4) return fail(PQT_ERR_INVALID, "sr_scan_split: 1, 2 or 4"); idx->srScanSplit = value; return PQT_OK; }
  if (strcmp(name, "sr_scan_depth") == 0) { if (value != 4 && value != 8) return fail(PQT_ERR_INVALID, "sr_scan_depth: 4 or 8"); idx->srScanDepth = value; return PQT_OK; }
  // cooperative filter scan of the filtered rerank (configs[2]/[3] shape, where the shared-row pass does not run): 0 off (default), 1 on
  if (strcmp(name, "coop_rerank") == 0) { idx->coopRerank = value != 0; return PQT_OK; }
  if (strcmp(name, "sr_kernel") == 0) { if (value != 1 && value != 2) return fail(PQT_ERR_INVALID, "sr_kernel: 1 or 2"); idx->srKernel = value; return PQT_OK; }
  // first level of the presence bitmap in LDS for the wide enumeration (512 < bound_bins <= 4096, pqt_k_traverse_f1): 1 = on where it exists,
  // 0 / -1 (default) off.  MEASURED AND NOT THE DEFAULT (scripts/r05_wide_ab.py, SIFT1M shape): (4096, 4096) traversal 0.205 -> 0.197 ms,
  // (20000, 2048) 0.123 -> 0.147, (4096, 1024) 0.085 -> 0.116; 100 M: no change.  A query's enumeration takes half the clocks (133 k -> 72 k
  // per query) but 8 wavefronts per CU behind the 64 KB copy instead of 15 answer exactly as many queries per clock: the wide mode is not
  // bound by the probe rate of the 512 KB bitmap
  // (2, round 6: the rows that pass the first level are compacted and the bitmap is asked for full wavefronts of them only)
  if (strcmp(name, "filter_l1") == 0) { idx->useFilter1 = value < 0 ? -1 : (value >= 2 ? 2 : (value != 0)); return PQT_OK; }
  if (strcmp(name, "bin_runs") == 0) { idx->useRuns = value < 0 ? -1 : (value != 0); return PQT_OK; }
  if (strcmp(name, "overlap") == 0) { idx->overlap = value < 0 ? -1 : (int)std::min<int64_t>(value, pqt_index::kMaxViews + 1); return PQT_OK; }  // batch pieces on their own streams: 0 / -1 (default) never, 1 = two pieces whenever possible, 2..4 = that many pieces
  if (strcmp(name, "one_launch") == 0) { idx->oneLaunch = value < 0 ? -1 : (value != 0); return PQT_OK; }  // SIFT1M shape: traversal + rerank of a query by one wavefront in one launch (opt-in: measured slower)
  if (strcmp(name, "pad_side_stream") == 0) { idx->padSide = (value != 0); return PQT_OK; }  // 0: the k > 128 sort kernels write the padding of their rows themselves
  if (strcmp(name, "small_lists") == 0) { idx->smallLists = (value != 0); return PQT_OK; }  // 0: every query of a 128 < k <= 4096 call through the block-wide select kernel
  if (strcmp(name, "exact_filter") == 0) { idx->exactFilter = (value != 0); return PQT_OK; }  // 0: workgroup-per-query exact kernel for big coarse tables
  if (strcmp(name, "static_shapes") == 0) { idx->noShape = (value == 0); return PQT_OK; }  // 0: run-time-shape traversal even on the BASELINE shapes
  // per-kernel start/stop events (they cost ~5 us per kernel launch): 1 = every call (default), N = every N-th call, 0 = never;
  // pqt_get_stage_ms_history reports the timed calls only
  if (strcmp(name, "stage_timing") == 0) { idx->stageTiming = value < 0 ? 0 : (int)std::min<int64_t>(value, 1 << 20); idx->timingPhase = 0; return PQT_OK; }
  if (strcmp(name, "balance") == 0) { idx->balance = value < 0 ? -1 : (value >= 2 ? 2 : (int)value); return PQT_OK; }  // rerank schedule: -1 automatic, 0 static, 1 workgroup-local, 2 global pools
  if (strcmp(name, "debug_bits") == 0) { idx->dbg = (uint32_t)value; return PQT_OK; }  // ablation switches (PQT_DBG), wrong results
  if (strcmp(name, "order_all_rows") == 0) { idx->dbg = value ? (idx->dbg | 32u) : (idx->dbg & ~32u); return PQT_OK; }
  // test switch: the traversal's part lists always through the one-list-at-a-time code that settles near-ties of the second-level distances
  // exactly (compile-time shapes: normally only the queries whose row-parallel sort meets such a pair take it); results unchanged
  if (strcmp(name, "exact_part_sorts") == 0) { idx->dbg = value ? (idx->dbg | 64u) : (idx->dbg & ~64u); return PQT_OK; }
  // "enumerate_beyond_wrap" = 1: the number of enumerable heuristic rows is the true (W*C2)^P instead of the reference's uint32 product
  // (treequantizer.hpp:40-41), which wraps to 0 at BASELINE configs[4] (64^8 = 2^48) and makes its orderBins enumerate nothing.  NO
  // reference counterpart: a throughput-only mode for that shape, used with a supplied prefix (pqt_index_set_heuristic).  Bin ids keep
  // the reference's uint32 wrap-around.  Default 0 = the reference's behaviour.
  if (strcmp(name, "enumerate_beyond_wrap") == 0) {
    if (idx->isView) return fail(PQT_ERR_INVALID, "set enumerate_beyond_wrap on the owner");
    uint64_t full = 1;
    for (uint32_t i = 0; i < idx->dp.P; ++i) { full *= idx->dp.WC; if (full > ((uint64_t)1 << 62) / std::max<uint32_t>(idx->dp.WC, 1)) { full = (uint64_t)1 << 62; break; } }
    idx->maxMultiIndex = value ? full : (uint64_t)upow(idx->dp.WC, idx->dp.P);
    return PQT_OK;
  }
  if (strcmp(name, "scratch_mb") == 0) { if (value < 1) return fail(PQT_ERR_INVALID, "scratch_mb must be >= 1"); idx->scratchBudget = (size_t)value << 20; return PQT_OK; }
  return fail(PQT_ERR_INVALID, std::string("unknown option ") + name);
}

int pqt_index_set_codebooks(pqt_index* idx, const float* cb1, const float* cb2) {
  if (idx && idx->isView) return fail(PQT_ERR_INVALID, "pqt_index_set_codebooks: a view handle shares the owner's index; load into the owner");
  if (!idx || !cb1 || !cb2) return fail(PQT_ERR_INVALID, "null argument");
  int rc = setDevice(idx);
  if (rc) return rc;
  const PqtDevParams& d = idx->dp;
  const size_t n1 = (size_t)d.C1 * d.D, n2 = (size_t)d.P * d.C1 * d.C2 * d.S, nc = (size_t)d.LP * d.C1 * d.C1;
  if ((rc = devAlloc(&idx->d_cb1, n1))) return rc;
  if ((rc = devAlloc(&idx->d_cb2, n2))) return rc;
  if ((rc = devAlloc(&idx->d_coarse, nc))) return rc;
  HIPCHK(hipMemcpy(idx->d_cb1, cb1, n1 * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(idx->d_cb2, cb2, n2 * 4, hipMemcpyHostToDevice));
  {  // line-part-major copy [lp][c][SS] for the table kernel of the bins-in path (pqt_l1virt_block)
    std::vector<float> t(n1);
    for (uint32_t lp = 0; lp < d.LP; ++lp)
      for (uint32_t c = 0; c < d.C1; ++c)
        for (uint32_t e = 0; e < d.SS; ++e) t[((size_t)lp * d.C1 + c) * d.SS + e] = cb1[(size_t)c * d.D + lp * d.SS + e];
    if ((rc = devAlloc(&idx->d_cb1L, n1))) return rc;
    HIPCHK(hipMemcpy(idx->d_cb1L, t.data(), n1 * 4, hipMemcpyHostToDevice));
  }
  if (idx->d_cb2T) { (void)hipFree(idx->d_cb2T); idx->d_cb2T = nullptr; }
  if (d.S % 4 == 0) {
    std::vector<float> t(n2);
    const uint32_t V = d.S / 4;
    for (size_t cell = 0; cell < (size_t)d.P * d.C1; ++cell)
      for (uint32_t h = 0; h < d.C2; ++h)
        for (uint32_t v = 0; v < V; ++v)
          for (uint32_t e = 0; e < 4; ++e) t[(cell * V + v) * d.C2 * 4 + (size_t)h * 4 + e] = cb2[(cell * d.C2 + h) * d.S + v * 4 + e];
    if ((rc = devAlloc(&idx->d_cb2T, n2))) return rc;
    HIPCHK(hipMemcpy(idx->d_cb2T, t.data(), n2 * 4, hipMemcpyHostToDevice));
  }
  hipLaunchKernelGGL(pqt_k_coarse, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, idx->stream, idx->d_cb1, idx->d_coarse, d);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(idx->stream));
  {  // largest coarse entry: bound of the MODE 2 error band (pqt_rs_query)
    std::vector<float> hc(nc);
    HIPCHK(hipMemcpy(hc.data(), idx->d_coarse, nc * 4, hipMemcpyDeviceToHost));
    float m = 0.f;
    for (float v : hc) if (v > m) m = v;  // NaN entries (none for finite codebooks) are skipped
    idx->coarseMax = m;
  }
  idx->biasReady = false;
  idx->haveTree = true;
  return PQT_OK;
}

int pqt_index_get_coarse(const pqt_index* idx, float* out) {
  if (!idx || !out) return fail(PQT_ERR_INVALID, "null argument");
  if (!idx->haveTree) return fail(PQT_ERR_STATE, "no codebooks");
  int rc = setDevice(idx);
  if (rc) return rc;
  HIPCHK(hipMemcpy(out, idx->d_coarse, (size_t)idx->dp.LP * idx->dp.C1 * idx->dp.C1 * 4, hipMemcpyDeviceToHost));
  return PQT_OK;
}

static int uploadHeuristic(pqt_index* idx) {
  int rc = setDevice(idx);
  if (rc) return rc;
  idx->heur2d = false;  // a shared table replaces the per-query 2-D rows (pqt_index_build_heuristic_2d sets the flag again after this call)
  const uint32_t P = idx->dp.P;
  std::vector<uint16_t> h16(idx->heurRows * P);
  for (size_t i = 0; i < h16.size(); ++i) {
    if (idx->heurHost[i] >= idx->dp.WC) return fail(PQT_ERR_INVALID, "heuristic digit out of range (must be < W*C2)");
    h16[i] = (uint16_t)idx->heurHost[i];
  }
  if ((rc = devAlloc(&idx->d_heur, h16.size()))) return rc;
  if (!h16.empty()) HIPCHK(hipMemcpy(idx->d_heur, h16.data(), h16.size() * 2, hipMemcpyHostToDevice));
  // 16-byte rows (8 x u16, zero padded) for the one-read-per-row traversal kernel
  std::vector<uint16_t> h8(idx->heurRows * 8, 0);
  for (uint64_t r = 0; r < idx->heurRows; ++r) for (uint32_t pp = 0; pp < P; ++pp) h8[r * 8 + pp] = h16[r * P + pp];
  if ((rc = devAlloc(&idx->d_heur8, h8.size()))) return rc;
  if (!h8.empty()) HIPCHK(hipMemcpy(idx->d_heur8, h8.data(), h8.size() * 2, hipMemcpyHostToDevice));
  // packed rows (4 x u8 in one dword) for P <= 4 and digits < 256: the traversal requests them a block ahead
  if (idx->d_heur4) { (void)hipFree(idx->d_heur4); idx->d_heur4 = nullptr; }
  if (P <= 4 && idx->dp.WC <= 256) {
    std::vector<uint32_t> h4(std::max<uint64_t>(idx->heurRows, 1), 0u);
    for (uint64_t r = 0; r < idx->heurRows; ++r) for (uint32_t pp = 0; pp < P; ++pp) h4[r] |= (uint32_t)h16[r * P + pp] << (8 * pp);
    if ((rc = devAlloc(&idx->d_heur4, h4.size()))) return rc;
    HIPCHK(hipMemcpy(idx->d_heur4, h4.data(), h4.size() * 4, hipMemcpyHostToDevice));
  }
  return PQT_OK;
}

// prepareHeuristic (treequantizer.hpp:75-127): every tuple of {0..W*C2-1}^P (digit p of idx in base W*C2),
// ordered by squared norm with std::sort and the reference's comparator, so the order of equal norms is the
// one the reference's own build (same libstdc++ introsort) produces.  Only `rows` rows are kept.
int pqt_index_build_heuristic(pqt_index* idx, uint64_t rows) {
  if (idx && idx->isView) return fail(PQT_ERR_INVALID, "pqt_index_build_heuristic: a view handle shares the owner's index; load into the owner");
  if (!idx) return fail(PQT_ERR_INVALID, "null argument");
  const uint64_t M = idx->maxMultiIndex;
  if (M == 0) {
    // (W*C2)^P wraps to 0 in the reference's uint arithmetic (treequantizer.hpp:40-41, helper.hpp:19-22; BASELINE configs[4]:
    // 64^8 = 2^48): its orderBins loop runs min(boundBins, 0) = 0 times, every query returns an empty candidate list.
    // Same here: an empty heuristic, queries enumerate no rows.
    idx->heurHost.clear();
    idx->heurRows = 0;
    return uploadHeuristic(idx);
  }
  if (M > ((uint64_t)1 << 28)) return fail(PQT_ERR_LIMIT, "(W*C2)^P > 2^28 tuples: supply a prefix with pqt_index_set_heuristic");
  const uint32_t base = idx->dp.WC, P = idx->dp.P;
  std::vector<float> norm(M);
  std::vector<uint32_t> order(M);
  for (uint64_t i = 0; i < M; ++i) {
    uint32_t dec = (uint32_t)i;
    float dig[PQT_MAXP] = {0};
    uint32_t p = 0;
    while (dec > 0) { dig[p] = (float)(dec % base); dec /= base; ++p; }
    float s = 0.f;
    for (uint32_t j = 0; j < P; ++j) s += dig[j] * dig[j];
    norm[i] = s; order[i] = (uint32_t)i;
  }
  const float* nptr = norm.data();
  std::sort(order.begin(), order.end(), [nptr](const uint32_t& l, const uint32_t& r) { return nptr[l] < nptr[r]; });
  rows = std::min<uint64_t>(rows, M);
  idx->heurHost.assign(rows * P, 0);
  for (uint64_t h = 0; h < rows; ++h) {
    uint32_t dec = order[h], p = 0;
    while (dec > 0) { idx->heurHost[h * P + p] = dec % base; dec /= base; ++p; }
  }
  idx->heurRows = rows;
  return uploadHeuristic(idx);
}

// Optional mode ("next" row 8f-4 tail): the CUDA library's traversal heuristic, ProTree::prepareDistSequence
// (pqt/ProTree.cu:128-207): digits in base b = min(16, max_cluster), all b^P tuples (digit p = i / b^p % b), key =
// sum_p sqrt(digit) accumulated in f32 in part order, std::sort of (key, i) pairs (ties by tuple index: deterministic),
// at most NUM_DISTSEQ = 65536 entries kept.  Only the ORDER of enumeration changes; bin ids, the cut and the rerank keep
// cpu_version semantics, so results under this table are not the reference CPU path's (nor bit-comparable with the CUDA
// path, whose bin-id digit order differs: SURVEY 8a "divergences").
int pqt_index_build_heuristic_cuda(pqt_index* idx, uint32_t max_cluster, uint64_t rows) {
  if (idx && idx->isView) return fail(PQT_ERR_INVALID, "pqt_index_build_heuristic_cuda: a view handle shares the owner's index; load into the owner");
  if (!idx) return fail(PQT_ERR_INVALID, "null argument");
  const uint32_t P = idx->dp.P;
  uint32_t b = std::min<uint32_t>(std::min<uint32_t>(max_cluster, 16u), idx->dp.WC);
  if (b == 0) return fail(PQT_ERR_INVALID, "max_cluster must be > 0");
  uint64_t nVec = 1;
  for (uint32_t p = 0; p < P; ++p) { nVec *= b; if (nVec > (1ull << 26)) return fail(PQT_ERR_LIMIT, "min(16, max_cluster)^P tuples exceed 2^26 (the CUDA library builds them all, 16 bytes each, before keeping 65536)"); }
  std::vector<std::pair<float, uint32_t> > dists((size_t)nVec);
  std::vector<uint64_t> denom(P, 1);
  for (uint32_t p = 1; p < P; ++p) denom[p] = denom[p - 1] * b;
  for (uint64_t i = 0; i < nVec; ++i) {
    float dist = 0.f;
    for (uint32_t p = 0; p < P; ++p) dist += sqrtf((float)((i / denom[p]) % b));
    dists[(size_t)i] = std::make_pair(dist, (uint32_t)i);
  }
  std::sort(dists.begin(), dists.end());
  rows = std::min<uint64_t>(std::min<uint64_t>(rows, nVec), 65536);
  idx->heurHost.assign(rows * P, 0);
  for (uint64_t h = 0; h < rows; ++h)
    for (uint32_t p = 0; p < P; ++p) idx->heurHost[h * P + p] = (uint32_t)((dists[(size_t)h].second / denom[p]) % b);
  idx->heurRows = rows;
  return uploadHeuristic(idx);
}

// Optional mode ("next" row 8f-4 tail): the CUDA library's 2-D anisotropic sequences, ProTree::prepare2DDistSequence(maxCluster)
// (pqt/ProTree.cu:50-126): for each of NUM_ANISO_DIR = 10 slopes s = (0.9 * 1.2f)^(slope - 5) (double power, rounded to f32) all
// maxCluster^2 cells i (x = i % maxCluster, y = i / maxCluster) keyed x^0.8 + s * y^0.8 in f32 (powf), std::sort of (key, i)
// pairs, the first NUM_DISTSEQ = 65536 cells kept (zero filled behind a shorter list).  The per-query use is pqt_k_rows_2d.
int pqt_index_build_heuristic_2d(pqt_index* idx, uint32_t max_cluster) {
  if (idx && idx->isView) return fail(PQT_ERR_INVALID, "pqt_index_build_heuristic_2d: a view handle shares the owner's index; load into the owner");
  if (!idx) return fail(PQT_ERR_INVALID, "null argument");
  if (idx->dp.P != 4) return fail(PQT_ERR_LIMIT, "the 2-D sequences merge parts (0,1) and (2,3): p = 4 only (pqt/PerturbationProTree.cu:2914-3100)");
  // (pqt_k_rows_2d reads the first 256 cells of an order and its slope samples at positions 44 / 45: a grid below 16 x 16 would put the
  // zero-filled tail -- cell (0, 0), the minimum -- on top of both pair lists)
  if (max_cluster < 16 || max_cluster > 4096) return fail(PQT_ERR_INVALID, "max_cluster must be in [16, 4096] (test/test1B.cpp:941 passes 512)");
  constexpr uint32_t kSeq = 65536, kDir = 10;
  const uint32_t nVec = max_cluster * max_cluster;
  std::vector<uint32_t> seq((size_t)kSeq * kDir, 0u);
  std::vector<std::pair<float, uint32_t> > dists(nVec);
  for (uint32_t slope = 0; slope < kDir; ++slope) {
    const float s = (float)std::pow(0.9 * (double)1.2f, (double)((int)slope - (int)(kDir / 2)));
    for (uint32_t i = 0; i < nVec; ++i) {
      const float x = (float)(i % max_cluster), y = (float)(i / max_cluster);
      const float n = 0.8f;
      dists[i] = std::make_pair(powf(x, n) + s * powf(y, n), i);
    }
    std::sort(dists.begin(), dists.end());
    const uint32_t keep = std::min<uint32_t>(nVec, kSeq);
    for (uint32_t i = 0; i < keep; ++i) seq[(size_t)slope * kSeq + i] = dists[i].second;
  }
  int rc = setDevice(idx);
  if (rc) return rc;
  // (queries check for a heuristic through d_heur / heurRows: one all-zero row stands in for the shared table)
  idx->heurHost.assign(idx->dp.P, 0u);
  idx->heurRows = 1;
  if ((rc = uploadHeuristic(idx))) return rc;
  if ((rc = devAlloc(&idx->d_seq2d, seq.size()))) return rc;
  HIPCHK(hipMemcpy(idx->d_seq2d, seq.data(), seq.size() * 4, hipMemcpyHostToDevice));
  idx->seq2dDc = max_cluster;
  // slope index = number of boundaries 1.2^(j - 4.5), j = 0 .. 8, at or below the slope  (== roundf(log_1.2(slope)) + 5 clamped to [0, 9])
  for (int j = 0; j < 9; ++j) idx->slopeThr[j] = powf(1.2f, (float)j - 4.5f);
  idx->heurRows = std::min<uint64_t>(kSeq, nVec);
  idx->heur2d = true;
  return PQT_OK;
}

int pqt_index_set_heuristic(pqt_index* idx, const uint32_t* tuples, uint64_t rows) {
  if (idx && idx->isView) return fail(PQT_ERR_INVALID, "pqt_index_set_heuristic: a view handle shares the owner's index; load into the owner");
  if (!idx || (!tuples && rows)) return fail(PQT_ERR_INVALID, "null argument");
  idx->heurHost.assign(tuples, tuples + rows * idx->dp.P);
  idx->heurRows = rows;
  return uploadHeuristic(idx);
}

int pqt_index_get_heuristic(const pqt_index* idx, uint32_t* out, uint64_t rows) {
  if (!idx || !out) return fail(PQT_ERR_INVALID, "null argument");
  rows = std::min<uint64_t>(rows, idx->heurRows);
  memcpy(out, idx->heurHost.data(), rows * idx->dp.P * 4);
  return PQT_OK;
}

int pqt_index_set_bins(pqt_index* idx, uint64_t nbins, const uint32_t* ids, const uint32_t* sizes, const uint32_t* members) {
  if (idx && idx->isView) return fail(PQT_ERR_INVALID, "pqt_index_set_bins: a view handle shares the owner's index; load into the owner");
  if (!idx || (nbins && (!ids || !sizes || !members))) return fail(PQT_ERR_INVALID, "null argument");
  std::vector<BinDesc> bins(nbins);
  uint64_t off = 0;
  for (uint64_t b = 0; b < nbins; ++b) {
    if (off > 0xffffffffull) return fail(PQT_ERR_LIMIT, "more than 2^32 members");
    bins[b] = {ids[b], sizes[b], (uint32_t)off, sizes[b], 0};
    off += sizes[b];
  }
  std::vector<uint32_t> local(members, members + off);
  idx->dp.hashMod = 0;
  idx->nTotal = off;
  return uploadBins(idx, bins, local, false);
}

int pqt_index_set_bins_shard(pqt_index* idx, uint64_t nbins, const uint32_t* ids, const uint32_t* sizes, const uint32_t* members,
                             uint32_t id_lo, uint32_t id_hi) {
  if (idx && idx->isView) return fail(PQT_ERR_INVALID, "pqt_index_set_bins_shard: a view handle shares the owner's index; load into the owner");
  if (!idx || (nbins && (!ids || !sizes || !members))) return fail(PQT_ERR_INVALID, "null argument");
  std::vector<BinDesc> bins(nbins);
  std::vector<uint32_t> local;
  uint64_t off = 0;
  for (uint64_t b = 0; b < nbins; ++b) {
    // the local members of a bin must be one contiguous run of its member list (true for id-range shards of the
    // reference's insertion-ordered lists): lower = members before the run
    uint32_t lower = 0, cnt = 0; bool started = false, ended = false;
    const uint32_t lstart = (uint32_t)local.size();
    for (uint32_t j = 0; j < sizes[b]; ++j) {
      const uint32_t v = members[off + j];
      const bool in = v >= id_lo && v < id_hi;
      if (in) {
        if (ended) return fail(PQT_ERR_INVALID, "shard members of a bin are not contiguous in its member list");
        if (!started) { started = true; lower = j; }
        local.push_back(v); ++cnt;
      } else if (started) ended = true;
    }
    bins[b] = {ids[b], sizes[b], lstart, cnt, lower};
    off += sizes[b];
  }
  idx->dp.hashMod = 0;
  idx->nTotal = off;
  return uploadBins(idx, bins, local, true);
}

int pqt_index_set_bins_local(pqt_index* idx, uint64_t nbins, const uint32_t* ids, const uint32_t* gsizes, const uint32_t* lower,
                             const uint32_t* lsizes, const uint32_t* members, uint64_t n_total) {
  if (idx && idx->isView) return fail(PQT_ERR_INVALID, "pqt_index_set_bins_local: a view handle shares the owner's index; load into the owner");
  if (!idx || (nbins && (!ids || !gsizes || !lower || !lsizes)) ) return fail(PQT_ERR_INVALID, "null argument");
  std::vector<BinDesc> bins(nbins);
  uint64_t off = 0;
  for (uint64_t b = 0; b < nbins; ++b) {
    if ((uint64_t)lower[b] + lsizes[b] > gsizes[b]) return fail(PQT_ERR_INVALID, "lower + local members exceed the bin's global population");
    if (off > 0xffffffffull) return fail(PQT_ERR_LIMIT, "more than 2^32 local members");
    bins[b] = {ids[b], gsizes[b], (uint32_t)off, lsizes[b], lower[b]};
    off += lsizes[b];
  }
  if (off && !members) return fail(PQT_ERR_INVALID, "null argument");
  std::vector<uint32_t> local(members, members + off);
  idx->dp.hashMod = 0;
  idx->nTotal = n_total;
  return uploadBins(idx, bins, local, true);
}

int pqt_index_set_db_hashed(pqt_index* idx, uint32_t n, const uint32_t* prefix, const uint32_t* counts, const uint32_t* dbidx,
                            uint32_t hash_size) {
  if (idx && idx->isView) return fail(PQT_ERR_INVALID, "pqt_index_set_db_hashed: a view handle shares the owner's index; load into the owner");
  if (!idx || !prefix || !counts || !dbidx || !hash_size) return fail(PQT_ERR_INVALID, "null argument");
  std::vector<BinDesc> bins;
  for (uint32_t s = 0; s < hash_size; ++s)
    if (counts[s]) {
      if ((uint64_t)prefix[s] + counts[s] > n) return fail(PQT_ERR_INVALID, "prefix/count outside dbIdx");
      bins.push_back({s, counts[s], prefix[s], counts[s], 0});
    }
  std::vector<uint32_t> local(dbidx, dbidx + n);
  idx->dp.hashMod = hash_size;
  idx->nTotal = n;
  return uploadBins(idx, bins, local, false);
}

int pqt_index_set_lines_host(pqt_index* idx, const uint32_t* codes, uint64_t nvec, uint64_t id_base) {
  if (idx && idx->isView) return fail(PQT_ERR_INVALID, "pqt_index_set_lines_host: a view handle shares the owner's index; load into the owner");
  if (!idx || (!codes && nvec)) return fail(PQT_ERR_INVALID, "null argument");
  int rc = setDevice(idx);
  if (rc) return rc;
  if (idx->codesOwned && idx->d_codes) (void)hipFree(idx->d_codes);
  idx->d_codes = nullptr;
  if ((rc = devAlloc(&idx->d_codes, (size_t)nvec * idx->dp.LP))) return rc;
  idx->codesOwned = true;
  if (nvec) HIPCHK(hipMemcpy(idx->d_codes, codes, (size_t)nvec * idx->dp.LP * 4, hipMemcpyHostToDevice));
  idx->nCodes = nvec; idx->idBase = id_base; idx->binOrdered = false; idx->linesDropped = false;
  return PQT_OK;
}

int pqt_index_set_lines_dev(pqt_index* idx, const uint32_t* codes_dev, uint64_t nvec, uint64_t id_base) {
  if (idx && idx->isView) return fail(PQT_ERR_INVALID, "pqt_index_set_lines_dev: a view handle shares the owner's index; load into the owner");
  if (!idx || !codes_dev) return fail(PQT_ERR_INVALID, "null argument");
  if (((uintptr_t)codes_dev & 15) != 0) return fail(PQT_ERR_INVALID, "line-code buffer must be 16-byte aligned");
  if (idx->codesOwned && idx->d_codes) { (void)hipSetDevice(idx->device); (void)hipFree(idx->d_codes); }
  idx->d_codes = const_cast<uint32_t*>(codes_dev);
  idx->codesOwned = false; idx->nCodes = nvec; idx->idBase = id_base; idx->binOrdered = false; idx->linesDropped = false;
  return PQT_OK;
}

int pqt_build_assign_encode(pqt_index* idx, const float* vecs_dev, uint64_t n, uint32_t* out_bin, uint32_t* out_codes, void* stream) {
  if (!idx || !vecs_dev || !out_bin || !out_codes) return fail(PQT_ERR_INVALID, "null argument");
  if (!idx->haveTree) return fail(PQT_ERR_STATE, "no codebooks");
  if (idx->dp.C1 < 2) return fail(PQT_ERR_INVALID, "line encoding needs C1 >= 2");
  int rc = setDevice(idx);
  if (rc) return rc;
  hipStream_t st = stream ? (hipStream_t)stream : idx->stream;
  const size_t lds = ldsEncode(idx->dp);
  if ((rc = allowLds(pqt_k_assign_encode, lds))) return rc;
  const uint64_t maxGrid = 1u << 30;
  for (uint64_t v0 = 0; v0 < n; v0 += maxGrid) {
    const uint32_t nb = (uint32_t)std::min<uint64_t>(maxGrid, n - v0);
    hipLaunchKernelGGL(pqt_k_assign_encode, dim3(nb), dim3(PQT_BLOCK), lds, st, vecs_dev + v0 * idx->dp.D, idx->d_cb1, idx->d_cb2,
                       idx->d_coarse, idx->dp, out_bin + v0, out_codes + v0 * idx->dp.LP, (idx->dbg >> 13) & 1u);
  }
  HIPCHK(hipGetLastError());
  if (!stream) HIPCHK(hipStreamSynchronize(st));
  return PQT_OK;
}

int pqt_rerank_exact(pqt_index* idx, const float* q_dev, uint32_t qn, uint32_t k, const uint32_t* in_idx_dev, const void* raw_dev,
                     int raw_is_u8, uint64_t raw_id_base, uint64_t raw_rows, uint32_t* out_idx_dev, float* out_dist_dev,
                     void* stream, int sync) {
  if (!idx || !q_dev || !in_idx_dev || !raw_dev || !out_idx_dev || !out_dist_dev) return fail(PQT_ERR_INVALID, "null argument");
  if (k == 0 || k > 512) return fail(PQT_ERR_LIMIT, "exact re-rank supports 1 <= k <= 512");
  if (in_idx_dev == out_idx_dev) return fail(PQT_ERR_INVALID, "in place re-rank is not supported");
  int rc = setDevice(idx);
  if (rc) return rc;
  hipStream_t st = stream ? (hipStream_t)stream : idx->stream;
  constexpr int NW = 4;
  const uint32_t D = idx->dp.D;
  const size_t lds = (size_t)NW * D * 4;
  const dim3 grid((qn + NW - 1) / NW), block(NW * 64);
#define PQT_LAUNCH_EX(RR)                                                                                              \
  do { if (raw_is_u8) hipLaunchKernelGGL((pqt_k_rerank_exact<NW, RR, true>), grid, block, lds, st, q_dev, qn, D, k, in_idx_dev, raw_dev, \
                                         raw_id_base, raw_rows, out_idx_dev, out_dist_dev);                            \
       else hipLaunchKernelGGL((pqt_k_rerank_exact<NW, RR, false>), grid, block, lds, st, q_dev, qn, D, k, in_idx_dev, raw_dev,          \
                               raw_id_base, raw_rows, out_idx_dev, out_dist_dev); } while (0)
  if (qn) { if (k <= 64) PQT_LAUNCH_EX(1); else if (k <= 128) PQT_LAUNCH_EX(2); else if (k <= 256) PQT_LAUNCH_EX(4); else PQT_LAUNCH_EX(8); }
#undef PQT_LAUNCH_EX
  HIPCHK(hipGetLastError());
  if (sync) HIPCHK(hipStreamSynchronize(st));
  return PQT_OK;
}

int pqt_kmeans_assign(int device, const float* x_dev, uint64_t n, uint32_t dim, uint32_t ld, const uint32_t* rows_dev,
                       const float* cen_dev, uint32_t ncen, uint32_t cen_ld, uint32_t* out_assign_dev, float* out_dist_dev,
                       void* stream) {
  if (!x_dev || !cen_dev || !out_assign_dev || !out_dist_dev || !dim || !ncen) return fail(PQT_ERR_INVALID, "null argument");
  HIPCHK(hipSetDevice(device));
  const size_t lds = (size_t)ncen * dim * 4;
  int rc = allowLds(pqt_k_kmeans_assign, lds);
  if (rc) return rc;
  if (n) hipLaunchKernelGGL(pqt_k_kmeans_assign, dim3((unsigned)((n + 255) / 256)), dim3(256), lds, (hipStream_t)stream, x_dev, n, dim, ld,
                            rows_dev, cen_dev, ncen, cen_ld, out_assign_dev, out_dist_dev);
  HIPCHK(hipGetLastError());
  if (!stream) HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  return PQT_OK;
}


// ---- overlapped halves ------------------------------------------------------------------------------------------------
// Everything a query call READS of an index (arrays, their sizes, flags, options); a view handle carries copies of these
// words and owns only scratch.  Compared before/after the first half to see whether that call built something lazily
// (bin-ordered store, group-major copy, row bias): the view's stream must then wait for it.
namespace {
struct SharedWords {
  PqtDevParams dp; pqt_params prm;
  const void* p[20]; uint64_t u[8]; uint32_t w[8]; float f[12]; int i[12]; bool b[16];
};
void captureShared(const pqt_index* x, SharedWords& v) {
  memset(&v, 0, sizeof(v));
  v.dp = x->dp; v.prm = x->prm;
  const void* ps[] = {x->d_cb1, x->d_cb2, x->d_coarse, x->d_cb1L, x->d_cb2T, x->d_heur, x->d_heur8, x->d_heur4, x->d_table, x->d_lower, x->d_ids,
                      x->d_codes, x->d_codesBin, x->d_bias, x->d_codesGrp, x->d_filter, x->d_codesX, x->d_seq2d, x->d_filter1};
  for (size_t j = 0; j < sizeof(ps) / sizeof(ps[0]); ++j) v.p[j] = ps[j];
  v.u[0] = x->heurRows; v.u[1] = x->maxMultiIndex; v.u[2] = x->nIds; v.u[3] = x->nTotal; v.u[4] = x->nCodes; v.u[5] = x->idBase; v.u[6] = x->scratchBudget;
  v.w[0] = x->tableBits; v.w[1] = x->maxBin; v.w[2] = x->filterBits; v.w[3] = x->dbg; v.w[4] = x->seq2dDc; v.w[5] = x->filter1Bits;
  v.f[0] = x->coarseMax;
  for (int j = 0; j < 9; ++j) v.f[1 + j] = x->slopeThr[j];
  v.i[0] = x->grpG; v.i[1] = x->useRuns; v.i[2] = x->numCUs; v.i[3] = x->balance; v.i[4] = x->xcodeShift; v.i[5] = x->useXCode; v.i[6] = x->sharedRows; v.i[7] = x->useFilter1; v.i[8] = x->srKernel; v.i[9] = x->srScanSplit | (x->srScanDepth << 8); v.i[10] = x->coopRerank;
  const bool bs[] = {x->haveTree, x->sharded, x->haveBins, x->binOrdered, x->linesDropped, x->biasReady, x->adcBias, x->exactFilter, x->smallLists,
                     x->forceUnfused, x->useWgRerank, x->noShape, x->heur2d};
  for (size_t j = 0; j < sizeof(bs) / sizeof(bs[0]); ++j) v.b[j] = bs[j];
}
void applyShared(pqt_index* t, const SharedWords& v) {
  t->dp = v.dp; t->prm = v.prm;
  t->d_cb1 = (float*)v.p[0]; t->d_cb2 = (float*)v.p[1]; t->d_coarse = (float*)v.p[2]; t->d_cb1L = (float*)v.p[3]; t->d_cb2T = (float*)v.p[4];
  t->d_heur = (uint16_t*)v.p[5]; t->d_heur8 = (uint16_t*)v.p[6]; t->d_heur4 = (uint32_t*)v.p[7];
  t->d_table = (PqtBinEntry*)v.p[8]; t->d_lower = (uint32_t*)v.p[9]; t->d_ids = (uint32_t*)v.p[10];
  t->d_codes = (uint32_t*)v.p[11]; t->d_codesBin = (uint32_t*)v.p[12]; t->d_bias = (float*)v.p[13]; t->d_codesGrp = (uint32_t*)v.p[14]; t->d_filter = (uint32_t*)v.p[15]; t->d_codesX = (uint32_t*)v.p[16]; t->d_seq2d = (uint32_t*)v.p[17]; t->d_filter1 = (uint32_t*)v.p[18];
  t->codesOwned = false;
  t->heurRows = v.u[0]; t->maxMultiIndex = v.u[1]; t->nIds = v.u[2]; t->nTotal = v.u[3]; t->nCodes = v.u[4]; t->idBase = v.u[5]; t->scratchBudget = (size_t)v.u[6];
  t->tableBits = v.w[0]; t->maxBin = v.w[1]; t->filterBits = v.w[2]; t->dbg = v.w[3]; t->seq2dDc = v.w[4]; t->filter1Bits = v.w[5];
  t->coarseMax = v.f[0];
  for (int j = 0; j < 9; ++j) t->slopeThr[j] = v.f[1 + j];
  t->grpG = v.i[0]; t->useRuns = v.i[1]; t->numCUs = v.i[2]; t->balance = v.i[3]; t->xcodeShift = v.i[4]; t->useXCode = v.i[5]; t->sharedRows = v.i[6]; t->useFilter1 = v.i[7]; t->srKernel = v.i[8] ? v.i[8] : 1; if (v.i[9]) { t->srScanSplit = v.i[9] & 0xff; t->srScanDepth = v.i[9] >> 8; } t->coopRerank = v.i[10];
  t->haveTree = v.b[0]; t->sharded = v.b[1]; t->haveBins = v.b[2]; t->binOrdered = v.b[3]; t->linesDropped = v.b[4]; t->biasReady = v.b[5]; t->adcBias = v.b[6];
  t->exactFilter = v.b[7]; t->smallLists = v.b[8]; t->forceUnfused = v.b[9]; t->useWgRerank = v.b[10]; t->noShape = v.b[11]; t->heur2d = v.b[12];
  t->stageTiming = 0;  // a view never carries stage events (timed calls are not split)
}

// Opt-in ("overlap" >= 1).  History: with one statistics atomic per LANE in the rerank (pqt_count_ties) two half-size launches on two
// handles ran 1.24x faster than one launch -- two counter words instead of one -- and the split was made the default for the SIFT1M
// shape; with the atomics reduced to one per workgroup the one-piece call is the faster one (0.167 against 0.173 ms per 10 k queries),
// so nothing is split automatically any more.
bool overlapWanted(const pqt_index* idx, uint32_t qn, uint32_t k) {
  if (!idx || idx->isView || idx->overlap < 1 || idx->d_tstamp || (idx->dbg & 0xffffu) || k > PQT_RS_BEST || qn < 2) return false;
  if (!idx->haveTree || !idx->haveBins || !(idx->d_codes || idx->binOrdered) || !idx->d_heur) return false;  // (queryImpl reports it)
  if (idx->stageTiming > 0 && (idx->timingPhase % (unsigned long long)idx->stageTiming) == 0) return false;      // the next call is a timed one
  return true;
}

// A user view (pqt_index_create_view) re-reads the owner's shared words at every call: whatever the owner loaded or built since
// (heuristic prefix, bin-ordered store, group-major copy, row bias, options) is what the view works with.  It keeps its own
// scratch, stream, statistics and "stage_timing".
void refreshUserView(pqt_index* v) {
  if (!v || !v->userView || !v->owner) return;
  SharedWords s;
  captureShared(v->owner, s);
  const int keepTiming = v->stageTiming;
  applyShared(v, s);
  v->stageTiming = keepTiming;
}

int queryTop(pqt_index* idx, const float* q_dev, uint32_t qn, uint32_t Bv, uint32_t Bb, uint32_t k, uint32_t* outIdx, float* outDist,
             uint32_t* outPos, uint32_t* outCount, hipStream_t st, int sync, const unsigned long long* binsIn = nullptr, uint32_t binsCap = 0) {
  refreshUserView(idx);
  if (!overlapWanted(idx, qn, k)) return queryImpl(idx, q_dev, qn, Bv, Bb, k, outIdx, outDist, outPos, outCount, st, sync, binsIn, binsCap);
  if (!q_dev || !outIdx || !outDist || (idx->sharded && !outPos)) return queryImpl(idx, q_dev, qn, Bv, Bb, k, outIdx, outDist, outPos, outCount, st, sync, binsIn, binsCap);
  int rc = setDevice(idx);
  if (rc) return rc;
  // pieces and their shares.  Each piece's persistent rerank launch takes 1/P of the workgroup slots so that all launches are
  // resident together (full-size grids queue behind each other on the LDS); the first piece starts first and gets a slightly
  // larger share of the queries
  static const int envPieces = getenv("PQT_OVERLAP_PIECES") ? atoi(getenv("PQT_OVERLAP_PIECES")) : 0;
  static const int envFirst = getenv("PQT_OVERLAP_FIRST_PCT") ? atoi(getenv("PQT_OVERLAP_FIRST_PCT")) : 0;
  uint32_t P = idx->overlap >= 2 ? (uint32_t)idx->overlap : (envPieces >= 2 ? (uint32_t)envPieces : 2u);
  P = std::min<uint32_t>(std::min<uint32_t>(P, (uint32_t)pqt_index::kMaxViews + 1u), qn);
  for (uint32_t i = 0; i + 1 < P; ++i) {
    if (idx->views[i]) continue;
    pqt_index* t = nullptr;
    if ((rc = pqt_index_create(&idx->prm, idx->device, &t))) return rc;
    t->isView = true; t->owner = idx; idx->views[i] = t;
    HIPCHK(hipEventCreateWithFlags(&idx->evJoin[i], hipEventDisableTiming));
  }
  if (!idx->evFork) HIPCHK(hipEventCreateWithFlags(&idx->evFork, hipEventDisableTiming));
  if (!st) st = idx->stream;
  if (idx->stageTiming > 0) idx->timingPhase++;  // this call's turn in the every-N-th-call rhythm (the pieces are never timed)
  const int keepTiming = idx->stageTiming, keepCUs = idx->numCUs;
  const size_t D = idx->dp.D;
  uint32_t start[pqt_index::kMaxViews + 2] = {0};
  {
    const uint32_t firstPct = envFirst > 0 ? (uint32_t)envFirst : (100u / P + (P == 2 ? 6u : 4u));
    uint32_t first = (uint32_t)(((uint64_t)qn * firstPct + 99) / 100);
    first = std::max<uint32_t>(1u, std::min<uint32_t>(first, qn - (P - 1)));
    start[1] = first;
    for (uint32_t i = 2; i <= P; ++i) start[i] = first + (uint32_t)(((uint64_t)(qn - first) * (i - 1)) / (P - 1));
  }
  const int share = std::max(1, keepCUs / (int)P);
  SharedWords s0, s1;
  idx->numCUs = share;
  captureShared(idx, s0);
  HIPCHK(hipEventRecord(idx->evFork, st));
  for (uint32_t i = 0; i + 1 < P; ++i) HIPCHK(hipStreamWaitEvent(idx->views[i]->stream, idx->evFork, 0));
  idx->stageTiming = 0;
  rc = queryImpl(idx, q_dev, start[1], Bv, Bb, k, outIdx, outDist, outPos, outCount, st, 0, binsIn, binsCap);
  idx->stageTiming = keepTiming;
  captureShared(idx, s1);
  idx->numCUs = keepCUs;
  if (rc) return rc;
  if (memcmp(&s0, &s1, sizeof(s0)) != 0) {  // built lazily on `st` by the first piece: visible to the views' streams from here
    HIPCHK(hipEventRecord(idx->evFork, st));
    for (uint32_t i = 0; i + 1 < P; ++i) HIPCHK(hipStreamWaitEvent(idx->views[i]->stream, idx->evFork, 0));
  }
  int rcPiece = PQT_OK;
  for (uint32_t i = 1; i < P; ++i) {
    pqt_index* const v = idx->views[i - 1];
    applyShared(v, s1);
    const uint32_t a = start[i], n = start[i + 1] - start[i];
    const int r2 = queryImpl(v, q_dev + (size_t)a * D, n, Bv, Bb, k, outIdx + (size_t)a * k, outDist + (size_t)a * k,
                             outPos ? outPos + (size_t)a * k : nullptr, outCount ? outCount + a : nullptr, v->stream, 0,
                             binsIn ? binsIn + (size_t)a * (binsCap + 1u) : nullptr, binsCap);
    if (r2 && !rcPiece) rcPiece = r2;
    HIPCHK(hipEventRecord(idx->evJoin[i - 1], v->stream));
    HIPCHK(hipStreamWaitEvent(st, idx->evJoin[i - 1], 0));
  }
  if (rcPiece) return rcPiece;
  idx->lastPieces = P;
  for (uint32_t i = 0; i <= P; ++i) idx->pieceStart[i] = start[i];
  idx->lastPath += " overlap=" + std::to_string(P) + "-pieces";
  if (sync) HIPCHK(hipStreamSynchronize(st));
  return PQT_OK;
}
}  // namespace

int pqt_query(pqt_index* idx, const float* q_dev, uint32_t qn, uint32_t Bv, uint32_t Bb, uint32_t k, uint32_t* outIdx,
              float* outDist, uint32_t* outCount, void* stream, int sync) {
  refreshUserView(idx);  // a view takes the owner's state (sharded-ness included) before anything is checked
  if (idx && idx->sharded) return fail(PQT_ERR_INVALID, "sharded index: use pqt_query_shard + pqt_merge_topk");
  return queryTop(idx, q_dev, qn, Bv, Bb, k, outIdx, outDist, nullptr, outCount, (hipStream_t)stream, sync);
}

int pqt_query_shard(pqt_index* idx, const float* q_dev, uint32_t qn, uint32_t Bv, uint32_t Bb, uint32_t k, uint32_t* outIdx,
                    float* outDist, uint32_t* outPos, uint32_t* outCount, void* stream, int sync) {
  refreshUserView(idx);  // a view takes the owner's state (sharded-ness included) before anything is checked
  if (idx && !idx->sharded) return fail(PQT_ERR_INVALID, "index was not loaded with pqt_index_set_bins_shard");
  return queryTop(idx, q_dev, qn, Bv, Bb, k, outIdx, outDist, outPos, outCount, (hipStream_t)stream, sync);
}

int pqt_traverse_bins(pqt_index* idx, const float* q_dev, uint32_t qn, uint32_t Bv, uint32_t Bb, uint32_t cap,
                      unsigned long long* out_bins_dev, void* stream, int sync) {
  if (!idx || !out_bins_dev || (qn && !q_dev)) return fail(PQT_ERR_INVALID, "null argument");
  refreshUserView(idx);
  if (!idx->sharded) return fail(PQT_ERR_INVALID, "pqt_traverse_bins needs a range-sharded index (pqt_index_set_bins_shard / _local)");
  if (cap == 0 || cap > PQT_GBIN_MAX) return fail(PQT_ERR_LIMIT, "bin-list capacity must be 1..256");
  if (!idx->haveTree || !idx->haveBins || !idx->d_heur) return fail(PQT_ERR_STATE, "index needs codebooks, heuristic and bins before traversing");
  if (qn == 0) return PQT_OK;
  int rc = setDevice(idx);
  if (rc) return rc;
  hipStream_t st = stream ? (hipStream_t)stream : idx->stream;
  const PqtDevParams& d = idx->dp;
  const uint64_t He64 = std::min<uint64_t>(Bb, idx->maxMultiIndex);
  if (He64 > idx->heurRows) return fail(PQT_ERR_STATE, "bound_bins exceeds the heuristic rows held by the index (build/set a longer prefix)");
  const uint32_t He = (uint32_t)He64;
  TravPlan tp;
  if ((rc = planTraversal(idx, He, tp))) return rc;
  if (!tp.fused) {
    // bounds / shapes outside the fused traversal: every receiver traverses every query itself (correct, not fast)
    hipLaunchKernelGGL(pqt_k_gbins_overflow, dim3((qn + 255) / 256), dim3(256), 0, st, out_bins_dev, cap, qn);
  } else {
    if ((rc = ensureQueryScratch(idx, qn))) return rc;
    const PqtTravArgs targs{q_dev, idx->d_cb1, idx->d_cb2, (const float4*)idx->d_cb2T, d, (const uint4*)idx->d_heur8, (idx->dbg & 4096u) ? nullptr : idx->d_heur4, He, Bv,
                            idx->d_table, idx->d_lower, idx->tableBits, idx->d_ids, qn, idx->d_qL1virt, idx->d_cand,
                            idx->d_candPos, idx->d_nCand, idx->d_nLocal, idx->d_nIncl, 0, idx->d_counters + 8 * kCtrRing /* spare statistics block */, nullptr,
                            idx->d_segD, idx->d_segBin, idx->d_ovList, idx->d_ovCount,
                            (idx->dbg & 2048u) ? nullptr : idx->d_filter, idx->filterBits,
                            nullptr, nullptr, nullptr, 0u, nullptr, (idx->dbg >> 5) & 3u,
                            nullptr, nullptr, 0u, nullptr, nullptr, out_bins_dev, cap,
                            (idx->useFilter1 > 0 && !(idx->dbg & 2048u)) ? idx->d_filter1 : nullptr, idx->filter1Bits, idx->useFilter1 == 2 ? 1u : 0u};
    launchFusedTraversal(idx, targs, tp, qn, st, nullptr, nullptr);
  }
  HIPCHK(hipGetLastError());
  if (sync) HIPCHK(hipStreamSynchronize(st));
  return PQT_OK;
}

int pqt_query_shard_bins(pqt_index* idx, const float* q_dev, uint32_t qn, uint32_t Bv, uint32_t Bb, uint32_t k,
                         const unsigned long long* bins_dev, uint32_t cap, uint32_t* outIdx, float* outDist, uint32_t* outPos,
                         uint32_t* outCount, void* stream, int sync) {
  refreshUserView(idx);  // a view takes the owner's state (sharded-ness included) before anything is checked
  if (idx && !idx->sharded) return fail(PQT_ERR_INVALID, "index was not loaded with pqt_index_set_bins_shard / _local");
  if (!bins_dev || cap == 0 || cap > PQT_GBIN_MAX) return fail(PQT_ERR_INVALID, "bin lists missing or capacity outside 1..256");
  return queryTop(idx, q_dev, qn, Bv, Bb, k, outIdx, outDist, outPos, outCount, (hipStream_t)stream, sync, bins_dev, cap);
}

int pqt_query_host(pqt_index* idx, const float* q, uint32_t qn, uint32_t Bv, uint32_t Bb, uint32_t k, uint32_t* outIdx,
                   float* outDist, uint32_t* outCount) {
  if (!idx || !q || !outIdx || !outDist) return fail(PQT_ERR_INVALID, "null argument");
  int rc = setDevice(idx);
  if (rc) return rc;
  // staging buffers live in the handle and only grow: a caller looping over single vectors (treequantizer::query style)
  // pays two copies per call, not four allocations
  const size_t nq = (size_t)qn * idx->dp.D, nk = (size_t)qn * k;
  if (nq > idx->h2dQCap) { if ((rc = devAlloc(&idx->h2dQ, nq))) { idx->h2dQCap = 0; return rc; } idx->h2dQCap = nq; }
  if (nk > idx->h2dKCap) {
    idx->h2dKCap = 0;
    if ((rc = devAlloc(&idx->h2dI, nk)) || (rc = devAlloc(&idx->h2dD, nk))) return rc;
    idx->h2dKCap = nk;
  }
  if (qn > idx->h2dCCap) { if ((rc = devAlloc(&idx->h2dC, (size_t)qn))) { idx->h2dCCap = 0; return rc; } idx->h2dCCap = qn; }
  HIPCHK(hipMemcpyAsync(idx->h2dQ, q, nq * 4, hipMemcpyHostToDevice, idx->stream));
  if ((rc = pqt_query(idx, idx->h2dQ, qn, Bv, Bb, k, idx->h2dI, idx->h2dD, idx->h2dC, idx->stream, 0))) return rc;
  HIPCHK(hipMemcpyAsync(outIdx, idx->h2dI, nk * 4, hipMemcpyDeviceToHost, idx->stream));
  HIPCHK(hipMemcpyAsync(outDist, idx->h2dD, nk * 4, hipMemcpyDeviceToHost, idx->stream));
  if (outCount) HIPCHK(hipMemcpyAsync(outCount, idx->h2dC, (size_t)qn * 4, hipMemcpyDeviceToHost, idx->stream));
  HIPCHK(hipStreamSynchronize(idx->stream));
  return PQT_OK;
}

int pqt_query_candidates(pqt_index* idx, const float* q_dev, uint32_t qn, uint32_t Bv, uint32_t Bb, uint32_t cap, uint32_t* outIdx,
                         float* outDist, uint32_t* outCount, void* stream, int sync) {
  if (!outCount) return fail(PQT_ERR_INVALID, "pqt_query_candidates needs out_count_dev (a list longer than cap is truncated to its first cap entries)");
  return pqt_query(idx, q_dev, qn, Bv, Bb, cap, outIdx, outDist, outCount, stream, sync);
}

int pqt_index_device_arrays(const pqt_index* idx, const uint32_t** ids_dev, const uint32_t** codes_bin_dev, uint64_t* n_local) {
  if (!idx) return fail(PQT_ERR_INVALID, "null argument");
  if (!idx->haveBins) return fail(PQT_ERR_STATE, "no bins loaded");
  if (codes_bin_dev) {
    if (!idx->binOrdered) {
      if (!idx->d_codes) return fail(PQT_ERR_STATE, "no line codes loaded");
      int rc = reorderLines(const_cast<pqt_index*>(idx));
      if (rc) return rc;
    }
    *codes_bin_dev = idx->d_codesBin;
  }
  if (ids_dev) *ids_dev = idx->d_ids;
  if (n_local) *n_local = idx->nIds;
  return PQT_OK;
}

int pqt_merge_topk(pqt_index* idx, uint32_t nsh, uint32_t qn, uint32_t k, const uint32_t* inIdx, const float* inDist,
                   const uint32_t* inPos, uint64_t shard_stride, uint32_t* outIdx, float* outDist, void* stream, int sync) {
  if (shard_stride == 0) shard_stride = (uint64_t)qn * k;
  if (!idx || !inIdx || !inDist || !inPos || !outIdx || !outDist || !nsh || !k) return fail(PQT_ERR_INVALID, "null argument");
  int rc = setDevice(idx);
  if (rc) return rc;
  hipStream_t st = stream ? (hipStream_t)stream : idx->stream;
  const uint32_t mP2 = np2(std::max<uint64_t>((uint64_t)nsh * k, 2));
  const size_t lds = (size_t)mP2 * 12;
  if (lds > kMaxLds || (uint64_t)nsh * k > (1ull << 30)) {
    // lists too long for the LDS-resident merge (e.g. whole candidate lists, k = 8192): rank every entry by binary searches
    if (nsh > 64) return fail(PQT_ERR_LIMIT, "at most 64 shards in a merge");
    if (qn) hipLaunchKernelGGL(pqt_k_merge_ranked, dim3(qn), dim3(PQT_BLOCK), 0, st, inIdx, inDist, inPos, nsh, qn, k, shard_stride, outIdx, outDist);
  } else {
    if ((rc = allowLds(pqt_k_merge, lds))) return rc;
    if (qn) hipLaunchKernelGGL(pqt_k_merge, dim3(qn), dim3(PQT_BLOCK), lds, st, inIdx, inDist, inPos, nsh, qn, k, shard_stride, mP2, outIdx, outDist);
  }
  HIPCHK(hipGetLastError());
  if (sync) HIPCHK(hipStreamSynchronize(st));
  return PQT_OK;
}

int pqt_compact_results(pqt_index* idx, uint32_t qn, uint32_t k, const uint32_t* idx_dev, const float* dist_dev, const uint32_t* count_dev,
                        uint32_t* offsets_dev, uint32_t* packed_idx_dev, float* packed_dist_dev, void* stream, int sync) {
  if (!idx || !idx_dev || !dist_dev || !count_dev || !offsets_dev || !packed_idx_dev || !packed_dist_dev || !k) return fail(PQT_ERR_INVALID, "null argument");
  if ((uint64_t)qn * k > 0xffffffffull) return fail(PQT_ERR_LIMIT, "more than 2^32 result slots in one batch");
  int rc = setDevice(idx);
  if (rc) return rc;
  hipStream_t st = stream ? (hipStream_t)stream : idx->stream;
  hipLaunchKernelGGL(pqt_k_row_offsets, dim3(1), dim3(1024), 0, st, count_dev, qn, k, offsets_dev);
  if (qn) hipLaunchKernelGGL(pqt_k_compact_rows, dim3((qn + 3) / 4), dim3(256), 0, st, idx_dev, dist_dev, offsets_dev, qn, k, packed_idx_dev, packed_dist_dev);
  HIPCHK(hipGetLastError());
  if (sync) HIPCHK(hipStreamSynchronize(st));
  return PQT_OK;
}

int pqt_index_device_bytes(const pqt_index* idx, uint64_t* out8) {
  if (!idx || !out8) return fail(PQT_ERR_INVALID, "null argument");
  const PqtDevParams& d = idx->dp;
  const uint64_t rows = idx->nIds, lp4 = (uint64_t)d.LP * 4;
  out8[0] = (idx->d_codes && idx->codesOwned) ? idx->nCodes * lp4 : 0;                 // id-ordered line store as handed over (owned copy; dropped after the reorder)
  out8[1] = idx->d_codesBin ? rows * lp4 : 0;                                           // bin-ordered line store
  out8[2] = idx->d_codesGrp ? rows * lp4 : 0;                                           // group-major copy (coarse table beyond the LDS)
  out8[3] = idx->d_codesX ? rows * lp4 : 0;                                             // X-code copy (LDS-table rerank at C1 = 32)
  out8[4] = (idx->d_bias ? rows * 4 : 0) + rows * 4;                                    // row bias + member ids
  out8[5] = (idx->tableBits ? ((uint64_t)1 << idx->tableBits) * (sizeof(PqtBinEntry) + (idx->d_lower ? 4 : 0)) : 0) + (idx->filterBits ? ((uint64_t)1 << idx->filterBits) / 8 : 0);  // bin table + presence bitmap
  out8[6] = (uint64_t)d.C1 * d.D * 8 + (uint64_t)d.P * d.C1 * d.C2 * d.S * 8 + (uint64_t)d.LP * d.C1 * d.C1 * 4 + idx->heurRows * 22;  // codebooks (+ re-tiled copies), coarse table, heuristic
  out8[7] = idx->candCap * (idx->sharded ? 12 : 8) + (uint64_t)idx->qCap * ((uint64_t)d.LP * d.C1 * 4 + (uint64_t)d.P * d.WC * 8 + 32) + idx->sortCap * 8 +
            idx->srTableCap * 4 + idx->srPairCap * 4 + idx->srBlockCap * 4 + idx->srItemCap * 8 + idx->srKeysCap * 8 + idx->srSegCap * 8;  // scratch arena of this handle (the shared-row pass's tables and lists included)
  return PQT_OK;
}

uint64_t pqt_debug_stride(const pqt_index* idx) { return idx ? idx->stride : 0; }

int pqt_debug_read(const pqt_index* idx, uint32_t qn, float* l1virt, float* segd, uint32_t* segbin, uint32_t* candIdx,
                   float* candDist, uint32_t* ncand) {
  if (!idx) return fail(PQT_ERR_INVALID, "null argument");
  if (idx->lastPieces > 1 && qn > idx->pieceStart[1]) {  // overlapped call: the later pieces' scratch lives in the view handles
    const PqtDevParams& dd = idx->dp;
    if (qn > idx->pieceStart[idx->lastPieces]) return fail(PQT_ERR_STATE, "no such batch held");
    for (uint32_t i = 0; i < idx->lastPieces && idx->pieceStart[i] < qn; ++i) {
      const pqt_index* h = i == 0 ? idx : idx->views[i - 1];
      const uint32_t a = idx->pieceStart[i], n = std::min<uint32_t>(qn, idx->pieceStart[i + 1]) - a;
      if (h->stride != idx->stride) return fail(PQT_ERR_STATE, "the pieces of the last call used different candidate strides");
      const int rc2 = pqt_debug_read(h, n, l1virt ? l1virt + (size_t)a * dd.LP * dd.C1 : nullptr, segd ? segd + (size_t)a * dd.P * dd.WC : nullptr,
                                     segbin ? segbin + (size_t)a * dd.P * dd.WC : nullptr, candIdx ? candIdx + (size_t)a * idx->stride : nullptr,
                                     candDist ? candDist + (size_t)a * idx->stride : nullptr, ncand ? ncand + a : nullptr);
      if (rc2) return rc2;
    }
    return PQT_OK;
  }
  if (qn > idx->lastQn) return fail(PQT_ERR_STATE, "no such batch held");
  if (idx->nChunks > 1 && (candIdx || candDist)) return fail(PQT_ERR_STATE, "last batch ran in several chunks; candidates of earlier chunks are gone");
  if ((segd || segbin) && !idx->lastSegKept)
    return fail(PQT_ERR_STATE, "the last call ran the fused traversal, which keeps seg_d2/seg_bin on chip: set_option(\"fused\", 0) first");
  if (candIdx && idx->lastRuns)
    return fail(PQT_ERR_STATE, "the last call handed bin runs from the traversal to the rerank, no candidate list was materialised: set_option(\"bin_runs\", 0) or use k > 128");
  if (candDist && !idx->lastDistKept)
    return fail(PQT_ERR_STATE, "the last call ran the fused rerank+select, which never writes cand_dist: use k > 128 or set_option(\"fused\", 0)");
  int rc = setDevice(idx);
  if (rc) return rc;
  const PqtDevParams& d = idx->dp;
  HIPCHK(hipDeviceSynchronize());
  if (l1virt) HIPCHK(hipMemcpy(l1virt, idx->d_qL1virt, (size_t)qn * d.LP * d.C1 * 4, hipMemcpyDeviceToHost));
  if (segd) HIPCHK(hipMemcpy(segd, idx->d_segD, (size_t)qn * d.P * d.WC * 4, hipMemcpyDeviceToHost));
  if (segbin) HIPCHK(hipMemcpy(segbin, idx->d_segBin, (size_t)qn * d.P * d.WC * 4, hipMemcpyDeviceToHost));
  if (candIdx) {
    HIPCHK(hipMemcpy(candIdx, idx->d_cand, (size_t)qn * idx->stride * 4, hipMemcpyDeviceToHost));
    // the device list holds positions in the bin-ordered store; report vector ids like the reference's list
    std::vector<uint32_t> hid(idx->nIds), hn(qn);
    if (idx->nIds) HIPCHK(hipMemcpy(hid.data(), idx->d_ids, idx->nIds * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(hn.data(), idx->d_nLocal, (size_t)qn * 4, hipMemcpyDeviceToHost));
    for (uint32_t q = 0; q < qn; ++q)
      for (uint32_t j = 0; j < hn[q]; ++j) { uint32_t& v = candIdx[(size_t)q * idx->stride + j]; v = v < idx->nIds ? hid[v] : 0xffffffffu; }
  }
  if (candDist) HIPCHK(hipMemcpy(candDist, idx->d_candDist, (size_t)qn * idx->stride * 4, hipMemcpyDeviceToHost));
  if (ncand) HIPCHK(hipMemcpy(ncand, idx->d_nLocal, (size_t)qn * 4, hipMemcpyDeviceToHost));
  return PQT_OK;
}

int pqt_debug_tstamps(const pqt_index* idx, unsigned long long* out, uint32_t qn) {
  if (!idx || !idx->d_tstamp || qn > (1u << 16)) return fail(PQT_ERR_STATE, "timestamps not enabled (PQT_TSTAMP=1)");
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(out, idx->d_tstamp, (size_t)qn * PQT_TS_WORDS * 8, hipMemcpyDeviceToHost));
  return PQT_OK;
}

int pqt_debug_calibrate_gather(int device, uint32_t log2_rows, uint32_t row_bytes, uint64_t gathers, float* out_ms) {
  const bool coop = (row_bytes & 0x1000u) != 0;  // + 0x1000: one lane per 16-byte piece (adjacent lanes share a row)
  row_bytes &= 0xfffu;
  if ((row_bytes != 64 && row_bytes != 128) || log2_rows < 10 || log2_rows > 34) return fail(PQT_ERR_INVALID, "row_bytes 64|128, 10 <= log2_rows <= 34");
  HIPCHK(hipSetDevice(device));
  const uint64_t rows = 1ull << log2_rows;
  if (gathers > rows) gathers = rows;
  void* table = nullptr; unsigned long long* sink = nullptr;
  HIPCHK(hipMalloc(&table, rows * row_bytes));
  HIPCHK(hipMalloc((void**)&sink, 8));
  HIPCHK(hipMemset(table, 1, rows * row_bytes));
  HIPCHK(hipMemset(sink, 0, 8));
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipEventRecord(e0, 0));
  const unsigned grid = (unsigned)((gathers + 255) / 256);
  if (coop) {
    const unsigned gridc = (unsigned)((gathers * (row_bytes / 16) + 255) / 256);
    if (row_bytes == 64) hipLaunchKernelGGL(pqt_k_calib_gather_coop<4>, dim3(gridc), dim3(256), 0, 0, (const uint4*)table, rows, gathers, sink);
    else hipLaunchKernelGGL(pqt_k_calib_gather_coop<8>, dim3(gridc), dim3(256), 0, 0, (const uint4*)table, rows, gathers, sink);
  } else if (row_bytes == 64) hipLaunchKernelGGL(pqt_k_calib_gather<4>, dim3(grid), dim3(256), 0, 0, (const uint4*)table, rows, gathers, sink);
  else hipLaunchKernelGGL(pqt_k_calib_gather<8>, dim3(grid), dim3(256), 0, 0, (const uint4*)table, rows, gathers, sink);
  HIPCHK(hipEventRecord(e1, 0));
  HIPCHK(hipDeviceSynchronize());
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  if (out_ms) *out_ms = ms;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipFree(table); (void)hipFree(sink);
  return PQT_OK;
}

int pqt_debug_stream_read(int device, uint64_t bytes, int reps, float* out_ms) {
  if (bytes < (1u << 20) || reps < 1 || !out_ms) return fail(PQT_ERR_INVALID, "bytes >= 1 MiB, reps >= 1");
  HIPCHK(hipSetDevice(device));
  void* buf = nullptr; unsigned long long* sink = nullptr;
  HIPCHK(hipMalloc(&buf, bytes));
  if (hipMalloc((void**)&sink, 8) != hipSuccess) { (void)hipFree(buf); return fail(PQT_ERR_DEVICE, "allocation failed"); }
  hipDeviceProp_t prop;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  float ms = 0;
  hipError_t e = hipMemset(buf, 1, bytes);
  if (e == hipSuccess) e = hipMemset(sink, 0, 8);
  if (e == hipSuccess) e = hipGetDeviceProperties(&prop, device);
  if (e == hipSuccess) e = hipEventCreate(&e0);
  if (e == hipSuccess) e = hipEventCreate(&e1);
  if (e == hipSuccess) {
    // the best of a few launch shapes (loads in flight per lane x grid-stride / contiguous share per workgroup x workgroups per CU):
    // which one wins differs between boxes by a few per cent, and the probe is meant to say what the memory system gives a plain reader
    using Kern = void (*)(const uint4*, uint64_t, unsigned long long*);
    const Kern kerns[] = {pqt_k_stream_read<4, false>, pqt_k_stream_read<8, false>, pqt_k_stream_read<16, false>, pqt_k_stream_read<8, true>, pqt_k_stream_read<16, true>};
    const unsigned perCu[] = {8u, 16u, 32u};
    float best = 0.f;
    for (const Kern kern : kerns) {
      for (const unsigned wpc : perCu) {
        const unsigned grid = (unsigned)std::max(1, prop.multiProcessorCount) * wpc;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, (const uint4*)buf, bytes / 16, sink);  // warm-up
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e == hipSuccess) e = hipEventRecord(e0, 0);
        for (int r = 0; r < reps && e == hipSuccess; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, (const uint4*)buf, bytes / 16, sink);
        if (e == hipSuccess) e = hipEventRecord(e1, 0);
        if (e == hipSuccess) e = hipDeviceSynchronize();
        float t = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&t, e0, e1);
        if (e == hipSuccess && (best == 0.f || t < best)) best = t;
      }
    }
    ms = best;
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipFree(buf); (void)hipFree(sink);
  if (e != hipSuccess) return fail(PQT_ERR_DEVICE, hipGetErrorString(e));
  *out_ms = ms / (float)reps;
  return PQT_OK;
}

int pqt_debug_sort_scan(int device, uint32_t mode, uint32_t n, uint32_t* out_host) {
  if (!out_host || mode > 8) return fail(PQT_ERR_INVALID, "mode 0..8, out_host[n + 1]");
  if (mode == 6) n = 512;  // six exchanges, the lane + 1 move and a 64-key u32 sort x 64 lanes
  if (mode == 7) n = 64;
  if (mode == 8) n = 256;  // four 64-key row sorts
  const bool p2 = (n >= 64 && (n & (n - 1)) == 0) || mode == 6;
  if (!p2 || n > 8192 || (mode == 0 && n > 2048) || (mode == 2 && n != 512 && n != 1024) || (mode == 5 && n < 256))
    return fail(PQT_ERR_INVALID, "n: a power of two, 64..8192 (wave sort <= 2048, wave select 512 | 1024, block scan >= 256)");
  HIPCHK(hipSetDevice(device));
  uint32_t* d = nullptr;
  HIPCHK(hipMalloc((void**)&d, ((size_t)n + 1) * 4));
  hipError_t e = hipMemset(d, 0xff, ((size_t)n + 1) * 4);
  const size_t lds = (size_t)n * 8 + 256 * 4 + 4 * 8 + 64;
  if (e == hipSuccess) {
    int rc = allowLds(pqt_k_debug_sortscan, lds);
    if (rc) { (void)hipFree(d); return rc; }
    hipLaunchKernelGGL(pqt_k_debug_sortscan, dim3(1), dim3(256), lds, 0, mode, n, d);
    e = hipDeviceSynchronize();
  }
  if (e == hipSuccess) e = hipMemcpy(out_host, d, ((size_t)n + 1) * 4, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) return fail(PQT_ERR_DEVICE, hipGetErrorString(e));
  return PQT_OK;
}

int pqt_get_stats(const pqt_index* cidx, pqt_stats* out) {
  if (!cidx || !out) return fail(PQT_ERR_INVALID, "null argument");
  pqt_index* idx = const_cast<pqt_index*>(cidx);
  int rc = setDevice(idx);
  if (rc) return rc;
  HIPCHK(hipDeviceSynchronize());
  if (idx->d_coopErr) {
    // the cooperative scan's wavefronts meet through LDS with a bounded wait: a wavefront that gave up left its queries unanswered
    uint32_t ce = 0;
    HIPCHK(hipMemcpy(&ce, idx->d_coopErr, 4, hipMemcpyDeviceToHost));
    if (ce) return fail(PQT_ERR_DEVICE, "cooperative rerank: a wavefront gave up waiting for its partner (results of that call are incomplete)");
  }
  unsigned long long c[8] = {0};
  HIPCHK(hipMemcpy(c, idx->ctr ? idx->ctr : idx->d_counters, sizeof(c), hipMemcpyDeviceToHost));
  pqt_stats s{};
  s.queries = idx->lastQn; s.ties_l1 = c[0]; s.ties_l2 = c[1]; s.ties_bins = c[2]; s.ties_final = c[3];
  s.max_bin = idx->maxBin;
  if (idx->lastFilter && idx->d_fbCount) { uint32_t fb = 0; HIPCHK(hipMemcpy(&fb, idx->d_fbCount, 4, hipMemcpyDeviceToHost)); s.filter_fallbacks = fb; }
  {
    std::vector<uint32_t> nl(idx->lastQn), ni(idx->lastQn);
    if (idx->lastQn) {
      HIPCHK(hipMemcpy(nl.data(), idx->d_nLocal, (size_t)idx->lastQn * 4, hipMemcpyDeviceToHost));
      HIPCHK(hipMemcpy(ni.data(), idx->d_nIncl, (size_t)idx->lastQn * 4, hipMemcpyDeviceToHost));
    }
    for (uint32_t v : nl) s.candidates += v;
    for (uint32_t v : ni) s.bins_nonempty += v;
    s.bins_visited = (uint64_t)idx->lastHe * idx->lastQn;
  }
  for (uint32_t pi = 1; pi < idx->lastPieces; ++pi) {  // the later pieces of an overlapped call ran on the view handles
    const pqt_index* tw = idx->views[pi - 1];
    unsigned long long c2[8] = {0};
    HIPCHK(hipMemcpy(c2, tw->ctr ? tw->ctr : tw->d_counters, sizeof(c2), hipMemcpyDeviceToHost));
    s.queries += tw->lastQn; s.ties_l1 += c2[0]; s.ties_l2 += c2[1]; s.ties_bins += c2[2]; s.ties_final += c2[3];
    if (tw->lastFilter && tw->d_fbCount) { uint32_t fb = 0; HIPCHK(hipMemcpy(&fb, tw->d_fbCount, 4, hipMemcpyDeviceToHost)); s.filter_fallbacks += fb; }
    std::vector<uint32_t> nl(tw->lastQn), ni(tw->lastQn);
    if (tw->lastQn) {
      HIPCHK(hipMemcpy(nl.data(), tw->d_nLocal, (size_t)tw->lastQn * 4, hipMemcpyDeviceToHost));
      HIPCHK(hipMemcpy(ni.data(), tw->d_nIncl, (size_t)tw->lastQn * 4, hipMemcpyDeviceToHost));
    }
    for (uint32_t v : nl) s.candidates += v;
    for (uint32_t v : ni) s.bins_nonempty += v;
    s.bins_visited += (uint64_t)tw->lastHe * tw->lastQn;
  }
  // stage times: those of the last call when it carried events, else of the most recent call that did (stage_timing = N > 1)
  int lastEv = -1;
  const int tslot = lastTimedSlot(idx);
  const int tchunks = tslot >= 0 ? idx->ringChunks[tslot] : 0;
  for (int ch = 0; ch < tchunks; ++ch) {
    float st[5] = {0, 0, 0, 0, 0};
    lastEv = stageMs(idx, tslot, ch, st);
    s.ms_tables += st[0]; s.ms_bins += st[1]; s.ms_rerank += st[2] + st[3]; s.ms_select += st[4];
  }
  if (tchunks > 0 && lastEv > 0) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, idx->evRing[tslot][0][EV_BEGIN], idx->evRing[tslot][tchunks - 1][lastEv]) == hipSuccess) s.ms_total = ms;
  }
  idx->stats = s;
  *out = s;
  return PQT_OK;
}

int pqt_get_shared_rows_stats(const pqt_index* idx, uint64_t* out8) {
  if (!idx || !out8) return fail(PQT_ERR_INVALID, "null argument");
  if (!idx->srStats || !idx->srStatPtr || !idx->lastShared) return fail(PQT_ERR_STATE, "no statistics held: set_option(\"sr_stats\", 1) and run a query that takes the shared-row pass");
  int rc = setDevice(idx);
  if (rc) return rc;
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(out8, idx->srStatPtr, 8 * sizeof(uint64_t), hipMemcpyDeviceToHost));
  return PQT_OK;
}

int pqt_get_last_path(const pqt_index* idx, char* out, int cap) {
  if (!idx || !out || cap < 1) return fail(PQT_ERR_INVALID, "null argument");
  const size_t n = std::min<size_t>(idx->lastPath.size(), (size_t)cap - 1);
  memcpy(out, idx->lastPath.data(), n);
  out[n] = 0;
  return (int)n;
}

int pqt_get_stage_ms_history(const pqt_index* idx, float* out, int cap) {
  if (!idx || !out) return fail(PQT_ERR_INVALID, "null argument");
  if (hipSetDevice(idx->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return fail(PQT_ERR_DEVICE, "sync failed");
  const int have = (int)std::min<unsigned long long>(idx->calls, (unsigned long long)kRing);
  // the most recent TIMED calls of the ring (option "stage_timing": not every call carries events), oldest first
  int slots[kRing], n = 0;
  for (int b = 1; b <= have && n < cap; ++b) {
    const int slot = (int)((idx->calls - b) % kRing);
    if (idx->ringChunks[slot] > 0 && (idx->evMask[slot][0] & 1u)) slots[n++] = slot;
  }
  for (int i = 0; i < n; ++i) {
    const int slot = slots[n - 1 - i];
    float st[5] = {0, 0, 0, 0, 0};
    for (int ch = 0; ch < idx->ringChunks[slot]; ++ch) (void)stageMs(idx, slot, ch, st);
    for (int e = 0; e < 5; ++e) out[i * 5 + e] = st[e];
  }
  return n;
}

int pqt_get_rerank_launch_ms(const pqt_index* idx, float* out, int cap) {
  if (!idx || !out) return fail(PQT_ERR_INVALID, "null argument");
  if (hipSetDevice(idx->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return fail(PQT_ERR_DEVICE, "sync failed");
  // the last call when it carried events, else the most recent call that did (stage_timing = N > 1); 0 launches when no
  // call of the ring was timed (stage_timing = 0)
  const int tslot = lastTimedSlot(idx);
  int n = 0;
  for (int ch = 0; tslot >= 0 && ch < idx->ringChunks[tslot] && n < cap; ++ch) {
    float st[5] = {0, 0, 0, 0, 0};
    if (stageMs(idx, tslot, ch, st) < 0) return fail(PQT_ERR_DEVICE, "event read failed");
    const float ms = st[3];
    out[n++] = ms;
  }
  return n;
}

int pqt_dev_triangle(const float* a, const float* b, const float* c, const float* l, uint32_t n, float* outDist, float* outRatio,
                     uint16_t* outU16, float* outRound, int device) {
  if (!a || !b || !c || !l || !outDist || !outRatio || !outU16 || !outRound) return fail(PQT_ERR_INVALID, "null argument");
  HIPCHK(hipSetDevice(device));
  float* d = nullptr; uint16_t* du = nullptr;
  HIPCHK(hipMalloc((void**)&d, (size_t)n * 7 * 4 + 16));
  HIPCHK(hipMalloc((void**)&du, (size_t)n * 2 + 16));
  float *da = d, *db = d + n, *dc = d + 2 * (size_t)n, *dl = d + 3 * (size_t)n, *dD = d + 4 * (size_t)n, *dR = d + 5 * (size_t)n, *dF = d + 6 * (size_t)n;
  hipError_t e = hipMemcpy(da, a, (size_t)n * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(db, b, (size_t)n * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(dc, c, (size_t)n * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(dl, l, (size_t)n * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(pqt_k_triangle, dim3((n + 255) / 256), dim3(256), 0, 0, da, db, dc, dl, n, dD, dR, du, dF);
    e = hipDeviceSynchronize();
  }
  if (e == hipSuccess) e = hipMemcpy(outDist, dD, (size_t)n * 4, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(outRatio, dR, (size_t)n * 4, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(outRound, dF, (size_t)n * 4, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(outU16, du, (size_t)n * 2, hipMemcpyDeviceToHost);
  (void)hipFree(d); (void)hipFree(du);
  if (e != hipSuccess) return fail(PQT_ERR_DEVICE, hipGetErrorString(e));
  return PQT_OK;
}

}  // extern "C"
