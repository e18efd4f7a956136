// pqt_internal.h -- shared by the translation units of libpqt_hip.so (pqt_hip.hip: index management, query orchestration and the
// C-ABI; pqt_rerank_launch.hip: the instantiations of the fused rerank + select kernels; pqt_traverse_launch.hip: those of the
// fused traversal).  The split exists for build time only: the template instantiations compile in parallel.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <cmath>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/pqt_hip.h"
#include "pqt_kernels.h"

// thread-local error text behind pqt_last_error() (defined in pqt_hip.hip)
int pqtFail(int code, const std::string& msg);
#define fail pqtFail

#define HIPCHK(expr)                                                                         \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess)                                                                    \
      return fail(PQT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));        \
  } while (0)

inline uint32_t upow(uint32_t x, uint32_t n) { uint32_t r = 1; for (uint32_t i = 0; i < n; ++i) r = x * r; return r; }
inline uint32_t np2(uint64_t x) { uint64_t r = 1; while (r < x && r < (1ull << 31)) r <<= 1; return (uint32_t)r; }

enum { EV_BEGIN = 0, EV_TABLES, EV_BINS, EV_ORDER, EV_RERANK, EV_SELECT, EV_COUNT };
constexpr int kMaxChunks = 16;
constexpr int kRing = 32;        // per-stage event sets of the last kRing query calls
#ifndef PQT_RS_NW
#define PQT_RS_NW 12
#endif
constexpr int kFusedWaves = PQT_RS_NW;
// the exact rerank over X-code rows (pqt_rs_query XC, SIFT1M shape): 16 wavefronts per workgroup = 4 per SIMD under the 128-VGPR budget,
// 384 key slots each (64 KB coarse table + 16 x (3 KB keys + 2 KB L1virt) + ticket lists = 147 KB of the 160 KB LDS)
// k > 128 short-list kernels (pqt_k_rerank_sort_small): wavefronts per workgroup of the first pass (lists <= 1024) and of the second (<= 2048)
constexpr int kSmallWaves = 12;
constexpr int kMidWaves = 8;
#ifndef PQT_XC_WAVES
#define PQT_XC_WAVES 16
#endif
constexpr int kXcWaves = PQT_XC_WAVES;
constexpr int kXcSlots = 384;
constexpr int kCtrRing = 4;
constexpr int kPoolRing = 4;  // blocks of 16 draw counters + 8 x 64 registration counts behind the statistics ring (rerank schedule 2)
constexpr size_t kPoolWords = 16 + 8 * PQT_SCHED_CLASSES;
#ifndef PQT_TR_NW
#define PQT_TR_NW 1
#endif
constexpr int kTravWaves = PQT_TR_NW;    // wavefronts (= queries) per workgroup of the fused traversal kernel  // wavefronts per workgroup of the fused rerank+select kernel


struct pqt_index {
  pqt_params prm{};
  PqtDevParams dp{};
  int device = 0;
  hipStream_t stream = nullptr;
  // tree
  float* d_cb1 = nullptr; float* d_cb2 = nullptr; float* d_coarse = nullptr;
  float* d_cb1L = nullptr;  // cb1 line-part-major [LP][C1][SS] (coalesced reads of the table kernel)
  float* d_cb2T = nullptr;  // cb2 re-tiled per cell as [S/4][C2] 16-byte vectors (coalesced row walks), when S % 4 == 0
  bool haveTree = false;
  // heuristic prefix (a3)
  std::vector<uint32_t> heurHost; uint64_t heurRows = 0; uint16_t* d_heur = nullptr; uint16_t* d_heur8 = nullptr; uint32_t* d_heur4 = nullptr;
  // optional 2-D anisotropic traversal heuristic (pqt_index_build_heuristic_2d): the 10 cell orders [10][65536], their grid width,
  // the slope boundaries, and the per-query row tables of the current chunk (scratch of this handle)
  bool heur2d = false; uint32_t* d_seq2d = nullptr; uint32_t seq2dDc = 0; float slopeThr[9] = {0}; uint4* d_heurQ = nullptr; uint64_t heurQCap = 0;
  uint64_t maxMultiIndex = 0;
  // bin store (a5)
  PqtBinEntry* d_table = nullptr; uint32_t* d_lower = nullptr; uint32_t tableBits = 0;
  uint32_t* d_ids = nullptr; uint64_t nIds = 0; uint32_t maxBin = 0; bool sharded = false; bool haveBins = false;
  uint64_t nTotal = 0;  // database size (all shards)
  // line codes (a7)
  uint32_t* d_codes = nullptr; bool codesOwned = false; uint64_t nCodes = 0; uint64_t idBase = 0;  // as handed over (id order)
  uint32_t* d_codesBin = nullptr; bool binOrdered = false; bool linesDropped = false;
  float* d_bias = nullptr; bool biasReady = false; bool adcBias = false; bool exactFilter = true; float coarseMax = 0.f;
  unsigned long long* d_runs = nullptr; uint32_t* d_runGpos = nullptr; uint32_t* d_nRuns = nullptr; uint64_t runsCap = 0; int useRuns = -1 /* -1 auto, 0 off, 1 on where supported */; bool lastRuns = false; uint32_t curRunCap = 0;
  uint32_t* d_fbList = nullptr; uint32_t* d_fbCount = nullptr; bool lastFilter = false;  // MODE 2 fallback list
  uint32_t* d_tvList = nullptr; uint32_t* d_tvCount = nullptr;  // pqt_query_shard_bins: queries whose exchanged bin list overflowed (traversed here)
   // opt-in adc_bias mode: per-row query-independent part of the ADC sum
  uint32_t* d_codesX = nullptr; int xcodeShift = 0; int useXCode = -1 /* -1 auto, 0 off, 1 on where supported */; bool curXCode = false;  // X-code copy (pqt_k_xcode) for the LDS-table rerank
  uint32_t* d_codesGrp = nullptr; int grpG = 0;  // optional group-major copy [LP/G][nIds][G] for the workgroup-per-query rerank kernel  // bin-ordered copy the kernels read (row pos = code of ids[pos])
  // scratch arena
  float* d_qL1virt = nullptr; float* d_segD = nullptr; uint32_t* d_segBin = nullptr; uint32_t qCap = 0;
  uint32_t* d_cand = nullptr; float* d_candDist = nullptr; uint32_t* d_candPos = nullptr; uint64_t candCap = 0;
  uint32_t* d_nCand = nullptr; uint32_t* d_nLocal = nullptr; uint32_t* d_nIncl = nullptr;
  hipEvent_t lev0 = nullptr, lev1 = nullptr;  // start/stop events attached to the next fused launch (lean timing), or null
  bool curRuns = false;  // the current chunk hands bin runs (not a candidate list) from the traversal to the rerank
  uint32_t curDynamic = 0; unsigned long long* curZero8 = nullptr; uint32_t* curPool = nullptr; uint32_t* curPoolNext = nullptr; uint32_t poolPos = 0; unsigned long long* d_schedList = nullptr; uint64_t schedCapQ = 0; uint32_t curSchedCap = 0;  // rerank schedule and next statistics block of the current chunk
  // shared-row pass of the filtered rerank (pqt_shared_rows.h; scratch of this handle): per-batch bin table, pair records, block sums, items
  uint32_t* d_srTable = nullptr; uint64_t srTableCap = 0; uint32_t* d_srPairs = nullptr; uint64_t srPairCap = 0; uint32_t* d_srBlocks = nullptr; uint64_t srBlockCap = 0;
  unsigned long long* d_srItems = nullptr; uint64_t srItemCap = 0; unsigned long long* d_srKeys = nullptr; uint64_t srKeysCap = 0; const uint32_t* curPreOk = nullptr; const float* curPreQmax = nullptr;
  int sharedRows = -1 /* -1 auto, 0 off, 1 on where supported */; bool lastShared = false;
  const uint32_t* curPreFlags = nullptr;  // PqtSrArgs::total of the pass just launched ([2]: capacity flag)
  uint32_t srSlotBits = 0, srProbes = 128; bool srStats = false; const unsigned long long* srStatPtr = nullptr;  // test / measurement knobs of the pass (pqt_index_set_option)
  unsigned long long* d_srSeg = nullptr; uint64_t srSegCap = 0; int srScanSplit = 1, srScanDepth = 4;  // opt-in range scan of the selection (pqt_k_sr_scan_seg / pqt_k_sr_merge)
  int coopRerank = 0; uint32_t* d_coopErr = nullptr; bool lastCoop = false;  // opt-in cooperative filter scan (pqt_k_pair_scan); its give-up flag (never cleared: pqt_get_stats reports it)
  int srKernel = 1;  // evaluating kernel of the pass: 1 pqt_k_sr_adc (one table per query), 2 pqt_k_sr_adc2 (pair-interleaved tables, decode hoisted)
  uint32_t* d_filter1 = nullptr; uint32_t filter1Bits = 0; int useFilter1 = -1 /* -1 auto, 0 off, 1 on */;  // first level of the presence bitmap, folded for the LDS (wide enumeration)
  uint32_t* d_filter = nullptr; uint32_t filterBits = 0;  // presence bitmap over the bin keys (the fused traversal probes it first)
  uint32_t* d_ovList = nullptr; uint32_t* d_ovCount = nullptr;  // queries deferred to the full-size bins pass; [0] list length, [1] append cursor
  uint64_t* d_sortKeys = nullptr; uint64_t sortCap = 0;
  unsigned long long* d_counters = nullptr;  // kCtrRing blocks of 8 statistics words (one per call, the next one is zeroed on the fly) + 1 spare block
  unsigned long long* ctr = nullptr; int ctrPos = 0;
  unsigned long long* d_tstamp = nullptr;    // optional per-query phase timestamps (debug)
  uint64_t stride = 0;
  // results of the last call
  pqt_stats stats{};
  uint32_t lastQn = 0; uint32_t lastHe = 0; bool lastSegKept = false; bool lastDistKept = false;
  hipEvent_t evRing[kRing][kMaxChunks][EV_COUNT]{}; int ringChunks[kRing]{}; int ringPos = 0; unsigned long long calls = 0;
  uint32_t evMask[kRing][kMaxChunks]{};      // which events of a ring slot were recorded (an event record costs ~5 us of stream time)
  int nChunks = 0; bool evCreated = false;
  size_t scratchBudget = (size_t)24 << 30;
  // persistent staging buffers of the host-pointer entry point (pqt_query_host): grown on demand, never freed per call
  float* h2dQ = nullptr; uint32_t* h2dI = nullptr; float* h2dD = nullptr; uint32_t* h2dC = nullptr; size_t h2dQCap = 0, h2dKCap = 0, h2dCCap = 0;
  // overlapped halves (pqt_index_set_option "overlap"): a view handle shares every read-only array of this index and owns its own
  // scratch, stream and counters; a large batch is split in pieces that run on their own streams, each rerank launch on its share of the workgroup
  // slots, so that one piece's (latency-bound) traversal and the tail of its rerank launch fill the gaps of the others'
  static constexpr int kMaxViews = 3;
  pqt_index* views[kMaxViews] = {nullptr, nullptr, nullptr}; pqt_index* owner = nullptr; bool isView = false; int overlap = -1 /* -1 auto, 0 off, 1 on, n >= 2: n pieces */;
  hipStream_t padStream = nullptr; hipEvent_t evPadFork = nullptr, evPadJoin = nullptr; bool padSide = true;  // k > 128: padding of the result rows on a side stream
  bool userView = false;                 // created by pqt_index_create_view: refreshed from the owner at every call
  std::vector<pqt_index*> userViews;     // views handed to the caller (destroyed with the owner at the latest)
  hipEvent_t evFork = nullptr, evJoin[kMaxViews] = {nullptr, nullptr, nullptr};
  uint32_t lastPieces = 0, pieceStart[kMaxViews + 2] = {0, 0, 0, 0, 0};  // last call: pieces (0: one piece on this handle); piece i = queries [pieceStart[i], pieceStart[i+1]), piece 0 on this handle, piece i >= 1 on views[i-1]
  bool poolDirty = false;  // a traversal registered queries in the current pool block and no rerank launch has consumed (and re-zeroed) them yet
  bool lastTravF1 = false;  // the last fused traversal ran pqt_k_traverse_f1 (LDS first level of the presence bitmap)
  std::string lastPath;    // kernel variants of the last query call (pqt_get_last_path)
  int oneLaunch = -1;  // SIFT1M shape: traversal + rerank of a query by the same wavefront in ONE launch (pqt_k_query_fused): 1 on, 0 / -1 (default) off -- measured slower than the two launches
  bool smallLists = true;  // 128 < k <= 4096: lists of <= 1024 candidates go through the wave-per-query evaluate + sort kernel
  int numCUs = 256; bool forceUnfused = false; bool useWgRerank = true; int balance = -1 /* auto */; int stageTiming = 1; bool timedCall = true; unsigned long long timingPhase = 0; bool noShape = false; uint32_t dbg = 0;
};

inline int setDevice(const pqt_index* idx) {
  HIPCHK(hipSetDevice(idx->device));
  return PQT_OK;
}

template <class T>
int devAlloc(T** p, size_t n) {
  if (*p) { (void)hipFree(*p); *p = nullptr; }
  if (n == 0) n = 1;
  HIPCHK(hipMalloc((void**)p, n * sizeof(T)));
  return PQT_OK;
}

constexpr size_t kMaxLds = 160 * 1024;
// the dynamic-LDS ceiling of a kernel is raised once per (device, kernel, size), not on every query call (pqt_hip.hip)
int pqtAllowLds(const void* kernel, size_t bytes);
template <class K>
int allowLds(K kernel, size_t bytes) { return pqtAllowLds((const void*)kernel, bytes); }

inline uint32_t* poolBlock(pqt_index* idx, uint32_t pos) {
  return reinterpret_cast<uint32_t*>(idx->d_counters + 8 * (kCtrRing + 1)) + (size_t)(pos % kPoolRing) * kPoolWords;
}

// ---- pqt_hip.hip
int ensureGroupMajor(pqt_index* idx, int G);
int ensureXCode(pqt_index* idx, int c1Shift);

// ---- pqt_rerank_launch.hip: fused rerank + select launchers (one entry per kernel family; the template dispatch lives there)
int launchRerankSelect(pqt_index* idx, bool cl, uint32_t grid, size_t lds, hipStream_t st, const float* v, const uint32_t* nl,
                       uint64_t stride, uint32_t k, uint32_t nq, uint32_t* oI, float* oD, uint32_t* oP);
// MODE 1 (adc_bias distances) / MODE 2 (reference distances through the MODE 1 filter) kernels: nw = 12 | 6 wavefronts
int launchRSBiasAny(pqt_index* idx, int nw, bool filter, uint32_t grid, size_t lds, hipStream_t st, const float* v, const uint32_t* nl,
                    uint64_t stride, uint32_t k, uint32_t nq, uint32_t* oI, float* oD, uint32_t* oP);
int rswgGroup(const PqtDevParams& d);
int launchRSWGAny(pqt_index* idx, int G, uint32_t nq, hipStream_t st, const float* v, const uint32_t* nl, uint64_t stride, uint32_t k,
                  uint32_t* oI, float* oD, uint32_t* oP);
// 128 < k <= 4096: wave-per-query evaluate + sort of the short lists, block-wide select of the (listed) rest
int launchSmallLists(pqt_index* idx, bool cl, size_t lds, uint32_t grid, hipStream_t st, const PqtRsArgs& sa, hipEvent_t ev0);
int launchMidLists(pqt_index* idx, size_t lds, uint32_t grid, hipStream_t st, const PqtRsArgs& sa, uint32_t* outList, uint32_t* outCount);
int launchBigK(pqt_index* idx, bool cl, size_t lBig, uint32_t nq, hipStream_t st, const float* v, const uint32_t* nl, uint64_t stride, uint32_t k,
               uint32_t kP2, uint32_t kcap, uint32_t* oI, float* oD, uint32_t* oP, const uint32_t* qlist, const uint32_t* qcount, hipEvent_t ev0, hipEvent_t ev1);

// ---- pqt_shared_launch.hip: shared-row pass (pqt_shared_rows.h) in front of the filtered selection, and that selection reading its distances
bool sharedRowsShape(const pqt_index* idx);
int launchSharedRows(pqt_index* idx, hipStream_t st, const float* qL1virt, const uint32_t* nLocal, uint64_t stride, uint32_t nq, hipEvent_t ev0);
// the queries a filtered selection handed back (fbList): exact distances by whole workgroups + the exact selection over them
int launchHandedBack(pqt_index* idx, hipStream_t st, const PqtRsArgs& rargs);
int launchCoopRerank(pqt_index* idx, hipStream_t st, const float* v, const uint32_t* nl, uint64_t stride, uint32_t k, uint32_t nq, uint32_t* oI, float* oD, uint32_t* oP);
int launchSharedSelect(pqt_index* idx, uint32_t grid, size_t lds, hipStream_t st, const float* v, const uint32_t* nl,
                       uint64_t stride, uint32_t k, uint32_t nq, uint32_t* oI, float* oD, uint32_t* oP);

// ---- pqt_traverse_launch.hip: fused traversal (pqt_k_traverse): LDS plan and launch
struct TravPlan { bool fused = false, wide = false, p2 = false; size_t lTrav = 0; uint32_t perWave = 0; };
int planTraversal(pqt_index* idx, uint32_t He, TravPlan& tp);
int travShape(const pqt_index* idx, const PqtTravArgs& targs);
void launchFusedTraversal(pqt_index* idx, const PqtTravArgs& targs, const TravPlan& tp, uint32_t waves, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1);
size_t queryFusedPerWave(const pqt_index* idx, const TravPlan& tp);
bool queryFusedShape(const pqt_index* idx);
int launchQueryFused(pqt_index* idx, const PqtTravArgs& targs, const TravPlan& tp, uint32_t grid, hipStream_t st, const float* qL1virt,
                     const uint32_t* nLocal, uint64_t stride, uint32_t k, uint32_t nq, uint32_t* oI, float* oD, hipEvent_t ev0, hipEvent_t ev1);
