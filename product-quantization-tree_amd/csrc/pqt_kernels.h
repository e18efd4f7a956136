// pqt_kernels.h -- the gfx950 kernels of the PQT query hot path (and the offline encode kernel).
// Included once by pqt_hip.hip.  Stage names follow SURVEY.md §8a (a1..a10).
//
// All floating-point here is f32 with the exact association of the reference's cpu_version source,
// summed left to right with separate multiply and add (the TU is built with -ffp-contract=off), so
// tables, bin distances and ADC distances are bit-identical to the oracle and every sort sees the
// same keys.  Sorts are by (key, original position): the result equals any comparison sort of the
// reference whenever no two keys are exactly equal, and is the stable order otherwise.
#pragma once
#include <type_traits>
#include "pqt_device.h"
#include "pqt_wave.h"

#define PQT_BLOCK 256
#define PQT_TS_WORDS 24  // debug timestamp words per query (PQT_TSTAMP=1): 0-8 traversal phases, 9-14 rerank, 15 traversal hw id, 16-19 rerank detail

// ---------------------------------------------------------------------------------------------------
// setup (a9): coarse[(lp*C1 + i)*C1 + j] = || cb1[i]_lp - cb1[j]_lp ||^2     (treequantizer.hpp:183-203)
// The reference evaluates (AV - BV) with i <= j and mirrors; x-y and y-x square identically, so
// evaluating every (i,j) directly gives the same bits.
// ---------------------------------------------------------------------------------------------------
#ifdef PQT_MAIN_TU
__global__ void pqt_k_coarse(const float* __restrict__ cb1, float* __restrict__ coarse, PqtDevParams prm) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t total = prm.LP * prm.C1 * prm.C1;
  if (t >= total) return;
  const uint32_t j = t % prm.C1, i = (t / prm.C1) % prm.C1, lp = t / (prm.C1 * prm.C1);
  const uint32_t a = i < j ? i : j, b = i < j ? j : i;  // AV = row min, BV = row max, like the reference
  const float* x = cb1 + (size_t)a * prm.D + lp * prm.SS;
  const float* y = cb1 + (size_t)b * prm.D + lp * prm.SS;
  float s = 0.f;
  for (uint32_t d = 0; d < prm.SS; ++d) { const float df = x[d] - y[d]; s = s + df * df; }
  coarse[t] = s;
}
#endif  // PQT_MAIN_TU

typedef float pqt_f2 __attribute__((ext_vector_type(2)));

// One statistics atomic per WAVEFRONT.  The tie counters (pqt_stats.ties_*) were bumped by every lane that saw a tie, each with its own
// device-scope atomic on the same word: on the SIFT-shaped data a result list holds ~45 equal-distance neighbours, i.e. ~450 k
// same-address atomics per 10 k-query launch, and they serialise at the memory side -- 0.058 of the rerank launch's 0.157 ms at the
// SIFT1M shape (found because two half-size launches on two handles, i.e. two counter words, ran 1.4x faster than one launch).
// The persistent kernels accumulate in a register across their queries and call this once per wavefront.
// All lanes of the wavefront must be active.
__device__ __forceinline__ void pqt_count_ties(unsigned long long* ctr, uint32_t ties) {
  if (__any(ties != 0)) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) ties += (uint32_t)__shfl_xor((int)ties, d, 64);
    if ((threadIdx.x & 63u) == 0) atomicAdd(ctr, (unsigned long long)ties);
  }
}

// ---------------------------------------------------------------------------------------------------
// stage a1 + a2: per-query distance tables and the sorted second-level entry lists.
//   one workgroup per query.
//   a1  L1virt[lp][c] = ||q_lp - cb1[c]_lp||^2 ; L1[p][c] = sum_pp L1virt[p*R+pp][c]   (treequantizer.hpp:640-661)
//       W nearest cells per part in ascending order                                      (:663-671)
//   a2  for the W cells: d2[h1*C2+h2] = ||q_p - cb2[p][c1][h2]||^2, sorted by d2         (:597-630, vectorquantizer.hpp:104-115)
// outputs: qL1virt[q][LP*C1] ; segD[q][P][WC] ascending ; segBin[q][P][WC] = c1*C2+h2 in the same order
// LDS: D + LP*C1 + P*C1 + P*W + P*WC words.
// ---------------------------------------------------------------------------------------------------
#ifdef PQT_MAIN_TU
__global__ __launch_bounds__(PQT_BLOCK) void pqt_k_tables(
    const float* __restrict__ Q, const float* __restrict__ cb1, const float* __restrict__ cb2, PqtDevParams prm,
    float* __restrict__ qL1virt, float* __restrict__ segD, uint32_t* __restrict__ segBin,
    unsigned long long* __restrict__ counters) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const uint32_t D = prm.D, P = prm.P, C1 = prm.C1, C2 = prm.C2, W = prm.W, LP = prm.LP, S = prm.S, SS = prm.SS,
                 R = prm.R, WC = prm.WC;
  float* sQ = smem;
  float* sVirt = sQ + D;
  float* sL1 = sVirt + LP * C1;
  uint32_t* sOrd = (uint32_t*)(sL1 + P * C1);
  float* sD2 = (float*)(sOrd + P * W);
  const uint32_t q = blockIdx.x, tid = threadIdx.x;

  for (uint32_t i = tid; i < D; i += PQT_BLOCK) sQ[i] = Q[(size_t)q * D + i];
  __syncthreads();

  // a1: one accumulator per (centroid, line part); t = c*LP + lp walks cb1 contiguously
  for (uint32_t t = tid; t < C1 * LP; t += PQT_BLOCK) {
    const uint32_t c = t / LP, lp = t % LP;
    const float* cen = cb1 + (size_t)c * D + lp * SS;
    const float* qq = sQ + lp * SS;
    float s = 0.f;
    for (uint32_t d = 0; d < SS; ++d) { const float df = qq[d] - cen[d]; s = s + df * df; }
    sVirt[lp * C1 + c] = s;
  }
  __syncthreads();
  for (uint32_t t = tid; t < LP * C1; t += PQT_BLOCK) qL1virt[(size_t)q * LP * C1 + t] = sVirt[t];
  for (uint32_t t = tid; t < P * C1; t += PQT_BLOCK) {
    const uint32_t p = t / C1, c = t % C1;
    float d = 0.f;
    for (uint32_t pp = 0; pp < R; ++pp) d = d + sVirt[(p * R + pp) * C1 + c];
    sL1[t] = d;
  }
  __syncthreads();
  // rank of each cell inside its part by (distance, index): the W smallest land in sOrd in order
  uint32_t ties = 0;
  for (uint32_t t = tid; t < P * C1; t += PQT_BLOCK) {
    const uint32_t p = t / C1, c = t % C1;
    const float my = sL1[t];
    uint32_t rank = 0;
    for (uint32_t o = 0; o < C1; ++o) {
      const float v = sL1[p * C1 + o];
      rank += (v < my) || (v == my && o < c);
      ties += (v == my && o < c);
    }
    if (rank < W) sOrd[p * W + rank] = c;
  }
  pqt_count_ties(&counters[0], ties);
  __syncthreads();
  // a2: second-level distances of the W expanded cells
  for (uint32_t t = tid; t < P * WC; t += PQT_BLOCK) {
    const uint32_t p = t / WC, pos = t % WC, h1 = pos / C2, h2 = pos % C2;
    const uint32_t c1 = sOrd[p * W + h1];
    const float* cen = cb2 + (((size_t)p * C1 + c1) * C2 + h2) * S;
    const float* qq = sQ + p * S;
    float s = 0.f;
    for (uint32_t d = 0; d < S; ++d) { const float df = qq[d] - cen[d]; s = s + df * df; }
    sD2[t] = s;
  }
  __syncthreads();
  ties = 0;
  for (uint32_t t = tid; t < P * WC; t += PQT_BLOCK) {
    const uint32_t p = t / WC, pos = t % WC, h1 = pos / C2, h2 = pos % C2;
    const float my = sD2[t];
    uint32_t rank = 0;
    for (uint32_t o = 0; o < WC; ++o) {
      const float v = sD2[p * WC + o];
      rank += (v < my) || (v == my && o < pos);
      ties += (v == my && o < pos);
    }
    const size_t base = ((size_t)q * P + p) * WC;
    segD[base + rank] = my;
    segBin[base + rank] = sOrd[p * W + h1] * C2 + h2;
  }
  pqt_count_ties(&counters[1], ties);
}
#endif  // PQT_MAIN_TU

// ---------------------------------------------------------------------------------------------------
// Optional traversal heuristic "2-D anisotropic sequences" (SURVEY 8f-4): WHICH rows a query enumerates, the way the CUDA
// library's 1B path picks them (pqt/PerturbationProTree.cu: computeSlopeIdx :2839-2858, generate2DBins :2888-2910,
// selectBinKernel2D2Parts :2914-3006, selectBinKernel2DFinal :3012-3100; tables: ProTree::prepare2DDistSequence,
// pqt/ProTree.cu:50-126).  P = 4.  Parts (0,1) and (2,3) are merged first: with the sorted second-level distances v0, v1 of
// the two parts (first kMax = min(64, W*C2) entries each) a slope (v1[22]+v1[21]-2 v1[0]) / (v0[22]+v0[21]-2 v0[0]) picks one
// of 10 precomputed orders of the (x, y) grid (key x^0.8 + s y^0.8), whose first 256 cells -- those with x, y < kMax -- give
// the pair list, sorted by v0[x] + v1[y].  The two pair lists are merged the same way (slope sampled at 45 / 44): row r of the
// query is cell r of the chosen order over the 256 x 256 grid of pair-list ranks, i.e. the tuple of four part ranks
// (pair0[x].x, pair0[x].y, pair1[y].x, pair1[y].y).  Cells that leave the lists, or land on padding, are rows WITHOUT a bin
// (digit 0 = 0xffff; the CUDA kernel gives them distance 99999999999 and bin 0).
// Only the SET of enumerated rows changes: pqt_k_bins then forms bin ids and distances from the tuples exactly as for the
// shared table (cpu_version association and uint32 wrap), orders all rows by (distance, row) and applies the cpu cut -- the
// CUDA kernel's order inside 1024-row chunks, its 2-vectors-per-bin cap and its stop at k vectors are NOT reproduced.
// Differences from the CUDA text, on purpose: ties inside a pair list are ordered by cell number (bitonic3 leaves them in
// network order); the slope index is found by comparing against the 9 boundaries 1.2^(j - 4.5) (computed once on the host)
// instead of roundf(logf(slope) / logf(1.2)) + 5, so that device and checker agree bit for bit (a NaN / negative slope gives
// index 0 as the clamped CUDA expression does); sample positions are clamped to the list length.
// One wavefront per query, 4 per workgroup; rows[q][He] = 8 x u16 digits like the shared table.
#ifdef PQT_MAIN_TU
struct PqtRows2dArgs {
  const float* segD; const uint32_t* seq; uint4* rows;
  uint32_t dc, WC, kMax, He, nq;
  float thr[9];
};
__global__ __launch_bounds__(256) void pqt_k_rows_2d(const PqtRows2dArgs A) {
  __shared__ uint32_t sXY[4][2][256];
  __shared__ float sDist[4][2][256];
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t q = blockIdx.x * 4 + wave;
  if (q >= A.nq) return;  // (no workgroup barrier below)
  const float* sd = A.segD + (size_t)q * 4 * A.WC;
  auto slopeIdx = [&](const float a1, const float b1, const float c1, const float a0, const float b0, const float c0) -> uint32_t {
    const float slope = (a1 + b1 - 2.f * c1) / (a0 + b0 - 2.f * c0);
    uint32_t si = 0;
#pragma unroll
    for (int j = 0; j < 9; ++j) si += (slope >= A.thr[j]) ? 1u : 0u;
    return si;
  };
  const uint32_t s1 = A.kMax - 1 < 22u ? A.kMax - 1 : 22u, s1m = s1 ? s1 - 1 : 0u;
  for (uint32_t j = 0; j < 2; ++j) {
    const float* v0 = sd + (size_t)(2 * j) * A.WC;
    const float* v1 = v0 + A.WC;
    const uint32_t si = slopeIdx(v1[s1], v1[s1m], v1[0], v0[s1], v0[s1m], v0[0]);
    const uint32_t* sq = A.seq + (size_t)si * 65536;
    uint64_t key[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint32_t t = lane * 4 + r;
      const uint32_t c = sq[t], x = c % A.dc, y = c / A.dc;
      const bool in = x < A.kMax && y < A.kMax;
      const float dist = in ? v0[in ? x : 0] + v1[in ? y : 0] : 99999999999.f;
      key[r] = ((uint64_t)pqt_f2key(dist) << 32) | (t << 16) | (in ? (y << 8 | x) : 0xffffu);
    }
    pqt_wave_sort_u64<4>(key);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      sXY[wave][j][lane * 4 + r] = (uint32_t)key[r] & 0xffffu;
      sDist[wave][j][lane * 4 + r] = pqt_key2f((uint32_t)(key[r] >> 32));
    }
  }
  __builtin_amdgcn_wave_barrier();
  const float* l0 = sDist[wave][0];
  const float* l1 = sDist[wave][1];
  const uint32_t si2 = slopeIdx(l1[45], l1[44], l1[0], l0[45], l0[44], l0[0]);
  const uint32_t* sq = A.seq + (size_t)si2 * 65536;
  uint4* out = A.rows + (size_t)q * A.He;
  for (uint32_t r = lane; r < A.He; r += 64) {
    const uint32_t c = sq[r], x = c % A.dc, y = c / A.dc;
    uint4 row = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
    if (x < 256 && y < 256) {
      const uint32_t a = sXY[wave][0][x], b = sXY[wave][1][y];
      if (a != 0xffffu && b != 0xffffu) row = make_uint4((a & 0xffu) | ((a >> 8) << 16), (b & 0xffu) | ((b >> 8) << 16), 0u, 0u);
    }
    out[r] = row;
  }
}
#endif  // PQT_MAIN_TU

// ---------------------------------------------------------------------------------------------------
// stage a4 + a5 + a6 (staged structure): bin enumeration, probe, exact ordering, cut and candidate gather.
//   one workgroup per query.
//   a4  for h < He: dist_h = sum_p segD[p][heur[h][p]] ; glob_h = sum_p segBin[p][..]*powers[p] (uint32 wrap)
//       then order the bins by dist_h                                                  (treequantizer.hpp:548-588)
//   a5  probe the bin table for (start, population)                                    (:462-463, std::map lookup)
//   a6  visit bins in order, take whole bins, stop after the bin during which the running count
//       exceeded Bv (strict >)                                                         (:450-477)
// Only POPULATED bins take part in the ordering: an empty bin adds nothing to the running count, to the candidate
// list or to any visiting position, so dropping it before the sort leaves every result unchanged.  Populated bins
// are collected as compact LDS entries {key = (f32 key << 32 | row << 13 | entry), rec = (gcount, lstart[, lcount,
// lower])}.  The LDS arena holds `cap` entries: the first launch uses a small cap (high occupancy); a query with more
// populated bins than that appends itself to `ovList` and is redone by a second launch of the same kernel with
// cap = He (qlist = ovList), before the rerank stage runs.
// outputs: cand[q*stride + j] = position in the bin-ordered line store of the j-th candidate in visiting order (local
//          members only when sharded), candPos (sharded only) = global visiting position, nCand[q] = GLOBAL candidate
//          count, nLocal[q] = local candidate count, nIncl[q] = included populated bins.
// ---------------------------------------------------------------------------------------------------
template <bool SHARDED>
__global__ __launch_bounds__(PQT_BLOCK) void pqt_k_bins(
    const float* __restrict__ segD, const uint32_t* __restrict__ segBin, const uint16_t* __restrict__ heur,
    uint32_t He, uint32_t cap, uint32_t capP2, uint32_t Bv, PqtDevParams prm, const PqtBinEntry* __restrict__ table,
    const uint32_t* __restrict__ lower, uint32_t tableBits, uint32_t* __restrict__ cand, uint32_t* __restrict__ candPos,
    uint32_t* __restrict__ nCand, uint32_t* __restrict__ nLocal, uint32_t* __restrict__ nIncl, uint64_t stride,
    const uint32_t* __restrict__ qlist, const uint32_t* __restrict__ qcount, uint32_t* __restrict__ ovList,
    uint32_t* __restrict__ ovCount, unsigned long long* __restrict__ counters,
    uint32_t* __restrict__ schedCnt, unsigned long long* __restrict__ schedList, uint32_t schedCap /* rerank schedule 2: see PqtTravArgs */,
    uint64_t heurStride /* 0: one table for all queries; otherwise rows per query of a per-query table (pqt_k_rows_2d), whose rows with
                           digit 0 = 0xffff name no bin */) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  if (qlist && blockIdx.x >= *qcount) return;
  const uint32_t q = qlist ? qlist[blockIdx.x] : blockIdx.x;
  heur += (size_t)q * heurStride * 8;
  const uint32_t P = prm.P, WC = prm.WC;
  constexpr uint32_t RW = SHARDED ? 4 : 2;               // words per record
  uint64_t* sKey = (uint64_t*)smem_raw;                  // capP2
  uint32_t* sRec = (uint32_t*)(sKey + capP2);            // cap * RW : gcount, lstart [, lcount, lower]
  float* sSegD = (float*)(sRec + (size_t)cap * RW);      // P*WC
  uint32_t* sSegB = (uint32_t*)(sSegD + P * WC);         // P*WC
  uint32_t* sPart = sSegB + P * WC;                      // PQT_BLOCK/64 + 1
  uint32_t* sMisc = sPart + (PQT_BLOCK / 64 + 1);        // 4 : [0] included bins [1] candidates [2] local candidates [3] populated bins
  const uint32_t tid = threadIdx.x;

  for (uint32_t t = tid; t < P * WC; t += PQT_BLOCK) {
    sSegD[t] = segD[(size_t)q * P * WC + t];
    sSegB[t] = segBin[(size_t)q * P * WC + t];
  }
  if (tid == 0) { sMisc[0] = 0; sMisc[1] = 0; sMisc[2] = 0; sMisc[3] = 0; }
  __syncthreads();
  // 4 rows per thread in flight: heuristic-row reads, then 8 table probes, overlap instead of chaining
  for (uint32_t h0 = tid; h0 < He; h0 += PQT_BLOCK * 4) {
    uint4 hv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const uint32_t h = h0 + u * PQT_BLOCK; hv[u] = reinterpret_cast<const uint4*>(heur)[h < He ? h : h0]; }
    uint32_t glob[4]; float fine[4];
    bool nobin[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      nobin[u] = (hv[u].x & 0xffffu) == 0xffffu;
      if (nobin[u]) hv[u] = make_uint4(0u, 0u, 0u, 0u);
      const uint32_t dg[8] = {hv[u].x & 0xffffu, hv[u].x >> 16, hv[u].y & 0xffffu, hv[u].y >> 16, hv[u].z & 0xffffu, hv[u].z >> 16, hv[u].w & 0xffffu, hv[u].w >> 16};
      float f = 0.f; uint32_t g = 0;
#pragma unroll
      for (int p = 0; p < PQT_MAXP; ++p) {
        if ((uint32_t)p < P) { f = f + sSegD[p * WC + dg[p]]; g += sSegB[p * WC + dg[p]] * prm.powers[p]; }
      }
      if (prm.hashMod) g %= prm.hashMod;
      glob[u] = g; fine[u] = f;
    }
    uint4 e[4]; uint32_t slot[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) e[u] = pqt_table_lookup(reinterpret_cast<const uint4*>(table), glob[u], tableBits, prm.tableSeed, &slot[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t h = h0 + u * PQT_BLOCK;
      if (h < He && e[u].y && !nobin[u]) {
        const uint32_t ent = atomicAdd(&sMisc[3], 1u);
        if (ent < cap) {
          sRec[ent * RW] = e[u].y; sRec[ent * RW + 1] = e[u].z;
          if (SHARDED) { sRec[ent * RW + 2] = e[u].w; sRec[ent * RW + 3] = lower[slot[u]]; }
          sKey[ent] = ((uint64_t)pqt_f2key(fine[u]) << 32) | (h << 13) | ent;  // ordered by (distance, row); ent addresses the record
        }
      }
    }
  }
  __syncthreads();
  const uint32_t nEnt = sMisc[3];  // populated bins among the enumerated rows
  if (nEnt > cap) {  // does not fit this launch's arena: hand the query to the full-size pass
    if (tid == 0) { ovList[atomicAdd(ovCount, 1u)] = q; nCand[q] = 0; nLocal[q] = 0; nIncl[q] = 0; }
    return;
  }
  uint32_t entP2 = 2;
  while (entP2 < nEnt) entP2 <<= 1;
  for (uint32_t i = nEnt + tid; i < entP2; i += PQT_BLOCK) sKey[i] = ~0ull;
  __syncthreads();
  pqt_bitonic_sort_u64<PQT_BLOCK>(sKey, entP2);

  // exclusive scan of the global populations in visiting order; each thread owns a contiguous chunk
  const uint32_t per = (nEnt + PQT_BLOCK - 1) / PQT_BLOCK;
  const uint32_t i0 = tid * per < nEnt ? tid * per : nEnt, i1 = (i0 + per < nEnt) ? i0 + per : nEnt;
  uint32_t loc = 0, ties = 0;
  for (uint32_t i = i0; i < i1; ++i) {
    loc += sRec[((uint32_t)sKey[i] & 0x1fffu) * RW];
    if (i + 1 < nEnt && (uint32_t)(sKey[i] >> 32) == (uint32_t)(sKey[i + 1] >> 32)) ++ties;
  }
  pqt_count_ties(&counters[2], ties);
  uint32_t total;
  const uint32_t run = pqt_block_excl_scan<PQT_BLOCK>(loc, sPart, &total);
  // included bins = the prefix with exclusive count <= Bv (the count is non-decreasing)
  uint32_t myIncl = 0, myCand = 0, locIncl = 0;
  {
    uint32_t r = run;
    for (uint32_t i = i0; i < i1; ++i) {
      const uint32_t ent = (uint32_t)sKey[i] & 0x1fffu;
      const uint32_t g = sRec[ent * RW];
      if (r <= Bv) { ++myIncl; myCand = r + g; if (SHARDED) locIncl += sRec[ent * RW + 2]; }
      r += g;
    }
  }
  if (myIncl) { atomicAdd(&sMisc[0], myIncl); atomicMax(&sMisc[1], myCand); }
  uint32_t runL = run;
  if (SHARDED) {
    uint32_t totalL;
    runL = pqt_block_excl_scan<PQT_BLOCK>(locIncl, sPart, &totalL);
    if (tid == 0) sMisc[2] = totalL;
  }
  __syncthreads();
  const uint32_t nb = sMisc[0];
  const uint32_t nGlobal = sMisc[1];
  // rewrite the sorted keys as (start of the bin in the candidate list << 32 | entry) for the gather; the sharded
  // variant lists local members only and stashes the bin's GLOBAL start in record word 0 (the count is spent)
  {
    uint32_t r = run, rl = runL;
    for (uint32_t i = i0; i < i1; ++i) {
      const uint32_t ent = (uint32_t)sKey[i] & 0x1fffu;
      const uint32_t g = sRec[ent * RW];
      if (SHARDED) {
        sKey[i] = ((uint64_t)rl << 32) | ent;
        if (r <= Bv) rl += sRec[ent * RW + 2];
        sRec[ent * RW] = r;
      } else {
        sKey[i] = ((uint64_t)r << 32) | ent;
      }
      r += g;
    }
  }
  __syncthreads();
  const uint32_t nLoc = SHARDED ? sMisc[2] : nGlobal;
  if (tid == 0) {  // per-query outputs only: shared counters would serialise 10^4 workgroups on one L2 line
    nCand[q] = nGlobal;
    nLocal[q] = nLoc;
    nIncl[q] = nb;
    if (schedCnt) {  // registration for the rerank schedule (the fused traversal does the same in its finish step)
      const uint32_t slot = (q & 7u) * PQT_SCHED_CLASSES + pqt_sched_class(nLoc);
      const uint32_t pos = __hip_atomic_fetch_add(&schedCnt[slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (pos < schedCap) schedList[(size_t)slot * schedCap + pos] = (unsigned long long)q | ((unsigned long long)nLoc << 32);
    }
  }
  // a6 gather: candidate j lives in the last included bin whose list start is <= j
  for (uint32_t j = tid; j < nLoc; j += PQT_BLOCK) {
    uint32_t lo = 0, hi = nb;  // invariant: start[lo] <= j, answer in [lo, hi)
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if ((uint32_t)(sKey[mid] >> 32) <= j) lo = mid; else hi = mid;
    }
    const uint32_t ent = (uint32_t)sKey[lo];
    const uint32_t off = j - (uint32_t)(sKey[lo] >> 32);
    cand[(size_t)q * stride + j] = sRec[ent * RW + 1] + off;  // position in the bin-ordered line store (== index into ids[])
    if (SHARDED) candPos[(size_t)q * stride + j] = sRec[ent * RW] + sRec[ent * RW + 3] + off;
  }
}

// ---------------------------------------------------------------------------------------------------
// stage a7: ADC line rerank.  One workgroup per query, one lane per candidate; the lane streams the
// candidate's LP 4-byte codes (one 64/128-byte row, read as 16-byte vectors) and accumulates
//   sum_{p<LP} extractDistance(a = L1virt[p][B], b = L1virt[p][A], c = coarse[p][A][B], lambda)
// in p order -- treequantizer.hpp:423-439 + helper.hpp:132-136 -- so the sum is bit-identical.
// L1virt of the query sits in LDS; coarse[LP][C1][C1] is read through L2 (64 KB..512 KB, resident).
// ---------------------------------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(PQT_BLOCK) void pqt_k_rerank(
    const uint32_t* __restrict__ codes /* bin-ordered */, const float* __restrict__ qL1virt,
    const float* __restrict__ coarse, const uint32_t* __restrict__ cand, float* __restrict__ candDist,
    const uint32_t* __restrict__ nLocal, uint64_t stride, PqtDevParams prm) {
  extern __shared__ __attribute__((aligned(16))) float sVirt[];
  const uint32_t C1 = prm.C1, LP = prm.LP;
  const uint32_t q = blockIdx.x, tid = threadIdx.x;
  const uint32_t n = nLocal[q];
  if (n == 0) return;
  for (uint32_t t = tid; t < LP * C1; t += PQT_BLOCK) sVirt[t] = qL1virt[(size_t)q * LP * C1 + t];
  __syncthreads();
  for (uint32_t j = tid; j < n; j += PQT_BLOCK) {
    const uint32_t pos = cand[(size_t)q * stride + j];  // candidates are positions in the bin-ordered store
    const uint32_t* row = codes + (size_t)pos * LP;
    float acc = 0.f;
    if (VEC == 4) {
      const uint4* row4 = reinterpret_cast<const uint4*>(row);
      for (uint32_t p4 = 0; p4 < LP / 4; ++p4) {
        const uint4 v = row4[p4];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const uint32_t p = p4 * 4 + u;
          const uint32_t A = w[u] & 0xffu, B = (w[u] >> 8) & 0xffu;
          const float lam = pqt_lambda_decode(w[u] >> 16);
          const float sb = sVirt[p * C1 + A];
          const float sa = sVirt[p * C1 + B];
          const float sc = coarse[((size_t)p * C1 + A) * C1 + B];
          acc = acc + pqt_extract_distance(sa, sb, sc, lam);
        }
      }
    } else {
      for (uint32_t p = 0; p < LP; ++p) {
        const uint32_t w = row[p];
        const uint32_t A = w & 0xffu, B = (w >> 8) & 0xffu;
        const float lam = pqt_lambda_decode(w >> 16);
        const float sb = sVirt[p * C1 + A];
        const float sa = sVirt[p * C1 + B];
        const float sc = coarse[((size_t)p * C1 + A) * C1 + B];
        acc = acc + pqt_extract_distance(sa, sb, sc, lam);
      }
    }
    candDist[(size_t)q * stride + j] = acc;
  }
}

// ---------------------------------------------------------------------------------------------------
// stage a8: top-k of the candidate list by (distance, visiting position)   (treequantizer.hpp:479-483)
//   one workgroup per query.  Radix select of the k-th key (4 x 8-bit passes over the f32 keys), ordered
//   resolution of ties at the threshold, then a bitonic sort of the <= k survivors in LDS.
//   SHARDED: the tie-break position is candPos (global visiting position) and is written to outPos.
// LDS: 256 + 8 words histogram/misc, NP2(k) u64.
// ---------------------------------------------------------------------------------------------------
template <bool SHARDED>
__global__ __launch_bounds__(PQT_BLOCK) void pqt_k_select(
    const uint32_t* __restrict__ ids, const uint32_t* __restrict__ cand, const float* __restrict__ candDist, const uint32_t* __restrict__ candPos,
    const uint32_t* __restrict__ nLocal, uint64_t stride, uint32_t k, uint32_t kP2,
    uint32_t* __restrict__ outIdx, float* __restrict__ outDist, uint32_t* __restrict__ outPos,
    unsigned long long* __restrict__ counters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  uint64_t* sSel = (uint64_t*)smem_raw;        // kP2
  uint32_t* sHist = (uint32_t*)(sSel + kP2);   // 256
  uint32_t* sMisc = sHist + 256;               // 8
  uint32_t* sPart = sMisc + 8;                 // PQT_BLOCK/64+1
  const uint32_t q = blockIdx.x, tid = threadIdx.x;
  const uint32_t n = nLocal[q];
  const float* dist = candDist + (size_t)q * stride;
  const uint32_t* cid = cand + (size_t)q * stride;
  const uint32_t* cpos = SHARDED ? candPos + (size_t)q * stride : nullptr;
  const uint32_t kk = n < k ? n : k;  // number of real results

  for (uint32_t i = tid; i < kP2; i += PQT_BLOCK) sSel[i] = ~0ull;
  if (tid == 0) sMisc[0] = 0;
  __syncthreads();

  if (n <= k) {
    for (uint32_t j = tid; j < n; j += PQT_BLOCK) sSel[j] = ((uint64_t)pqt_f2key(dist[j]) << 32) | j;
  } else {
    // ---- radix select: find T = k-th smallest key, m = #{key < T}
    uint32_t prefix = 0, want = k;  // want-th smallest (1-based) among keys matching `prefix` on the bits fixed so far
    for (int pass = 0; pass < 4; ++pass) {
      const int shift = 24 - 8 * pass;
      sHist[tid] = 0;  // PQT_BLOCK == 256
      __syncthreads();
      const uint32_t himask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
      for (uint32_t j = tid; j < n; j += PQT_BLOCK) {
        const uint32_t key = pqt_f2key(dist[j]);
        if ((key & himask) == prefix) atomicAdd(&sHist[(key >> shift) & 0xffu], 1u);
      }
      __syncthreads();
      if (tid == 0) {
        uint32_t acc = 0, d = 0;
        for (; d < 256; ++d) { if (acc + sHist[d] >= want) break; acc += sHist[d]; }
        sMisc[1] = d; sMisc[2] = want - acc;
      }
      __syncthreads();
      prefix |= sMisc[1] << shift;
      want = sMisc[2];
      __syncthreads();
    }
    const uint32_t T = prefix;
    const uint32_t r = want;  // how many of the key == T group are needed (>= 1), in position order
    // ---- collect key < T (any order; the final sort fixes it)
    for (uint32_t j = tid; j < n; j += PQT_BLOCK) {
      const uint32_t key = pqt_f2key(dist[j]);
      if (key < T) { const uint32_t s = atomicAdd(&sMisc[0], 1u); sSel[s] = ((uint64_t)key << 32) | j; }
    }
    __syncthreads();
    const uint32_t m = sMisc[0];  // == k - r
    // ---- key == T: the r smallest tie-break positions.  Unsharded: position == j, so an ordered sweep
    // (chunks of PQT_BLOCK in j order with a block scan) accepts exactly the first r.  Sharded: positions
    // are increasing in j as well (local list order follows global visiting order), same sweep.
    uint32_t taken = 0;
    for (uint32_t base = 0; base < n && taken < r; base += PQT_BLOCK) {
      const uint32_t j = base + tid;
      const uint32_t flag = (j < n && pqt_f2key(dist[j]) == T) ? 1u : 0u;
      uint32_t tot;
      const uint32_t ex = pqt_block_excl_scan<PQT_BLOCK>(flag, sPart, &tot);
      if (flag && taken + ex < r) sSel[m + taken + ex] = ((uint64_t)T << 32) | j;
      taken += tot;
    }
    __syncthreads();
  }
  __syncthreads();
  // sort by (key, tie-break position).  Low word currently holds j; for the sharded case swap in the
  // global position for sorting, but keep j recoverable: positions are monotone in j, so sorting by j
  // is equivalent -- no swap needed.
  pqt_bitonic_sort_u64<PQT_BLOCK>(sSel, kP2);
  uint32_t ties = 0;
  for (uint32_t i = tid; i < k; i += PQT_BLOCK) {
    if (i < kk) {
      const uint32_t j = (uint32_t)sSel[i];
      outIdx[(size_t)q * k + i] = ids[cid[j]];
      outDist[(size_t)q * k + i] = dist[j];
      if (SHARDED) outPos[(size_t)q * k + i] = cpos[j];
      if (i + 1 < kk && (uint32_t)(sSel[i] >> 32) == (uint32_t)(sSel[i + 1] >> 32)) ++ties;
    } else {
      outIdx[(size_t)q * k + i] = 0xffffffffu;
      outDist[(size_t)q * k + i] = __uint_as_float(0x7f800000u);
      if (SHARDED) outPos[(size_t)q * k + i] = 0xffffffffu;
    }
  }
  pqt_count_ties(&counters[3], ties);
}

// full sort of every candidate list through a global-memory key buffer (parity / large-k path):
// keys[q][nP2] u64, one workgroup per query.
template <bool SHARDED>
__global__ __launch_bounds__(PQT_BLOCK) void pqt_k_fullsort(
    const uint32_t* __restrict__ ids, const uint32_t* __restrict__ cand, const float* __restrict__ candDist, const uint32_t* __restrict__ candPos,
    const uint32_t* __restrict__ nLocal, uint64_t stride, uint32_t k, uint64_t* __restrict__ keys, uint32_t nP2max,
    uint32_t* __restrict__ outIdx, float* __restrict__ outDist, uint32_t* __restrict__ outPos,
    unsigned long long* __restrict__ counters) {
  const uint32_t q = blockIdx.x, tid = threadIdx.x;
  const uint32_t n = nLocal[q];
  uint32_t nP2 = 1;
  while (nP2 < n) nP2 <<= 1;
  if (nP2 < 2) nP2 = 2;
  uint64_t* a = keys + (size_t)q * nP2max;
  const float* dist = candDist + (size_t)q * stride;
  for (uint32_t j = tid; j < nP2; j += PQT_BLOCK) a[j] = j < n ? (((uint64_t)pqt_f2key(dist[j]) << 32) | j) : ~0ull;
  __syncthreads();
  pqt_bitonic_sort_u64<PQT_BLOCK>(a, nP2);
  const uint32_t kk = n < k ? n : k;
  uint32_t ties = 0;
  for (uint32_t i = tid; i < k; i += PQT_BLOCK) {
    if (i < kk) {
      const uint32_t j = (uint32_t)a[i];
      outIdx[(size_t)q * k + i] = ids[cand[(size_t)q * stride + j]];
      outDist[(size_t)q * k + i] = dist[j];
      if (SHARDED) outPos[(size_t)q * k + i] = candPos[(size_t)q * stride + j];
      if (i + 1 < kk && (uint32_t)(a[i] >> 32) == (uint32_t)(a[i + 1] >> 32)) ++ties;
    } else {
      outIdx[(size_t)q * k + i] = 0xffffffffu;
      outDist[(size_t)q * k + i] = __uint_as_float(0x7f800000u);
      if (SHARDED) outPos[(size_t)q * k + i] = 0xffffffffu;
    }
  }
  pqt_count_ties(&counters[3], ties);
}

// ---------------------------------------------------------------------------------------------------
// multi-GPU merge: per query, the nsh*k gathered (dist, pos, id) triples -> first k by (dist, pos).
// Bitonic sort of slot indices with a (distance key, global visiting position) comparator in LDS.
// LDS: mP2 u32 indices + m u32 keys + m u32 positions.
// ---------------------------------------------------------------------------------------------------
#ifdef PQT_MAIN_TU
__global__ __launch_bounds__(PQT_BLOCK) void pqt_k_merge(
    const uint32_t* __restrict__ inIdx, const float* __restrict__ inDist, const uint32_t* __restrict__ inPos,
    uint32_t nsh, uint32_t qn, uint32_t k, uint64_t shardStride /* words between the [qn][k] blocks of consecutive shards */,
    uint32_t mP2, uint32_t* __restrict__ outIdx, float* __restrict__ outDist) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  uint32_t* sI = (uint32_t*)smem_raw;  // mP2 slot indices (0xffffffff = padding)
  uint32_t* sK = sI + mP2;             // mP2 distance keys
  uint32_t* sP = sK + mP2;             // mP2 positions
  const uint32_t q = blockIdx.x, tid = threadIdx.x;
  const uint32_t m = nsh * k;
  for (uint32_t i = tid; i < mP2; i += PQT_BLOCK) {
    uint32_t key = 0xffffffffu, pos = 0xffffffffu;
    if (i < m) {
      const size_t o = (size_t)(i / k) * shardStride + (size_t)q * k + (i % k);
      if (inIdx[o] != 0xffffffffu) { key = pqt_f2key(inDist[o]); pos = inPos[o]; }
    }
    sI[i] = i; sK[i] = key; sP[i] = pos;
  }
  __syncthreads();
  for (uint32_t kk = 2; kk <= mP2; kk <<= 1)
    for (uint32_t j = kk >> 1; j > 0; j >>= 1) {
      for (uint32_t i = tid; i < (mP2 >> 1); i += PQT_BLOCK) {
        const uint32_t lo = ((i & ~(j - 1)) << 1) | (i & (j - 1)), hi = lo | j;
        const bool up = ((lo & kk) == 0);
        const uint32_t x = sI[lo], y = sI[hi];
        const uint64_t kx = ((uint64_t)sK[x] << 32) | sP[x], ky = ((uint64_t)sK[y] << 32) | sP[y];
        const bool gt = kx > ky || (kx == ky && x > y);
        if (gt == up) { sI[lo] = y; sI[hi] = x; }
      }
      __syncthreads();
    }
  for (uint32_t i = tid; i < k; i += PQT_BLOCK) {
    const uint32_t s = sI[i];
    uint32_t id = 0xffffffffu; float d = __uint_as_float(0x7f800000u);
    if (s < m) {
      const size_t o = (size_t)(s / k) * shardStride + (size_t)q * k + (s % k);
      id = inIdx[o]; d = inDist[o];
    }
    outIdx[(size_t)q * k + i] = id;
    outDist[(size_t)q * k + i] = d;
  }
}
#endif  // PQT_MAIN_TU

// the same merge without the LDS-resident sort, for nsh*k beyond its 160 KiB (e.g. the class's whole-list query() through
// pqt_multi: k = 8192): every shard's list is already ascending in (distance, position), so an entry's place in the merged
// list is its own index plus, for every other list, the number of entries that precede it -- one binary search per other
// list (global memory).  Keys are unique (position), equal keys of different shards cannot occur; padding sits at the tail.
#ifdef PQT_MAIN_TU
__global__ __launch_bounds__(PQT_BLOCK) void pqt_k_merge_ranked(
    const uint32_t* __restrict__ inIdx, const float* __restrict__ inDist, const uint32_t* __restrict__ inPos,
    uint32_t nsh, uint32_t qn, uint32_t k, uint64_t shardStride, uint32_t* __restrict__ outIdx, float* __restrict__ outDist) {
  __shared__ uint32_t sN[65];
  const uint32_t q = blockIdx.x, tid = threadIdx.x;
  for (uint32_t s2 = tid; s2 < nsh; s2 += PQT_BLOCK) {  // real entries of list s2: first padding slot
    const uint32_t* li = inIdx + (size_t)s2 * shardStride + (size_t)q * k;
    uint32_t lo = 0, hi = k;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (li[mid] != 0xffffffffu) lo = mid + 1; else hi = mid; }
    sN[s2] = lo;
  }
  __syncthreads();
  uint32_t total = 0;
  for (uint32_t s2 = 0; s2 < nsh; ++s2) total += sN[s2];
  for (uint32_t e = tid; e < nsh * k; e += PQT_BLOCK) {
    const uint32_t sh = e / k, i = e % k;
    if (i >= sN[sh]) continue;
    const size_t o = (size_t)sh * shardStride + (size_t)q * k + i;
    const uint64_t key = ((uint64_t)pqt_f2key(inDist[o]) << 32) | inPos[o];
    uint32_t rank = i;
    for (uint32_t t = 0; t < nsh; ++t) {
      if (t == sh) continue;
      const size_t ot = (size_t)t * shardStride + (size_t)q * k;
      uint32_t lo = 0, hi = sN[t];  // entries of list t that precede `key` (ties, impossible for distinct positions, go to the lower shard)
      while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        const uint64_t km = ((uint64_t)pqt_f2key(inDist[ot + mid]) << 32) | inPos[ot + mid];
        if (km < key || (km == key && t < sh)) lo = mid + 1; else hi = mid;
      }
      rank += lo;
    }
    if (rank < k) { outIdx[(size_t)q * k + rank] = inIdx[o]; outDist[(size_t)q * k + rank] = inDist[o]; }
  }
  for (uint32_t i = total + tid; i < k; i += PQT_BLOCK) { outIdx[(size_t)q * k + i] = 0xffffffffu; outDist[(size_t)q * k + i] = __uint_as_float(0x7f800000u); }
}
#endif  // PQT_MAIN_TU

// ---------------------------------------------------------------------------------------------------
// offline ("next" row): insert = id() + prepareReranking for a batch of database vectors.
//   one workgroup per vector.  bin id: nearest first-level cell per part, nearest second-level centroid
//   inside it (treequantizer.hpp:673-684); line code per line part: the pair A < B minimising the
//   projection error, first minimum in (A,B) scan order (:356-391).
// LDS: D + LP*C1 + P*C1 + P + C2*P words + reduction scratch.
// ---------------------------------------------------------------------------------------------------
#ifdef PQT_MAIN_TU
__global__ __launch_bounds__(PQT_BLOCK) void pqt_k_assign_encode(
    const float* __restrict__ X, const float* __restrict__ cb1, const float* __restrict__ cb2,
    const float* __restrict__ coarse, PqtDevParams prm, uint32_t* __restrict__ outBin, uint32_t* __restrict__ outCodes,
    uint32_t dbgSkipPairs /* measurement only (debug_bits 8192): no pair search, codes are written as 0 */) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const uint32_t D = prm.D, P = prm.P, C1 = prm.C1, C2 = prm.C2, LP = prm.LP, S = prm.S, SS = prm.SS, R = prm.R;
  float* sX = smem;
  float* sVirt = sX + D;
  float* sL1 = sVirt + LP * C1;
  uint32_t* sBest1 = (uint32_t*)(sL1 + P * C1);   // P
  float* sD2 = (float*)(sBest1 + P);              // P*C2
  uint64_t* sRed = (uint64_t*)(sD2 + P * C2 + ((P * C2 + P) & 1));  // PQT_BLOCK  (8-byte aligned: D, LP*C1, P*C1 even in practice)
  const size_t v = blockIdx.x;
  const uint32_t tid = threadIdx.x;
  for (uint32_t i = tid; i < D; i += PQT_BLOCK) sX[i] = X[v * D + i];
  __syncthreads();
  for (uint32_t t = tid; t < C1 * LP; t += PQT_BLOCK) {
    const uint32_t c = t / LP, lp = t % LP;
    const float* cen = cb1 + (size_t)c * D + lp * SS;
    const float* xx = sX + lp * SS;
    float s = 0.f;
    for (uint32_t d = 0; d < SS; ++d) { const float df = xx[d] - cen[d]; s = s + df * df; }
    sVirt[lp * C1 + c] = s;
  }
  __syncthreads();
  for (uint32_t t = tid; t < P * C1; t += PQT_BLOCK) {
    const uint32_t p = t / C1, c = t % C1;
    float d = 0.f;
    for (uint32_t pp = 0; pp < R; ++pp) d = d + sVirt[(p * R + pp) * C1 + c];
    sL1[t] = d;
  }
  __syncthreads();
  if (tid < P) {  // nearest cell per part, ties -> lowest index (stable order of the reference's sort)
    uint32_t best = 0; float bd = sL1[tid * C1];
    for (uint32_t c = 1; c < C1; ++c) { const float d = sL1[tid * C1 + c]; if (d < bd) { bd = d; best = c; } }
    sBest1[tid] = best;
  }
  __syncthreads();
  for (uint32_t t = tid; t < P * C2; t += PQT_BLOCK) {
    const uint32_t p = t / C2, h2 = t % C2;
    const float* cen = cb2 + (((size_t)p * C1 + sBest1[p]) * C2 + h2) * S;
    const float* xx = sX + p * S;
    float s = 0.f;
    for (uint32_t d = 0; d < S; ++d) { const float df = xx[d] - cen[d]; s = s + df * df; }
    sD2[t] = s;
  }
  __syncthreads();
  if (tid == 0) {
    uint32_t bin = 0;
    for (uint32_t p = 0; p < P; ++p) {
      uint32_t best = 0; float bd = sD2[p * C2];
      for (uint32_t h = 1; h < C2; ++h) { const float d = sD2[p * C2 + h]; if (d < bd) { bd = d; best = h; } }
      bin += (sBest1[p] * C2 + best) * prm.powers[p];
    }
    outBin[v] = bin;
  }
  // line codes: for each line part, search all pairs A < B.  pair index e enumerates (A,B) in the
  // reference's scan order; reduce by (error key, e) so the first minimum wins like its strict '<'.
  const uint32_t npairs = C1 * (C1 - 1) / 2;
  // a thread evaluates the same pairs e = tid + 256 i for every line part: decode them ONCE (the row-major walk costs up to
  // C1 steps per pair and used to run LP times per pair -- two thirds of this kernel's instructions)
  constexpr int kPairCache = 8;  // pairs per thread kept in registers: covers C1 <= 64
  uint32_t pairAB[kPairCache];
#pragma unroll
  for (int i = 0; i < kPairCache; ++i) {
    const uint32_t e = tid + PQT_BLOCK * i;
    uint32_t A = 0, rem = e;
    if (e < npairs) { while (rem >= C1 - 1 - A) { rem -= C1 - 1 - A; ++A; } }
    pairAB[i] = A | ((A + 1 + rem) << 16);
  }
  if (dbgSkipPairs) { for (uint32_t lp = tid; lp < LP; lp += PQT_BLOCK) outCodes[v * LP + lp] = 0; return; }
  for (uint32_t lp = 0; lp < LP; ++lp) {
    uint64_t best = ~0ull;
    auto eval = [&](const uint32_t e, const uint32_t A, const uint32_t B) {
      const float sb = sVirt[lp * C1 + A], sa = sVirt[lp * C1 + B];
      const float sc = coarse[((size_t)lp * C1 + A) * C1 + B];
      const float lam = pqt_calc_ratio(sa, sb, sc);
      const float err = pqt_extract_distance(sa, sb, sc, lam);
      // strict '<' against HUGE_VAL start: NaN errors never win, +inf never wins
      if (err == err && err < __uint_as_float(0x7f800000u)) {
        const uint64_t key = ((uint64_t)pqt_f2key(err) << 32) | e;
        if (key < best) best = key;
      }
    };
#pragma unroll
    for (int i = 0; i < kPairCache; ++i) {
      const uint32_t e = tid + PQT_BLOCK * i;
      if (e < npairs) eval(e, pairAB[i] & 0xffffu, pairAB[i] >> 16);
    }
    for (uint32_t e = tid + PQT_BLOCK * kPairCache; e < npairs; e += PQT_BLOCK) {
      // e -> (A,B), A<B, row-major over A (rows have C1-1-A entries)
      uint32_t A = 0, rem = e;
      while (rem >= C1 - 1 - A) { rem -= C1 - 1 - A; ++A; }
      eval(e, A, A + 1 + rem);
    }
    // minimum over the block: butterfly inside each wavefront, then the 4 wave minima through LDS
    { uint64_t o = pqt_lane_xor_u64<1>(best); best = o < best ? o : best; }
    { uint64_t o = pqt_lane_xor_u64<2>(best); best = o < best ? o : best; }
    { uint64_t o = pqt_lane_xor_u64<4>(best); best = o < best ? o : best; }
    { uint64_t o = pqt_lane_xor_u64<8>(best); best = o < best ? o : best; }
    { uint64_t o = pqt_lane_xor_u64<16>(best); best = o < best ? o : best; }
    { uint64_t o = pqt_lane_xor_u64<32>(best); best = o < best ? o : best; }
    if ((tid & 63u) == 0) sRed[tid >> 6] = best;
    __syncthreads();
    if (tid == 0) {
      uint64_t w = sRed[0];
      for (uint32_t wv = 1; wv < PQT_BLOCK / 64; ++wv) w = sRed[wv] < w ? sRed[wv] : w;
      uint32_t code;
      if (w == ~0ull) {
        code = pqt_lambda_encode(0.f) << 16;  // best_id_A = best_id_B = 0, best_lambda = 0 (initial values, :362-365)
      } else {
        const uint32_t e = (uint32_t)w;
        uint32_t A = 0, rem = e;
        while (rem >= C1 - 1 - A) { rem -= C1 - 1 - A; ++A; }
        const uint32_t B = A + 1 + rem;
        const float sb = sVirt[lp * C1 + A], sa = sVirt[lp * C1 + B];
        const float sc = coarse[((size_t)lp * C1 + A) * C1 + B];
        const float lam = pqt_calc_ratio(sa, sb, sc);
        code = (A & 0xffu) | ((B & 0xffu) << 8) | (pqt_lambda_encode(lam) << 16);
      }
      outCodes[v * LP + lp] = code;
    }
    __syncthreads();
  }
}
#endif  // PQT_MAIN_TU

// line-quantisation scalars evaluated on the device (known-answer tests, run.cu:33-113)
#ifdef PQT_MAIN_TU
__global__ void pqt_k_triangle(const float* a, const float* b, const float* c, const float* l, uint32_t n,
                               float* outDist, float* outRatio, uint16_t* outU16, float* outRound) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  outDist[i] = pqt_extract_distance(a[i], b[i], c[i], l[i]);
  outRatio[i] = pqt_calc_ratio(a[i], b[i], c[i]);
  const uint32_t u = pqt_lambda_encode(l[i]);
  outU16[i] = (uint16_t)u;
  outRound[i] = pqt_lambda_decode(u);
}
#endif  // PQT_MAIN_TU

// ===================================================================================================
// Fused stage a7 + a8 (k <= 128): ADC line rerank + exact top-k, one WAVEFRONT per query.
//
//   * workgroup = NW independent wavefronts that share one LDS copy of coarse[LP][C1][C1] (when it fits:
//     <= 64 KB, i.e. cfg1/cfg2); each wavefront walks queries q = wg*NW + w, += grid*NW.
//   * lane = one candidate: reads its id, streams the 4*LP-byte code row as 16-byte vectors, accumulates the LP
//     terms in p order (bit-exact, see pqt_k_rerank); the three table look-ups per term are LDS reads
//     (L1virt of the query + coarse), not TA gathers.
//   * selection: keys (f32 key << 32 | visiting position) are unique, so "the k smallest keys" is exact.
//     A candidate enters the wave's pending buffer only if its key beats tau, the current k-th best;
//     when the buffer fills (or at the end) [best 128 | pending] is sorted by the in-register bitonic network
//     (pqt_wave_sort_u64<8>, 512 keys) and the first 128 become the new best.  Distances never touch HBM.
// LDS: coarse (optional) + NW * (LP*C1*4 + 384*8) bytes.
// ===================================================================================================

#define PQT_RS_BEST 128
#ifndef PQT_RS_PEND
#define PQT_RS_PEND 384
#endif
#ifndef PQT_RS_QUAD
#define PQT_RS_QUAD 0     // 1: 64-byte rows are fetched by the 4 lanes of a quad (one access per row) and transposed in registers.
                          // Measured r02 (sift1m bench, parity suite green): rerank+select 0.156 -> 0.163 ms, the per-query wait for
                          // rows unchanged (12.6 k clocks) -- the wait is the id -> row dependent round trips, not the number of L1
                          // accesses per row -- so the variant stays off.
#endif
#ifndef PQT_RS_FINAL_SORT8
#define PQT_RS_FINAL_SORT8 0  // 1: the last flush of a query sorts its <= 512 keys in one pass of the in-register network instead of
                              // radix select + compaction + the 128-key network.  Measured r02 (SIFT1M shape): flush 16.5 k -> 17.0 k
                              // clocks per query, rerank+select 0.155 -> 0.162 ms: the 1.5 k instructions of the big network cost
                              // what the select's LDS round trips cost; off.
#endif
#define PQT_RUNCAP 128   // bin runs per query handed from the traversal to the rerank (more: the plain candidate list is written)
#ifndef PQT_RS_STATIC_PCT
#define PQT_RS_STATIC_PCT 65  // rerank schedule 2: share of an XCD pool (its longest queries) handed out without atomics
#endif
#define PQT_RS_LIST 256   // queries of a workgroup's list that are ranked by candidate count (the rest follow in index order)

// arguments of the fused rerank + select (kernel-argument segment)
struct PqtRsArgs {
  const uint32_t* codes;  // bin-ordered line store
  const uint32_t* ids; const float* qL1virt; const float* coarse; const uint32_t* cand; const uint32_t* candPos;
  const uint32_t* nLocal; uint64_t stride; uint32_t k, qn; PqtDevParams prm;
  uint32_t* outIdx; float* outDist; uint32_t* outPos;
  unsigned long long* counters; uint32_t dbg; unsigned long long* tstamp;
  uint32_t dynamic;          // 0: static round-robin; 1: workgroup-local dynamic schedule over a fixed share of the queries,
                             // longest first; 2: chunks drawn from global per-XCD pools, other pools' leftovers when the own is empty
  unsigned long long* zero8; // statistics block of the next call, zeroed here (saves a memset launch)
  // opt-in "adc_bias" mode (MODE 1 of pqt_rs_query): group-major copy of the line store [LP/4][nIds] 16-byte pieces, rows in
  // it, and the per-row query-independent part of the ADC sum
  const uint4* codesGrp4; uint64_t nIds; const float* bias;
  // MODE 2 (exact results through the bias-mode filter): error-bound coefficients and the fallback list
  float epsKappa;   // 2.02 * 2^-24 * (LP^2 + 8 LP + 2)
  float cmax20;     // 20 * max coarse entry
  uint32_t* fbList; uint32_t* fbCount;  // queries whose near-tie band did not fit the wave's list: redone by the plain exact kernel
  const uint32_t* qlist; const uint32_t* qcount;  // pqt_k_rerank_select_list: the queries to process
  // bin runs instead of a candidate list (MODE 0 with the coarse table in LDS): runs[q][PQT_RUNCAP] = (first visiting
  // position | first store position << 32) of the included bins in visiting order, nRuns[q] = their number or 0xffffffff
  // when the traversal wrote the plain list after all; runGpos (sharded) = global visiting position of a run's first member
  const unsigned long long* runs; const uint32_t* runGpos; const uint32_t* nRuns;
  uint32_t runCap;  // run slots of a wavefront's LDS area (64 or PQT_RUNCAP); the traversal hands over at most this many
  // dynamic == 2: eight global draw counters (pool x = the queries q with q % 8 == x, in index order) of this launch, and the
  // block of the next launch, zeroed here
  uint32_t* pool; uint32_t* poolNext;
  // ... the traversal's registration lists (see PqtTravArgs): pool x is drawn class by class, largest first
  const uint32_t* schedCnt; const unsigned long long* schedList; uint32_t schedCap;
  uint32_t padDone;  // k > 128 kernels: the padding behind the results is written elsewhere (pqt_k_pad_rows)
  // PRE (shared-row pass, pqt_shared_rows.h): the filter distances d1 of the candidates were written by pqt_k_sr_adc -- once per
  // distinct (bin, query) pair, the rows of a bin read once for all the queries of the batch that include it -- into
  // preDist[q * stride + visiting position]; preOk[q] != 0 marks the queries it covered (the others evaluate their rows here)
  const float* preDist; const uint32_t* preOk; const float* preQmax /* [q] largest entry of the query's L1virt table (pqt_k_sr_visits) */;
  // PRE split in two launches (2: scan, 3: band): the scan leaves the <= 256 smallest filter keys of query q, ascending, in preKeys[q][256] and
  // their number in preCnt[q] (0xffffffff: handed back); the band launch takes them from there
  unsigned long long* preKeys; uint32_t* preCnt;
  const uint32_t* preFlags;  // PqtSrArgs::total of the pass: [2] != 0 = its items / lists did not fit, no query is covered
};

// a7 + a8 of query q (n local candidates) by the calling wavefront.  sKeys: its PQT_RS_BEST + PQT_RS_PEND key slots,
// sVirt: its LP*C1 floats (loaded here), cz: the coarse table (LDS copy at offset 0 of the dynamic LDS, or global).
// qN / nN: the wavefront's next query; a count still unknown (0xffffffff) is fetched under the final select + sort.
// MODE 0: the reference's association, term by term (bit-exact, the default).
// MODE 1 ("adc_bias", opt-in, SURVEY App. C "E-alt"): extractDistance(a, b, c, l) = b + l*(a - b) - l*(1 - l)*c, and the
//   last part does not depend on the query: bias[row] = sum_p (l*l*c - l*c) is computed once per index (pqt_k_adc_bias),
//   a term is b + l*(a - b) -- two LDS look-ups into the query's L1virt, no coarse[LP][C1][C1] table at all -- and the
//   distance is sum_p term_p + bias[row].  Same real number, different rounding: candidate SETS are untouched (they are
//   fixed before this stage), distances differ from MODE 0 in the last bits, so the top-k can differ among near-equal
//   distances.  The code words come from the group-major copy of the store (consecutive candidates = contiguous bytes).
// MODE 2 (default for first-level codebooks whose coarse table does not fit the LDS: BASELINE configs[2]/[3]): the
//   REFERENCE result, bit for bit, at MODE 1's cost.  Both formulas evaluate the same real number D; with M = the largest
//   entry of the query's L1virt table and Cmax = the largest coarse entry every intermediate of either formula is bounded
//   by G = 5 M + 20 Cmax, a term costs <= 8 roundings and the sequential sum of LP terms <= LP roundings of partial sums
//   <= LP*G, so |d0 - D| and |d1 - D| are each below u*G*(LP^2 + 8 LP + 2) (u = 2^-24) and |d0 - d1| <= eps =
//   2.02*u*G*(LP^2 + 8 LP + 2).  Hence every member of the exact top-k has d1 <= d1_(k) + 2 eps (d1_(k) = k-th smallest MODE 1
//   distance): the wave keeps the 256 smallest MODE 1 keys, re-evaluates the reference association only for the entries
//   inside that band (coarse look-ups from L2, ~k + a few candidates per query instead of thousands), and sorts those by
//   the exact key.  If the band reaches the end of a full list (a cluster of > 256 - k near-ties) the query is appended
//   to fbList and redone by the plain exact kernel (pqt_k_rerank_select_list) -- never a wrong answer, rarely a slow one.
// XC (MODE 0, compile-time C1, coarse table in LDS): the rows come from the X-code copy of the line store (pqt_k_xcode: the two
//   centroid bytes replaced by the byte offset A*4*C1 + B*4 of coarse[p][A][B] inside the part's table, lambda unchanged in the upper
//   half): the three LDS addresses of a term cost 4 VALU instructions instead of 6, lambda is scaled by one packed FMA (the product
//   u16 * 2^-13 is exact, so the FMA rounds like the separate multiply and add), and two CANDIDATES are evaluated per packed
//   instruction (the running sums of both in one packed add) -- every candidate's own sequence of roundings is unchanged: same bits.
// NSLOT: 8-byte key slots of the wavefront (best list + pending buffer): 512 by default, 384 in the 16-wavefront configuration
template <int LPV, int UREQ, bool COARSE_LDS, bool SHARDED, int C1M, int MODE = 0, bool RUNS = false, bool XC = false, int NSLOT = PQT_RS_BEST + PQT_RS_PEND,
          int PRE = 0 /* != 0: MODE 2 / 0 + RUNS, selection only (pqt_k_sr_select): the filter distances come from A.preDist (see PqtRsArgs), no row is
                         fetched in the batch loop; 1: the whole selection (sVirt may be the query's table in GLOBAL memory: only the band re-evaluation
                         reads it); 2: the scan alone -- the best list goes to A.preKeys; 3: the band re-evaluation, sort and results alone, from A.preKeys */,
          int COOP = 0 /* 1 (MODE 2 + RUNS, round 6): the calling wavefront is one of TWO that scan the candidates of query q around ONE LDS copy of its
                          table (pqt_k_pair_scan): it takes the batches coopHalf, coopHalf + 2, ... and leaves its <= 256 best filter keys in
                          A.preKeys[(2 q + coopHalf) * 256 ..] / A.preCnt[2 q + coopHalf]; the lists of a pair are merged and the band re-evaluated by
                          later launches (pqt_k_sr_merge, PRE = 3) */>
__device__ __forceinline__ void pqt_rs_query(const PqtRsArgs& A, const uint32_t q, const uint32_t n, uint64_t* const sKeys, float* const sVirt,
                                             const float* const cz, const uint32_t qN, uint32_t& nN, const uint32_t slot, uint32_t& tiesAcc,
                                             unsigned long long* const sRuns = nullptr /* PQT_RUNCAP u64 + PQT_RUNCAP u32 of this wave, or null */,
                                             const uint32_t coopHalf = 0 /* COOP: 0 | 1 */) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  static_assert(!COOP || (MODE == 2 && RUNS && !PRE && !XC && !COARSE_LDS), "pair scan: the filtered rerank with bin runs");
  const uint32_t* __restrict__ codes = A.codes; const uint32_t* __restrict__ ids = A.ids; const float* __restrict__ qL1virt = A.qL1virt;
  const uint32_t* __restrict__ cand = A.cand; const uint32_t* __restrict__ candPos = A.candPos; const uint32_t* __restrict__ nLocal = A.nLocal;
  const uint64_t stride = A.stride; const uint32_t k = A.k; uint32_t* __restrict__ outIdx = A.outIdx; float* __restrict__ outDist = A.outDist;
  uint32_t* __restrict__ outPos = A.outPos; const uint32_t dbg = A.dbg;
  unsigned long long* __restrict__ tstamp = A.tstamp;
  (void)candPos; (void)outPos; (void)cz;
  // U candidates per lane are in flight together (16 code vectors = 64 VGPRs): the id -> row -> table chain of
  // one candidate is ~3 dependent memory round trips, so memory-level parallelism has to come from here.
  constexpr int U = UREQ;
  constexpr uint32_t LP = LPV * 4;
  // candidates taken by this wavefront: base = jStart, jStart + jStep, ... (all of them unless COOP)
  constexpr uint32_t jStep = COOP ? 2u * 64u * UREQ : 64u * UREQ;
  const uint32_t jStart = COOP ? coopHalf * 64u * UREQ : 0u;
  const uint32_t C1 = C1M >= 2 ? (1u << C1M) : A.prm.C1;
  constexpr bool C1P2 = C1M != 0;
  const uint32_t c1sh = C1M >= 2 ? (uint32_t)C1M : (C1P2 ? (uint32_t)__builtin_ctz(C1) : 0u);  // power-of-two C1: shifts instead of quarter-rate multiplies
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t vOff = (uint32_t)(reinterpret_cast<unsigned char*>(sVirt) - smem_raw);  // byte offset of this wave's L1virt copy
  (void)c1sh; (void)vOff;
  // debug timestamps (slots 9..13 of the per-query record): start, cycles waiting for rows, ADC + filter, flushes, end
  unsigned long long tsLoad = 0, tsAdc = 0, tsFlush = 0, ts0 = 0;
  unsigned long long tsStart = 0, wallStart = 0;
  if (tstamp) { tsStart = __builtin_readcyclecounter(); wallStart = wall_clock64(); }
  if (tstamp && lane == 0) tstamp[(size_t)q * PQT_TS_WORDS + 9] = tsStart;
  const uint32_t* cid = cand + (size_t)q * stride;
  const uint32_t* cpos = SHARDED ? candPos + (size_t)q * stride : nullptr;
  if (tstamp) ts0 = __builtin_readcyclecounter();
  // Bin runs (MODE 0 with the LDS table): the traversal hands over the included bins as (first visiting position, first
  // store position) pairs instead of one store position per candidate; candidate j's row is found by a 7-step search
  // over <= 128 LDS entries -- no candidate list in HBM, no id round trip in front of the row round trip.
  constexpr bool kRuns = RUNS;  // compiled as separate variants (pqt_index_set_option "bin_runs")
  uint32_t mRuns = 0xffffffffu;
  // the run count and this lane's two run slots are requested together (one round trip; slots beyond the count hold
  // stale words that are masked below)
  unsigned long long rr0 = 0, rr1 = 0;
  if constexpr (kRuns) {
    if (sRuns && A.nRuns) {
      mRuns = A.nRuns[q];
      rr0 = A.runs[(size_t)q * PQT_RUNCAP + lane];
      if (A.runCap > 64) rr1 = A.runs[(size_t)q * PQT_RUNCAP + 64 + lane];
    }
  }
  if constexpr (PRE) {
    static_assert((MODE == 2 || MODE == 0) && RUNS && !COARSE_LDS && !XC, "precomputed distances: filter distances (MODE 2) or exact ones (MODE 0), bin-runs variant");
    if constexpr (MODE == 2) {
      // a query the shared-row pass did not cover (its runs did not fit the hand-over, or its bins the pass's table) goes where the
      // queries with an overflowing near-tie band go: exact distances by a whole workgroup, then the MODE 0 selection (pqt_k_sr_exact_list)
      if (n && (A.preOk[q] == 0u || (A.preFlags && A.preFlags[2]))) {
        if (PRE != 3 && lane == 0) { A.fbList[atomicAdd(A.fbCount, 1u)] = q; if (PRE == 2) A.preCnt[q] = 0xffffffffu; }
        return;
      }
    }
  }
  const bool useRuns = kRuns && mRuns != 0xffffffffu;
  constexpr bool usePre = PRE != 0;
  const float* const preRow = PRE ? A.preDist + (size_t)q * stride : nullptr;
  (void)preRow;
  // PRE: a batch costs ~50 instructions here, so its distances must be on their way long before they are used: a queue of PRE_DEPTH
  // batches (one register per 64 candidates each) instead of one request and one exposed round trip per batch
  constexpr int PRE_DEPTH = PRE ? 8 : 1;
  float preQ[PRE_DEPTH][UREQ];
  (void)preQ;
  if constexpr (PRE != 0 && PRE != 3) {
    if (usePre) {
#pragma unroll
      for (int dq = 0; dq < PRE_DEPTH; ++dq) {
#pragma unroll
        for (int u = 0; u < UREQ; ++u) { const uint32_t j = (uint32_t)(dq * UREQ + u) * 64u + lane; preQ[dq][u] = n ? preRow[j < n ? j : n - 1] : 0.f; }
      }
    }
  }
  uint32_t* const sRunG = reinterpret_cast<uint32_t*>(sRuns + A.runCap);
  // runs `lane` and `64 + lane` also live in registers: a batch of 64 consecutive candidates spans a handful of runs, which
  // are broadcast one after the other (v_readlane with a uniform index) -- no search, no LDS latency on the row path
  uint32_t rs0 = 0xffffffffu, rl0 = 0, rs1 = 0xffffffffu, rl1 = 0;
  if (useRuns) {
    if (lane < mRuns) { sRuns[lane] = rr0; rs0 = (uint32_t)rr0; rl0 = (uint32_t)(rr0 >> 32); }
    if (64 + lane < mRuns) { sRuns[64 + lane] = rr1; rs1 = (uint32_t)rr1; rl1 = (uint32_t)(rr1 >> 32); }
    if (SHARDED) for (uint32_t i = lane; i < mRuns; i += 64) sRunG[i] = A.runGpos[(size_t)q * PQT_RUNCAP + i];
  }
  // store position of candidate jb + lane for the 64 consecutive candidates starting at jb (uniform)
  auto expand64 = [&](const uint32_t jb) -> uint32_t {
    const uint32_t jm = jb + lane;
    uint32_t r = (uint32_t)__popcll(__ballot(rs0 <= jb)) + (uint32_t)__popcll(__ballot(rs1 <= jb));  // runs starting at or before jb
    r = r ? r - 1 : 0;
    uint32_t pos = 0;
    for (;;) {
      const uint32_t s0 = r < 64 ? (uint32_t)__builtin_amdgcn_readlane((int)rs0, (int)r) : (uint32_t)__builtin_amdgcn_readlane((int)rs1, (int)(r - 64));
      const uint32_t l0 = r < 64 ? (uint32_t)__builtin_amdgcn_readlane((int)rl0, (int)r) : (uint32_t)__builtin_amdgcn_readlane((int)rl1, (int)(r - 64));
      const uint32_t rn = r + 1;
      const uint32_t e0 = rn < mRuns ? (rn < 64 ? (uint32_t)__builtin_amdgcn_readlane((int)rs0, (int)rn) : (uint32_t)__builtin_amdgcn_readlane((int)rs1, (int)(rn - 64))) : 0xffffffffu;
      if (jm >= s0 && jm < e0) pos = l0 + (jm - s0);
      if (e0 >= jb + 64 || rn >= mRuns) break;
      r = rn;
    }
    return pos;
  };
  // run index of candidate j: last run whose first visiting position is <= j (branch-free, 7 LDS reads)
  auto runOf = [&](const uint32_t j) -> uint32_t {
    uint32_t lo = 0;
#pragma unroll
    for (uint32_t s2 = PQT_RUNCAP / 2; s2 >= 1; s2 >>= 1) {
      const uint32_t mid = lo + s2;
      if (mid < mRuns && (uint32_t)sRuns[mid] <= j) lo = mid;
    }
    return lo;
  };
  // the store positions of a batch are requested one batch ahead (the first batch's here, under the L1virt copy): the
  // id -> row chain of a batch is then ONE round trip on the critical path instead of two
  uint32_t idNext[UREQ];
#pragma unroll
  for (int u = 0; u < UREQ; ++u) {
    const uint32_t j = jStart + u * 64 + lane;
    idNext[u] = (!PRE && n && !useRuns) ? cid[j < n ? j : n - 1] : 0u;  // (PRE: positions are needed for the results only)
  }
  float qmax = 0.f;  // MODE 2: largest entry of the query's L1virt table
  if constexpr (PRE) {
    if constexpr (MODE == 2) qmax = A.preQmax[q];  // (no LDS copy of the table: sVirt points at it in global memory)
  } else
  if constexpr (C1M >= 2) {
    // compile-time shape: the whole table is requested before the first piece is stored (the run-time loop below compiled to one
    // load + s_waitcnt per 1 KB: 8 serialised round trips per query at the configs[2] shape, ~14 k clocks of set-up)
    constexpr uint32_t NV = LPV * 4 * (1u << C1M) / 4;
    constexpr uint32_t IT = (NV + 63) / 64;
    const float4* src4 = reinterpret_cast<const float4*>(qL1virt + (size_t)q * LP * C1);
    float4* dst4 = reinterpret_cast<float4*>(sVirt);
    float4 tmp[IT];
#pragma unroll
    for (uint32_t i = 0; i < IT; ++i) { const uint32_t t = lane + 64 * i; tmp[i] = src4[t < NV ? t : 0]; }
#pragma unroll
    for (uint32_t i = 0; i < IT; ++i) {
      const uint32_t t = lane + 64 * i;
      if (t < NV) {
        dst4[t] = tmp[i];
        if constexpr (MODE == 2) { const float m01 = tmp[i].x > tmp[i].y ? tmp[i].x : tmp[i].y, m23 = tmp[i].z > tmp[i].w ? tmp[i].z : tmp[i].w; const float m = m01 > m23 ? m01 : m23; qmax = m > qmax ? m : qmax; }
      }
    }
  } else if ((C1 & 3u) == 0) {  // LP*C1 floats as 16-byte pieces (both ends are 16-byte aligned)
    const float4* src4 = reinterpret_cast<const float4*>(qL1virt + (size_t)q * LP * C1);
    float4* dst4 = reinterpret_cast<float4*>(sVirt);
    for (uint32_t t = lane; t < LP * C1 / 4; t += 64) dst4[t] = src4[t];
  } else
  for (uint32_t t = lane; t < LP * C1; t += 64) sVirt[t] = qL1virt[(size_t)q * LP * C1 + t];
  __builtin_amdgcn_wave_barrier();
  uint64_t tau = ~0ull;
  uint32_t npend = 0;  // pending keys sit at sKeys[off0 ..], off0 = size of the best list kept by the last flush
  uint32_t off0 = 0;
  // best-list size: k normally; MODE 2 keeps the 256 smallest MODE 1 keys (the k-th plus a band of near-ties)
  constexpr uint32_t BESTN = MODE == 2 ? 256u : (uint32_t)PQT_RS_BEST;
  constexpr uint32_t SLOTS = NSLOT;
  static_assert(BESTN + 64u * UREQ <= SLOTS, "a batch of appended keys must fit behind the best list");
  const uint32_t kSel = MODE == 2 ? BESTN : k;
  if constexpr (MODE == 2 && !PRE) {
    if constexpr (C1M < 2) { for (uint32_t t = lane; t < LP * C1; t += 64) { const float v = sVirt[t]; qmax = v > qmax ? v : qmax; } }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { const float o = __shfl_xor(qmax, d, 64); qmax = o > qmax ? o : qmax; }
  }

  unsigned long long tsSetup = 0, tsVerify = 0, tsOut = 0;
  if (tstamp) tsSetup = __builtin_readcyclecounter() - tsStart;
  auto flush = [&](const bool final) {
    // [best off0 (unsorted, after the first flush) | pending npend]: keep the k smallest.  More than 128 keys are cut
    // down by an exact radix select (pqt_wave_kth_u64) instead of a full sort; only the final <= 128 survivors
    // go through the (small) in-register sorting network.  The keys live in registers during the select, so its
    // counters reuse the pending area of sKeys (1056 bytes behind the best list).
    uint32_t have = off0 + npend;
    if (PQT_RS_FINAL_SORT8 && final && have > BESTN) {
      // last flush: at most 512 keys are held; one pass of the 512-key in-register network (no LDS atomics, no scans)
      // replaces radix select + compaction + the small network
      constexpr int RK = NSLOT / 64;
      uint64_t key[RK];
#pragma unroll
      for (int r = 0; r < RK; ++r) {
        const uint32_t e = lane * RK + r;
        key[r] = (e < have) ? sKeys[e] : ~0ull;
      }
      __builtin_amdgcn_wave_barrier();
      if (!(dbg & 1)) pqt_wave_sort_u64<RK>(key);
      // the first BESTN sorted keys live in lanes 0 .. BESTN/RK - 1
      if (lane < BESTN / RK) {
#pragma unroll
        for (int r = 0; r < RK; ++r) sKeys[lane * RK + r] = key[r];
      }
      __builtin_amdgcn_wave_barrier();
      npend = 0;
      off0 = have < kSel ? have : kSel;
      return;
    }
    if (have > BESTN) {
      constexpr int RK = NSLOT / 64;
      uint64_t key[RK];
#pragma unroll
      for (int r = 0; r < RK; ++r) {
        const uint32_t e = r * 64 + lane;
        key[r] = (e < have) ? sKeys[e] : ~0ull;
      }
      __builtin_amdgcn_wave_barrier();
      tau = pqt_wave_kth_u64<RK>(key, kSel, reinterpret_cast<uint32_t*>(sKeys + BESTN));
      uint32_t cnt = 0;
#pragma unroll
      for (int r = 0; r < RK; ++r) {
        uint32_t tot;
        const uint32_t rk = pqt_ballot_rank(key[r] <= tau, &tot);
        if (key[r] <= tau) sKeys[cnt + rk] = key[r];
        cnt += tot;
      }
      have = kSel;
      __builtin_amdgcn_wave_barrier();
    }
    if (final) {
      constexpr int FR = (int)(BESTN / 64);
      uint64_t key[FR];
#pragma unroll
      for (int r = 0; r < FR; ++r) {
        const uint32_t e = lane * FR + r;
        key[r] = (e < have) ? sKeys[e] : ~0ull;
      }
      if (!(dbg & 1)) pqt_wave_sort_u64<FR>(key);
#pragma unroll
      for (int r = 0; r < FR; ++r) sKeys[lane * FR + r] = key[r];
      __builtin_amdgcn_wave_barrier();
    }
    npend = 0;
    off0 = have;
  };
  // MODE 0, first phase (no threshold yet): every candidate's distance key goes to the 32-bit slot of its own visiting position --
  // the key area holds 2 * SLOTS of them, so most queries (SIFT1M shape: 90 % have <= 743 candidates) are selected ONCE, at the end,
  // instead of once per full pending buffer; no ballot ranking and no 64-bit key per candidate.  flush32 keeps the k smallest
  // (radix select on the distance keys, ties at the k-th value in visiting order = the order of the 64-bit keys) as 64-bit keys in
  // sKeys[0 .. k) and hands over to the threshold-filtered second phase above if candidates remain.
#ifdef PQT_NO_PHASE1
  constexpr bool kPhase1 = false;
#else
  constexpr bool kPhase1 = MODE == 0;
#endif
  constexpr uint32_t CAP32 = 2u * SLOTS;
  uint32_t* const sK32 = reinterpret_cast<uint32_t*>(sKeys);
  bool phase1 = kPhase1;
  auto flush32 = [&](const uint32_t have, const bool final) {
    constexpr int R32 = (int)(CAP32 / 64);
    uint32_t k32[R32];
#pragma unroll
    for (int r = 0; r < R32; ++r) k32[r] = ((uint32_t)r * 64u < have) ? sK32[(uint32_t)r * 64u + lane < have ? (uint32_t)r * 64u + lane : 0u] : 0u;
    __builtin_amdgcn_wave_barrier();
    uint32_t kept = have;
    if (have > kSel) {
      const uint32_t t32 = pqt_wave_kth_u32<R32>(k32, have, kSel, reinterpret_cast<uint32_t*>(sKeys + BESTN));
      // keys below the k-th value all stay; of those equal to it the first (k - #below) in visiting order
      uint32_t below = 0;
#pragma unroll
      for (int r = 0; r < R32; ++r)
        if ((uint32_t)r * 64u < have) below += (uint32_t)__popcll(__ballot((uint32_t)r * 64u + lane < have && k32[r] < t32));
      const uint32_t needEq = kSel - below;
      uint32_t cnt = 0, eqSeen = 0, lastEq = 0;
#pragma unroll
      for (int r = 0; r < R32; ++r) {
        if ((uint32_t)r * 64u < have) {
          const uint32_t e = (uint32_t)r * 64u + lane;
          const bool v = e < have;
          const bool isEq = v && k32[r] == t32;
          uint32_t eqTot;
          const uint32_t eqRk = pqt_ballot_rank(isEq, &eqTot);
          const bool takeEq = isEq && eqSeen + eqRk < needEq;
          const bool take = (v && k32[r] < t32) || takeEq;
          uint32_t tot;
          const uint32_t rk = pqt_ballot_rank(take, &tot);
          if (take) sKeys[cnt + rk] = ((uint64_t)k32[r] << 32) | e;
          // position of the last tie that was taken = low word of the largest kept key (the threshold of the second phase)
          const unsigned long long tm = __ballot(takeEq);
          if (tm) lastEq = (uint32_t)r * 64u + (63u - (uint32_t)__builtin_clzll(tm));
          eqSeen += eqTot;
          cnt += tot;
        }
      }
      tau = ((uint64_t)t32 << 32) | lastEq;
      kept = kSel;
    } else {
#pragma unroll
      for (int r = 0; r < R32; ++r) {
        const uint32_t e = (uint32_t)r * 64u + lane;
        if ((uint32_t)r * 64u < have && e < have) sKeys[e] = ((uint64_t)k32[r] << 32) | e;
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (final) {
      constexpr int FR = (int)(BESTN / 64);
      uint64_t key[FR];
#pragma unroll
      for (int r = 0; r < FR; ++r) {
        const uint32_t e = lane * FR + r;
        key[r] = (e < kept) ? sKeys[e] : ~0ull;
      }
      if (!(dbg & 1)) pqt_wave_sort_u64<FR>(key);
#pragma unroll
      for (int r = 0; r < FR; ++r) sKeys[lane * FR + r] = key[r];
      __builtin_amdgcn_wave_barrier();
    }
    npend = 0;
    off0 = kept;
    phase1 = false;
  };

  if constexpr (PRE == 3) {
    // the scan launch left the best list (ascending) in global memory
    off0 = A.preCnt[q];
    if (off0 == 0xffffffffu) return;  // handed back by the scan
#pragma unroll
    for (int r = 0; r < 4; ++r) { const uint32_t e = r * 64 + lane; if (e < off0) sKeys[e] = A.preKeys[(size_t)q * 256 + e]; }
    __builtin_amdgcn_wave_barrier();
  } else
  for (uint32_t base = jStart;; base += jStep) {
    if (base < n) {
      if (tstamp) ts0 = __builtin_readcyclecounter();
      uint32_t id[U];
      float accPre[U];
      (void)accPre;
      if constexpr (PRE) {
        if (usePre) {
          // head of the request queue (asked for PRE_DEPTH batches ago), the queue moves up, the batch PRE_DEPTH ahead is requested
#pragma unroll
          for (int u = 0; u < U; ++u) accPre[u] = preQ[0][u];
#pragma unroll
          for (int dq = 0; dq + 1 < PRE_DEPTH; ++dq) {
#pragma unroll
            for (int u = 0; u < U; ++u) preQ[dq][u] = preQ[dq + 1][u];
          }
          // (requested unconditionally, clamped to the list: with a conditional request the compiler cannot count the requests in flight
          // and waits for ALL of them at every use -- the queue would hide nothing)
          const uint32_t ahead = base + (uint32_t)PRE_DEPTH * 64u * U;
#pragma unroll
          for (int u = 0; u < U; ++u) { const uint32_t j = ahead + u * 64 + lane; preQ[PRE_DEPTH - 1][u] = preRow[j < n ? j : n - 1]; }
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t j = base + u * 64 + lane;
        id[u] = idNext[u];  // position in the bin-ordered line store (requested one batch ago)
        if constexpr (kRuns && !PRE) { if (useRuns) { const uint32_t p0 = expand64(base + u * 64); id[u] = j < n ? p0 : 0u; } }
        if (dbg & 16) id[u] = (j & 1023u);  // debug: cache-resident rows (results wrong)
      }
      uint4 rows[U][LPV];
      float rbias[U];
      (void)rbias;
      if constexpr (PRE) {
        // nothing to fetch: the distances are in accPre
      } else
      if constexpr (MODE != 0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
          for (int v = 0; v < LPV; ++v) rows[u][v] = A.codesGrp4[(size_t)v * A.nIds + id[u]];
          rbias[u] = A.bias[id[u]];
        }
      } else if constexpr (LPV == 4 && PQT_RS_QUAD) {
        // 64-byte rows, ONE access per row: the 4 lanes of a quad read the 4 consecutive 16-byte pieces of one row per
        // instruction (instruction i fetches the row of the quad's lane i), so a wave instruction touches 16 rows once
        // instead of 64 rows a quarter each -- with one lane per row every row was requested four times, three of them
        // hits on a line whose fill was still pending (r01: TCP_PENDING_STALL 49 % of cycles, 15.3 M L1 accesses for
        // 2.67 M L2 requests).  A 4x4 transpose inside the quad (two butterfly steps of DPP quad_perm moves) then gives
        // every lane the 4 pieces of its own row.
        const uint32_t c = lane & 3u;
        const bool odd = (c & 1u) != 0, hi2 = (c & 2u) != 0;
        auto qbcast = [](uint32_t v, const int i) -> uint32_t {
          return i == 0 ? (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x00, 0xf, 0xf, true) : i == 1 ? (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x55, 0xf, 0xf, true)
               : i == 2 ? (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xAA, 0xf, 0xf, true) : (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xFF, 0xf, 0xf, true);
        };
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const uint32_t idi = qbcast(id[u], i);
            rows[u][i] = reinterpret_cast<const uint4*>(codes + (size_t)idi * LP)[c];  // piece c of the row of quad lane i
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          // T[c][i] = rows[u][i] at lane c  ->  want T[i][c].  Step A exchanges with lane^1 the elements with (i&1) != (c&1),
          // step B with lane^2 those with (i>>1) != (c>>1).
          auto xchg = [&](uint4& a, uint4& b, const bool upper, auto lx) {
            // the lower lane of the pair keeps a and receives the partner's a into b; the upper lane keeps b and receives
            // the partner's b into a
            const uint4 send = upper ? a : b;
            const uint4 recv = make_uint4(lx(send.x), lx(send.y), lx(send.z), lx(send.w));
            if (upper) a = recv; else b = recv;
          };
          auto x1 = [](uint32_t v) { return pqt_lane_xor_u32<1>(v); };
          auto x2 = [](uint32_t v) { return pqt_lane_xor_u32<2>(v); };
          xchg(rows[u][0], rows[u][1], odd, x1);
          xchg(rows[u][2], rows[u][3], odd, x1);
          xchg(rows[u][0], rows[u][2], hi2, x2);
          xchg(rows[u][1], rows[u][3], hi2, x2);
        }
      } else {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint4* row4 = reinterpret_cast<const uint4*>(codes + (size_t)id[u] * LP);
#pragma unroll
        for (int v = 0; v < LPV; ++v) rows[u][v] = row4[(dbg & 1024) ? 0 : v];  // debug bit 1024: one 16-byte piece per row (results wrong)
      }
      }
      if constexpr (!PRE) {
      if (!useRuns && base + jStep < n) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint32_t j = base + jStep + u * 64 + lane;
          idNext[u] = cid[j < n ? j : n - 1];
        }
      }
      }
      if (tstamp) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); const unsigned long long t = __builtin_readcyclecounter(); tsLoad += t - ts0; ts0 = t; }
      float accX[U];
      (void)accX;
      if constexpr (XC) {
        static_assert(MODE == 0 && C1M >= 2 && COARSE_LDS && (UREQ % 2) == 0, "X-code rows: exact rerank with the LDS table, compile-time C1, candidates in pairs");
        constexpr uint32_t kBmask = 4u * ((1u << C1M) - 1u);  // B*4 sits in bits 2 .. C1M+1, A in bits C1M+2 .. 2*C1M+1 of the low half
        const pqt_f2 kOff = {-1028.f, -1028.f};
        // absolute LDS byte addresses (the dynamic segment's base is a link-time constant the compiler otherwise adds to every
        // address computed from smem_raw: one v_add per look-up); the L1virt copy of a wavefront is 4*C1*LP-aligned inside the
        // segment, so with the segment at a 4*C1-aligned base (0: no static LDS in these kernels) OR-ing the centroid offset in is exact
        typedef __attribute__((address_space(3))) const float* lds_f32p;
        const uint32_t vAbs = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const unsigned char*)reinterpret_cast<const unsigned char*>(sVirt);
        const uint32_t cAbs = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const unsigned char*)smem_raw;
        if (vAbs & kBmask) __builtin_trap();  // (uniform; cannot happen with the launchers' LDS layouts: the kernel has no static LDS in front of the dynamic segment)
#pragma unroll
        for (int u = 0; u < U; u += 2) {
          pqt_f2 acc2 = {0.f, 0.f};
#pragma unroll
          for (int v = 0; v < LPV; ++v) {
            const uint32_t w0[4] = {rows[u][v].x, rows[u][v].y, rows[u][v].z, rows[u][v].w};
            const uint32_t w1[4] = {rows[u + 1][v].x, rows[u + 1][v].y, rows[u + 1][v].z, rows[u + 1][v].w};
#pragma unroll
            for (int x = 0; x < 4; ++x) {
              const uint32_t p = v * 4 + x;
              pqt_f2 sb2, sa2, sc2, lam2;
#pragma unroll
              for (int h = 0; h < 2; ++h) {
                const uint32_t ww = h ? w1[x] : w0[x];
                const uint32_t aC = (ww & 0xffffu) + cAbs;                                   // v_add_sdwa: coarse[p][A][B] inside the part's table
                const uint32_t bV = (ww & kBmask) | vAbs;                                    // v_and_or: L1virt[p][B]
                const uint32_t aV = ((ww >> C1M) & kBmask) | vAbs;                           // v_lshrrev + v_and_or: L1virt[p][A]
                // 1024 + u16 * 2^-13 assembled in the mantissa (one byte permute; v_cvt_f32_u32 occupies the wavefront for 8 cycles, a
                // plain VALU op for 5): lambda = that - 1028, exact like the reference's u16 * (8 / 65536) - 4
                lam2[h] = __uint_as_float(__builtin_amdgcn_perm(ww, 0x44800000u, 0x03020706u));
                sb2[h] = *(lds_f32p)(uintptr_t)(aV + p * (4u << C1M));
                sa2[h] = *(lds_f32p)(uintptr_t)(bV + p * (4u << C1M));
                sc2[h] = *(lds_f32p)(uintptr_t)(aC + p * (4u << (2 * C1M)));
              }
              lam2 = lam2 + kOff;  // == pqt_lambda_decode
              const pqt_f2 d2 = sb2 + lam2 * lam2 * sc2 + lam2 * (sa2 - sb2 - sc2);  // pqt_extract_distance per candidate
              acc2 = acc2 + d2;
            }
          }
          accX[u] = acc2[0];
          accX[u + 1] = acc2[1];
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t j = base + u * 64 + lane;
        const bool valid = j < n;
        float acc = 0.f;
        if constexpr (XC) acc = accX[u];
        else if constexpr (PRE) acc = accPre[u];
        else
        if (dbg & 8) {  // debug: no ADC arithmetic, the rows are still fetched and consumed (results wrong)
          uint32_t x = 0;
#pragma unroll
          for (int v = 0; v < LPV; ++v) x ^= rows[u][v].x ^ rows[u][v].y ^ rows[u][v].z ^ rows[u][v].w;
          acc = __uint_as_float(x & 0x3fffffffu);
        } else
#pragma unroll
        for (int v = 0; v < LPV; ++v) {
          const uint32_t w[4] = {rows[u][v].x, rows[u][v].y, rows[u][v].z, rows[u][v].w};
          // two line parts at a time: the per-part arithmetic in packed FP32 (v_pk_mul_f32 / v_pk_add_f32: IEEE per component, nothing
          // fused, the association of extractDistance unchanged), the running sum still one part after the other -- bit-identical keys
          // with ~a quarter fewer VALU instructions in the phase that bounds this kernel at the SIFT1M shape (3 wavefronts per SIMD)
#pragma unroll
          for (int x = 0; x < 4; x += 2) {
            pqt_f2 sb2, sa2, sc2, lam2;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const uint32_t p = v * 4 + x + h;
              const uint32_t ww = w[x + h];
              const uint32_t A = ww & 0xffu, B = (ww >> 8) & 0xffu;
              lam2[h] = (float)(ww >> 16);
              if constexpr (MODE != 0) {
                const uint32_t pv = C1P2 ? (p << c1sh) : p * C1;
                sb2[h] = sVirt[pv + A];
                sa2[h] = sVirt[pv + B];
              } else if constexpr (C1M >= 2 && COARSE_LDS) {
                // byte offsets with compile-time strides: one add per L1virt address, one shift-add for the coarse one,
                // the part offsets go into the instructions' immediate fields
                const uint32_t B4 = B << 2;
                const uint32_t aV = (A << 2) + vOff, bV = B4 + vOff, aC = (A << (2 + C1M)) + B4;  // v_lshl_add, v_add, v_lshl_add
                sb2[h] = *reinterpret_cast<const float*>(smem_raw + aV + p * (4u << C1M));
                sa2[h] = *reinterpret_cast<const float*>(smem_raw + bV + p * (4u << C1M));
                sc2[h] = *reinterpret_cast<const float*>(smem_raw + aC + p * (4u << (2 * C1M)));
              } else {
                sb2[h] = sVirt[(C1P2 ? (p << c1sh) : p * C1) + A];
                sa2[h] = sVirt[(C1P2 ? (p << c1sh) : p * C1) + B];
                sc2[h] = cz[C1P2 ? ((((p << c1sh) + A) << c1sh) + B) : ((p * C1 + A) * C1 + B)];
              }
            }
            const pqt_f2 kScale = {8.f / 65536.f, 8.f / 65536.f}, kOff = {-4.f, -4.f};
            lam2 = lam2 * kScale + kOff;  // == pqt_lambda_decode: the product is exact, so mul + add rounds like the scalar fma
            if constexpr (MODE != 0) {
              const pqt_f2 d2 = sb2 + lam2 * (sa2 - sb2);
              acc = acc + d2[0];
              acc = acc + d2[1];
            } else {
              const pqt_f2 d2 = sb2 + lam2 * lam2 * sc2 + lam2 * (sa2 - sb2 - sc2);  // pqt_extract_distance per component
              acc = acc + d2[0];
              acc = acc + d2[1];
            }
          }
        }
        if constexpr (MODE != 0 && !PRE) acc = acc + rbias[u];
        // visiting position is the tie-break; sharded lists keep j as the low word (positions are monotone in j)
        if (kPhase1 && phase1) {
          if (valid) sK32[j] = pqt_f2key(acc);
        } else {
        const uint64_t key = ((uint64_t)pqt_f2key(acc) << 32) | j;
        const bool pass = valid && key < tau;
        uint32_t tot;
        const uint32_t rk = pqt_ballot_rank(pass, &tot);
        if (pass) sKeys[off0 + npend + rk] = key;
        npend += tot;
        }
      }
      __builtin_amdgcn_wave_barrier();
      if (tstamp) { const unsigned long long t = __builtin_readcyclecounter(); tsAdc += t - ts0; }
    }
    // single flush site: when the pending buffer could overflow on the next batch, and once at the end
    const bool last = base + jStep >= n;
    if (last && qN != 0xffffffffu && nN == 0xffffffffu) nN = (dbg & 2) ? 0u : nLocal[qN];
    if (kPhase1 && phase1) {
      const uint32_t seen = base + 64 * U < n ? base + 64 * U : n;  // candidates evaluated so far = keys in the 32-bit slots (MODE 0: never COOP, jStep = 64 U)
      if (last || seen + 64 * U > CAP32) {
        if (tstamp) ts0 = __builtin_readcyclecounter();
        flush32(seen, last);
        if (tstamp) tsFlush += __builtin_readcyclecounter() - ts0;
      }
    } else
    if (last || off0 + npend + 64 * U > SLOTS) {
      if (tstamp) ts0 = __builtin_readcyclecounter();
      flush(last);
      if (tstamp) tsFlush += __builtin_readcyclecounter() - ts0;
    }
    if (last) break;
  }
  if constexpr (COOP) {
    // pair scan: this wavefront's best list (<= 256 keys of ITS batches) and, from half 0, the largest table entry (the band's error bound)
#pragma unroll
    for (int r = 0; r < 4; ++r) { const uint32_t e = r * 64 + lane; if (e < off0) A.preKeys[((size_t)q * 2 + coopHalf) * 256 + e] = sKeys[e]; }
    if (lane == 0) { A.preCnt[(size_t)q * 2 + coopHalf] = off0; if (coopHalf == 0) const_cast<float*>(A.preQmax)[q] = qmax; }
    return;
  }
  if constexpr (PRE == 2) {
    // scan launch: the best list (<= 256 keys, ascending after the final flush) goes to global memory for the band launch
    static_assert(MODE == 2, "the split selection is the filtered one");
#pragma unroll
    for (int r = 0; r < 4; ++r) { const uint32_t e = r * 64 + lane; if (e < off0) A.preKeys[(size_t)q * 256 + e] = sKeys[e]; }
    if (lane == 0) A.preCnt[q] = off0;
    return;
  }
  // results: first min(k, n) entries of the best list
  const uint32_t kk = n < k ? n : k;
  uint32_t ties = 0;
  if (tstamp) ts0 = __builtin_readcyclecounter();
  if constexpr (MODE == 2) {
    // the list holds the off0 (<= 256) smallest MODE 1 keys in ascending order: re-evaluate the band [.., d1_(k) + 2 eps]
    // with the reference association and order it by the exact key
    if (kk) {
      const float eps = A.epsKappa * (5.f * qmax + A.cmax20) * 1.001f;
      const float dk = pqt_key2f((uint32_t)(sKeys[kk - 1] >> 32));
      const float thr = dk + 2.f * eps + (dk < 0.f ? -dk : dk) * 1e-6f;
      const uint32_t thrKey = pqt_f2key(thr);
      uint32_t nT = 0;  // entries inside the band (a prefix of the sorted list)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const uint32_t e = r * 64 + lane;
        const bool in = e < off0 && (uint32_t)(sKeys[e] >> 32) <= thrKey;
        nT += (uint32_t)__popcll(__ballot(in));
      }
      if (nT == BESTN && n > BESTN) {
        // the band reaches the end of a full list: candidates outside the list may belong to it -> plain exact kernel
        if (lane == 0) A.fbList[atomicAdd(A.fbCount, 1u)] = q;
        __builtin_amdgcn_wave_barrier();
        return;
      }
      uint64_t xk[4];
      // (the first version evaluated an entry term by term: the compiler emitted one coarse[] gather + s_waitcnt per term, 32
      // serialised L2 round trips per entry and 41 k clocks per query; now an entry's code row is requested first, then all its
      // LP coarse values together, and only the sum runs in order)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        xk[r] = ~0ull;
        if ((uint32_t)r * 64u < nT) {  // uniform
          const uint32_t e = r * 64 + lane;
          const bool act = e < nT;
          const uint32_t j = act ? (uint32_t)sKeys[e] : 0u;
          uint32_t posj = 0;
          if (act) {
            if (useRuns) { const unsigned long long rr = sRuns[runOf(j)]; posj = (uint32_t)(rr >> 32) + (j - (uint32_t)rr); }
            else posj = cid[j];
          }
          const uint4* row4 = reinterpret_cast<const uint4*>(codes + (size_t)posj * LP);
          uint4 rv[LPV];
#pragma unroll
          for (int v = 0; v < LPV; ++v) rv[v] = row4[v];
          float scv[LP];
#pragma unroll
          for (int v = 0; v < LPV; ++v) {
            const uint32_t w[4] = {rv[v].x, rv[v].y, rv[v].z, rv[v].w};
#pragma unroll
            for (int x = 0; x < 4; ++x) {
              const uint32_t p = v * 4 + x;
              const uint32_t Aa = w[x] & 0xffu, Bb = (w[x] >> 8) & 0xffu;
              const uint32_t pv = C1P2 ? (p << c1sh) : p * C1;
              scv[p] = A.coarse[C1P2 ? (((pv + Aa) << c1sh) + Bb) : ((pv + Aa) * C1 + Bb)];
            }
          }
          // all LP gathers are in flight here: keep the compiler from sinking each one next to its use (it did, to shorten live
          // ranges under the kernel's register budget: one gather + s_waitcnt vmcnt(0) per term again)
          asm volatile("" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
          float acc = 0.f;
#pragma unroll
          for (int v = 0; v < LPV; ++v) {
            const uint32_t w[4] = {rv[v].x, rv[v].y, rv[v].z, rv[v].w};
#pragma unroll
            for (int x = 0; x < 4; ++x) {
              const uint32_t p = v * 4 + x;
              const uint32_t Aa = w[x] & 0xffu, Bb = (w[x] >> 8) & 0xffu;
              const float lam = __builtin_fmaf((float)(w[x] >> 16), 8.f / 65536.f, -4.f);
              const uint32_t pv = C1P2 ? (p << c1sh) : p * C1;
              const float sb = sVirt[pv + Aa], sa = sVirt[pv + Bb];
              acc = acc + pqt_extract_distance(sa, sb, scv[p], lam);
            }
          }
          if (act) xk[r] = ((uint64_t)pqt_f2key(acc) << 32) | j;
        }
      }
      __builtin_amdgcn_wave_barrier();
      // blocked layout for the network: element e = lane*4 + r; go through LDS (pending area) to re-distribute
#pragma unroll
      for (int r = 0; r < 4; ++r) sKeys[BESTN + r * 64 + lane] = xk[r];
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int r = 0; r < 4; ++r) xk[r] = sKeys[BESTN + lane * 4 + r];
      pqt_wave_sort_u64<4>(xk);
#pragma unroll
      for (int r = 0; r < 4; ++r) sKeys[lane * 4 + r] = xk[r];
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (tstamp) { const unsigned long long t = __builtin_readcyclecounter(); tsVerify = t - ts0; ts0 = t; }
  {
    // k <= 128: a lane owns the result slots lane and lane + 64; their store positions, then their ids (and positions) are
    // requested together -- two round trips for the whole result list instead of two per slot
    static_assert(PQT_RS_BEST <= 128, "two result slots per lane");
    uint32_t sp[2] = {0u, 0u}, gp[2] = {0xffffffffu, 0xffffffffu}, dk[2] = {0u, 0u};
    bool live[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const uint32_t i = lane + 64 * u;
      live[u] = i < kk;
      if (live[u]) {
        const uint64_t key = sKeys[i];
        const uint32_t j = (uint32_t)key;
        dk[u] = (uint32_t)(key >> 32);
        if (useRuns) {
          const uint32_t ri = runOf(j);
          const unsigned long long r = sRuns[ri];
          sp[u] = (uint32_t)(r >> 32) + (j - (uint32_t)r);
          if (SHARDED) gp[u] = sRunG[ri] + (j - (uint32_t)r);
        } else {
          sp[u] = j;  // resolved through cid[] below
        }
        if (i + 1 < kk && (uint32_t)(sKeys[i + 1] >> 32) == dk[u]) ++ties;
      }
    }
    if (!useRuns) {
      uint32_t c2[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) c2[u] = live[u] ? cid[sp[u]] : 0u;
      if (SHARDED) {
#pragma unroll
        for (int u = 0; u < 2; ++u) if (live[u]) gp[u] = cpos[sp[u]];
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) sp[u] = c2[u];
    }
    uint32_t idv[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) idv[u] = live[u] ? ids[sp[u]] : 0xffffffffu;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const uint32_t i = lane + 64 * u;
      if (i < k) {
        const size_t o = (size_t)q * k + i;
        outIdx[o] = idv[u];
        outDist[o] = live[u] ? pqt_key2f(dk[u]) : __uint_as_float(0x7f800000u);
        if (SHARDED) outPos[o] = gp[u];
      }
    }
  }
  tiesAcc += ties;  // flushed once per wavefront by the kernel (pqt_count_ties)
  __builtin_amdgcn_wave_barrier();
  if (tstamp) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); tsOut = __builtin_readcyclecounter() - ts0; }
  if (tstamp && lane == 0) {
    tstamp[(size_t)q * PQT_TS_WORDS + 16] = tsSetup; tstamp[(size_t)q * PQT_TS_WORDS + 17] = tsVerify; tstamp[(size_t)q * PQT_TS_WORDS + 18] = tsOut; tstamp[(size_t)q * PQT_TS_WORDS + 19] = n;
    tstamp[(size_t)q * PQT_TS_WORDS + 10] = tsLoad; tstamp[(size_t)q * PQT_TS_WORDS + 11] = tsAdc; tstamp[(size_t)q * PQT_TS_WORDS + 12] = tsFlush;
    // [13] = shader clocks of this query | start on the 100 MHz wall clock << 32; [14] = wave slot | XCC id << 16 | end on the
    // wall clock << 32 (the cycle counters of different XCDs have different origins, the wall clock is global)
    tstamp[(size_t)q * PQT_TS_WORDS + 13] = ((__builtin_readcyclecounter() - tsStart) & 0xffffffffull) | (wallStart << 32);
    tstamp[(size_t)q * PQT_TS_WORDS + 14] = (unsigned long long)slot | ((unsigned long long)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xfu) << 16) | (wall_clock64() << 32);
  }
}

template <int NW, int LPV, int UREQ, bool COARSE_LDS, bool SHARDED, int C1M /* 0: any C1, 1: power of two, >= 2: C1 == 1 << C1M at compile time */,
          int MODE = 0, bool RUNS = false, bool XC = false /* A.codes = the X-code copy of the store, see pqt_rs_query */,
          int NSLOT = PQT_RS_BEST + PQT_RS_PEND /* key slots per wavefront */>
// (the X-code kernel keeps its 128-VGPR budget whatever NW is: with fewer than 16 wavefronts the registers it leaves belong to the other
// batch's traversal wavefronts when two batches are in flight)
__global__ __launch_bounds__(NW * 64, XC ? 4 : 1) void pqt_k_rerank_select(const PqtRsArgs A) {
  const float* __restrict__ coarse = A.coarse; const uint32_t* __restrict__ nLocal = A.nLocal; const uint32_t qn = A.qn;
  const PqtDevParams& prm = A.prm; const uint32_t dbg = A.dbg; const uint32_t dynamic = A.dynamic; unsigned long long* __restrict__ zero8 = A.zero8;
  constexpr uint32_t LP = LPV * 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const uint32_t C1 = C1M >= 2 ? (1u << C1M) : prm.C1;
  const uint32_t nCoarse = COARSE_LDS ? LP * C1 * C1 : 0;
  float* sCoarse = (float*)smem_raw;
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  uint64_t* sKeys = (uint64_t*)(smem_raw + (size_t)nCoarse * 4) + (size_t)wave * NSLOT;
  float* sVirt = (float*)(smem_raw + (size_t)nCoarse * 4 + (size_t)NW * NSLOT * 8) + (size_t)wave * LP * C1;
  const size_t ticketOff = (size_t)nCoarse * 4 + (size_t)NW * ((size_t)NSLOT * 8 + (size_t)LP * C1 * 4);
  // Schedule.  Candidate counts differ several-fold between queries and wavefronts do not run equally fast (the
  // younger of two wavefronts on a SIMD loses the issue arbitration): with a static round-robin over the wavefronts the
  // launch lasted as long as its unluckiest wavefront, 50-60 % above the mean (debug timestamps).  So a workgroup owns
  // the queries b, b + G, b + 2G, ... (G workgroups) and its wavefronts draw them through a ticket counter in LDS,
  // longest first: wavefront 0 ranks the first PQT_RS_LIST of them by candidate count while the others copy the coarse
  // table.  Tickets in global memory were measured and cost more than the imbalance they remove (same-address
  // returning atomics serialise at ~0.15 us each even inside one XCD's L2), and a global longest-first order with
  // serpentine workgroup lists (a separate 8 us kernel) gained less than it cost.
  const uint32_t G = gridDim.x;
  uint32_t* sTicket = reinterpret_cast<uint32_t*>(smem_raw + ticketOff);
  uint32_t* sList = sTicket + 4;             // PQT_RS_LIST: ordinal in the workgroup's list, longest first
  uint32_t* sListN = sList + PQT_RS_LIST;    // PQT_RS_LIST: its candidate count
  uint32_t* sTmpN = sListN + PQT_RS_LIST;    // PQT_RS_LIST: counts in list order (ranking input)
  // per-wave bin-run area (MODE 0 with the LDS table only; the launcher sizes the LDS accordingly)
  unsigned long long* sRuns = (RUNS && A.runs)
      ? reinterpret_cast<unsigned long long*>(smem_raw + ((ticketOff + 16 + 3 * PQT_RS_LIST * 4 + 15) & ~(size_t)15)) + (size_t)wave * (A.runCap + A.runCap / 2)
      : nullptr;
  const uint32_t L = (dynamic == 1 && blockIdx.x < qn) ? (qn - blockIdx.x + G - 1) / G : 0u;
  const uint32_t Ls = L < PQT_RS_LIST ? L : PQT_RS_LIST;
  if (threadIdx.x < 4) sTicket[threadIdx.x] = 0;  // head, tail, done, lock
  if (blockIdx.x == 0 && threadIdx.x < 8 && zero8) zero8[threadIdx.x] = 0;
  if (blockIdx.x == 0 && A.poolNext) for (uint32_t t = threadIdx.x; t < 16u + 8u * PQT_SCHED_CLASSES; t += NW * 64) A.poolNext[t] = 0;  // draw counters + registration counts
  // ---- dynamic == 2: per-XCD pools in longest-first order, a static share and a dynamic remainder.
  // The per-query time differs several-fold, workgroups with a fixed share finish up to 25 % apart, and four of the eight
  // XCDs see ~25 % slower memory (scripts/micro/xcd_latency.hip, profiles/r02_xcd_latency.txt): with a fixed partition the
  // launch lasts as long as the slowest workgroup of the slow XCDs.  Pool x = the queries q with q % 8 == x (their
  // traversal ran on XCD x: candidate list and L1virt are in that L2), ordered by the traversal's registration lists --
  // size class by size class, largest first.  Entry i of the pool's first PQT_RS_STATIC_PCT % belongs to workgroup i % nW of
  // the pool (an interleaved longest-first share, taken without any atomic); the rest -- the short queries -- is drawn in
  // chunks of 6 .. 1 (shrinking with what the pool has left, like the ring's low-water mark) with ONE device-scope atomic per
  // chunk, by whichever wavefront finds the workgroup's LDS ring low (a lock word keeps it to one at a time; the others keep
  // taking tickets), first from the own pool, then from whichever pool has most left.
  uint32_t* sPoolCur = sTmpN;      // [0] pool drawn from, [1] next chunk size, [2] low-water mark
  uint32_t* sIncl = sTmpN + 4;     // 64: inclusive class counts of the pool being read
  constexpr uint32_t kLow = 6;     // request the next chunk when at most this many undrawn entries are left in the ring (fewer near a pool's end)
  uint32_t* sPoolTot = sTmpN + 72;  // 8: queries registered in each pool (from the traversal's counts)
  auto poolCount = [&](const uint32_t x) -> uint32_t { return *(volatile uint32_t*)&sPoolTot[x & 7u]; };
  auto poolWgs = [&](const uint32_t x) -> uint32_t { return x < G ? (G - x + 7u) / 8u : 0u; };
  // entries of the static share per workgroup of pool x (the ring holds 256)
  auto poolStatic = [&](const uint32_t x) -> uint32_t { const uint32_t w = poolWgs(x); uint32_t j = w ? (poolCount(x) * PQT_RS_STATIC_PCT / 100u) / w : 0u; return j > 192u ? 192u : j; };
  // chunk size and low-water mark shrink with what the pool has left: at the end nothing sits reserved in a ring while
  // other workgroups run dry (a query is a third of a wavefront's share of the launch: reservations are expensive there)
  auto chunkOf = [&](const uint32_t rem) -> uint32_t { const uint32_t c = rem / 128u; return c < 1u ? 1u : (c > 6u ? 6u : c); };
  auto lowOf = [&](const uint32_t rem) -> uint32_t { const uint32_t c = rem / 64u; return c > kLow ? kLow : c; };
  // lanes < cnt: ring[tail + lane] = entry idx (per lane) of pool x; then publishes tail + cnt
  auto fetchEntries = [&](const uint32_t x, const uint32_t idx, const uint32_t cnt, const uint32_t tail) {
    const uint32_t cl = PQT_SCHED_CLASSES - 1u - lane;  // classes in descending order: lane l looks at class 63 - l
    const uint32_t cn = A.schedCnt[x * PQT_SCHED_CLASSES + cl];
    const uint32_t incl = pqt_wave_incl_scan(cn);
    *(volatile uint32_t*)&sIncl[lane] = incl;
    __builtin_amdgcn_wave_barrier();
    if (lane < cnt) {
      uint32_t lo = 0;  // first lane whose inclusive count exceeds idx (6-step search)
#pragma unroll
      for (uint32_t st = 32; st >= 1; st >>= 1) { if (lo + st <= 63u && *(volatile uint32_t*)&sIncl[lo + st - 1] <= idx) lo += st; }
      const uint32_t ex = lo ? *(volatile uint32_t*)&sIncl[lo - 1] : 0u;
      const uint32_t within = idx - ex < A.schedCap ? idx - ex : A.schedCap - 1u;
      const unsigned long long ev = A.schedList[(size_t)(x * PQT_SCHED_CLASSES + (PQT_SCHED_CLASSES - 1u - lo)) * A.schedCap + within];
      const uint32_t qe = (uint32_t)ev;  // (clamped: a registration block left dirty by a failed call must not send a wave out of bounds)
      sList[(tail + lane) & (PQT_RS_LIST - 1)] = qe < qn ? qe : qn - 1u;
      sListN[(tail + lane) & (PQT_RS_LIST - 1)] = qe < qn ? (uint32_t)(ev >> 32) : 0u;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_store(&sTicket[1], tail + cnt, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    __builtin_amdgcn_wave_barrier();
  };
  // appends one chunk of the dynamic remainder to the ring (caller holds the lock); sets done when every pool is exhausted
  auto refill = [&]() {
    uint32_t x = *(volatile uint32_t*)&sPoolCur[0], ch = *(volatile uint32_t*)&sPoolCur[1];
    uint32_t got = 0, first = 0;
    bool exhausted = false;
    for (int attempt = 0; attempt < 4; ++attempt) {  // (a draw that loses the race for a pool's last entries is retried by the caller)
      const uint32_t dyn0 = poolWgs(x) * poolStatic(x);
      const uint32_t cx = poolCount(x) - dyn0;  // entries of the pool's dynamic remainder
      uint32_t c = cx;
      if (cx) { if (lane == 0) c = __hip_atomic_fetch_add(&A.pool[x], ch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c); }
      if (c < cx) {
        got = cx - c < ch ? cx - c : ch; first = dyn0 + c;
        if (lane == 0) { *(volatile uint32_t*)&sPoolCur[0] = x; *(volatile uint32_t*)&sPoolCur[1] = chunkOf(cx - c - got); *(volatile uint32_t*)&sPoolCur[2] = lowOf(cx - c - got); }
        break;
      }
      // this pool is empty: the one with most left (one round trip for all eight counters)
      uint32_t rem = 0;
      if (lane < 8) {
        const uint32_t ci = poolCount(lane) - poolWgs(lane) * poolStatic(lane);
        const uint32_t di = __hip_atomic_load(&A.pool[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        rem = di < ci ? ci - di : 0u;
      }
      uint32_t best = (rem << 3) | (7u - lane);
#pragma unroll
      for (int d = 4; d > 0; d >>= 1) { const uint32_t o = __shfl_xor(best, d, 64); best = o > best ? o : best; }
      best = (uint32_t)__builtin_amdgcn_readfirstlane((int)best);
      if ((best >> 3) == 0) { exhausted = true; break; }  // nothing left anywhere
      x = 7u - (best & 7u);
      ch = chunkOf(best >> 3);
    }
    if (got) fetchEntries(x, first + lane, got, __hip_atomic_load(&sTicket[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    else if (exhausted && lane == 0) __hip_atomic_store(&sTicket[2], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  if (dynamic == 2 && wave == 0) {
    for (uint32_t xx = 0; xx < 8; ++xx) {
      uint32_t c = A.schedCnt[xx * PQT_SCHED_CLASSES + lane];
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d, 64);
      if (lane == 0) *(volatile uint32_t*)&sPoolTot[xx] = c;
    }
    __builtin_amdgcn_wave_barrier();
    const uint32_t x = blockIdx.x & 7u, w = poolWgs(x), J = poolStatic(x), i = blockIdx.x >> 3;
    if (lane == 0) { *(volatile uint32_t*)&sPoolCur[0] = x; *(volatile uint32_t*)&sPoolCur[1] = chunkOf(poolCount(x) - w * J); *(volatile uint32_t*)&sPoolCur[2] = lowOf(poolCount(x) - w * J); }
    __builtin_amdgcn_wave_barrier();
    for (uint32_t j0 = 0; j0 < J; j0 += 64) fetchEntries(x, i + w * (j0 + lane), J - j0 < 64u ? J - j0 : 64u, j0);
  }
  if (dynamic == 1 && wave == 0) {
    uint32_t ne[PQT_RS_LIST / 64];
#pragma unroll
    for (int j = 0; j < PQT_RS_LIST / 64; ++j) {
      const uint32_t e = lane + 64 * j;
      ne[j] = (e < Ls && !(dbg & 2)) ? nLocal[blockIdx.x + (size_t)e * G] : 0u;
      if (e < Ls) sTmpN[e] = ne[j];
    }
    __builtin_amdgcn_wave_barrier();
    uint32_t rank[PQT_RS_LIST / 64];
#pragma unroll
    for (int j = 0; j < PQT_RS_LIST / 64; ++j) rank[j] = 0;
    for (uint32_t f = 0; f < Ls; ++f) {
      const uint32_t nf = sTmpN[f];  // broadcast read
#pragma unroll
      for (int j = 0; j < PQT_RS_LIST / 64; ++j) rank[j] += (nf > ne[j] || (nf == ne[j] && f < lane + 64 * j)) ? 1u : 0u;
    }
#pragma unroll
    for (int j = 0; j < PQT_RS_LIST / 64; ++j) {
      const uint32_t e = lane + 64 * j;
      if (e < Ls) { sList[rank[j]] = e; sListN[rank[j]] = ne[j]; }
    }
  }
  // the workgroup's tie count is summed in the last word of the (by now consumed) ranking scratch: thread 0 belongs to wavefront 0,
  // which is the only reader/writer of sTmpN above
  uint32_t* const sTies = sTmpN + (PQT_RS_LIST - 1);
  if (threadIdx.x == 0) *(volatile uint32_t*)sTies = 0;
  if (COARSE_LDS && !(dbg & 4)) for (uint32_t t = threadIdx.x; t < nCoarse; t += NW * 64) sCoarse[t] = coarse[t];
  __syncthreads();
  const float* cz = COARSE_LDS ? sCoarse : coarse;

  const uint32_t slot = blockIdx.x * NW + wave;
  // next query of this wavefront and its candidate count (0xffffffff: count still to be fetched from global memory)
  uint32_t round = 0;
  auto nextQuery = [&](uint32_t& cnt) -> uint32_t {
    if (dynamic == 2) {
      uint32_t t = 0;
      if (lane == 0) t = atomicAdd(sTicket, 1u);
      t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
      cnt = 0xffffffffu;
      for (;;) {
        uint32_t done = 0, tail = 0;
        if (lane == 0) { done = __hip_atomic_load(&sTicket[2], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); tail = __hip_atomic_load(&sTicket[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
        done = (uint32_t)__builtin_amdgcn_readfirstlane((int)done); tail = (uint32_t)__builtin_amdgcn_readfirstlane((int)tail);
        const uint32_t lw = *(volatile uint32_t*)&sPoolCur[2];
        if (!done && tail <= t + lw) {  // low water: request the next chunk unless somebody already does
          uint32_t won = 0;
          if (lane == 0) won = atomicCAS(&sTicket[3], 0u, 1u) == 0u ? 1u : 0u;
          won = (uint32_t)__builtin_amdgcn_readfirstlane((int)won);
          if (won) {
            uint32_t d2 = 0, t2 = 0;
            if (lane == 0) { d2 = __hip_atomic_load(&sTicket[2], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); t2 = __hip_atomic_load(&sTicket[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
            d2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)d2); t2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)t2);
            if (!d2 && t2 <= t + lw) refill();
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) __hip_atomic_store(&sTicket[3], 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            continue;
          }
        }
        if (t < tail) break;
        if (done) return 0xffffffffu;
        __builtin_amdgcn_s_sleep(4);
      }
      uint32_t qv = 0, nv = 0;
      if (lane == 0) { qv = sList[t & (PQT_RS_LIST - 1)]; nv = sListN[t & (PQT_RS_LIST - 1)]; }
      cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)nv);
      return (uint32_t)__builtin_amdgcn_readfirstlane((int)qv);
    }
    if (dynamic) {
      uint32_t t = 0;
      if (lane == 0) t = atomicAdd(sTicket, 1u);
      t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
      if (t >= L) return 0xffffffffu;
      uint32_t e = t;
      cnt = 0xffffffffu;
      if (t < Ls) { e = sList[t]; cnt = sListN[t]; }
      return blockIdx.x + e * G;
    }
    const uint64_t nx = (uint64_t)round * G * NW + slot;
    ++round;
    cnt = 0xffffffffu;
    return nx < qn ? (uint32_t)nx : 0xffffffffu;
  };
  // schedule 2 draws the next query when the current one is done (its count comes with the ring entry): nothing is reserved
  // ahead by a wavefront, which matters at the end of the launch; the other schedules choose it now and fetch a count that
  // is not in the LDS list under the final select + sort
  uint32_t n = 0;
  uint32_t tiesAcc = 0;  // ties seen by this lane over all queries of the wavefront
  uint32_t q = nextQuery(n);
  if (q != 0xffffffffu && n == 0xffffffffu) n = nLocal[q];
  if (dbg & 2) n = 0;
  while (q != 0xffffffffu) {
    uint32_t nN = 0, qN = 0xffffffffu;
    if (dynamic != 2) qN = nextQuery(nN);
    pqt_rs_query<LPV, UREQ, COARSE_LDS, SHARDED, C1M, MODE, RUNS, XC, NSLOT>(A, q, n, sKeys, sVirt, cz, qN, nN, slot, tiesAcc, sRuns);
    if (dynamic == 2) { qN = nextQuery(nN); if (dbg & 2) nN = 0; }
    q = qN;
    n = nN;
  }
  // statistics: ONE device-scope atomic per workgroup (3072 same-address atomics at the end of the launch, one per wavefront, still
  // cost 0.015 ms of drain: 0.117 against 0.099 ms with the counter switched off)
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) tiesAcc += (uint32_t)__shfl_xor((int)tiesAcc, d, 64);
  if (lane == 0 && tiesAcc) atomicAdd(sTies, tiesAcc);
  __syncthreads();
  if (threadIdx.x == 0) { const uint32_t t = *(volatile uint32_t*)sTies; if (t) atomicAdd(&A.counters[3], (unsigned long long)t); }
}

// the queries MODE 2 handed back (fbList): plain exact rerank+select (MODE 0, coarse through L2), one wavefront per list entry
template <int NW, int LPV, int UREQ, bool SHARDED, int C1M, bool RUNS = false>
__global__ __launch_bounds__(NW * 64) void pqt_k_rerank_select_list(const PqtRsArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr uint32_t LP = LPV * 4;
  const uint32_t C1 = C1M >= 2 ? (1u << C1M) : A.prm.C1;
  const uint32_t wave = threadIdx.x >> 6;
  uint64_t* sKeys = (uint64_t*)smem_raw + (size_t)wave * (PQT_RS_BEST + PQT_RS_PEND);
  float* sVirt = (float*)(smem_raw + (size_t)NW * (PQT_RS_BEST + PQT_RS_PEND) * 8) + (size_t)wave * LP * C1;
  const uint32_t cnt = *A.qcount;
  uint32_t tiesAcc = 0;
  for (uint32_t e = blockIdx.x * NW + wave; e < cnt; e += gridDim.x * NW) {
    const uint32_t q = A.qlist[e];
    uint32_t nN = 0;
    unsigned long long* sRuns = (RUNS && A.runs)
        ? reinterpret_cast<unsigned long long*>(smem_raw + (((size_t)NW * ((PQT_RS_BEST + PQT_RS_PEND) * 8 + (size_t)LP * C1 * 4) + 15) & ~(size_t)15)) + (size_t)wave * (A.runCap + A.runCap / 2)
        : nullptr;
    pqt_rs_query<LPV, UREQ, false, SHARDED, C1M, 0, RUNS>(A, q, A.nLocal[q], sKeys, sVirt, A.coarse, 0xffffffffu, nN, 0u, tiesAcc, sRuns);
  }
  pqt_count_ties(&A.counters[3], tiesAcc);
}

// ===================================================================================================
// Fused stage a1 + a2 + a4 + a5 + a6 (bound_bins <= 512): the whole traversal of one query by ONE WAVEFRONT.
//
//   workgroup = NW independent wavefronts (no __syncthreads anywhere), each owns a private LDS slice.
//   a1  lane = (centroid, line part) accumulators, sequential over SS dims; W-best by rank counting.
//   a2  lane = one second-level entry (row of cb2), sorted per part by the in-register network (pqt_wave_sort_u64<WCR>).
//   a4  8 heuristic rows per lane: two LDS look-ups per part, uint32 wrap-around bin id, bin-table probe
//       (8 independent 16-byte random reads in flight per lane), keys (f32 key << 32 | row) sorted in registers (512).
//   a6  exclusive scan of the populations in visiting order (wave scan), cut, compact list of the non-empty
//       included bins in LDS, candidates gathered by binary search over that list.
// LDS per wavefront: max(512 * 8 bytes, D + LP*C1 + P*WC words) time-shared + P*C1 + P*W + 2*P*WC words
// (6.5 KB for the SIFT configuration: 6 workgroups of 4 wavefronts per CU).
// ===================================================================================================
// arguments of the per-query traversal (pqt_traverse_query), shared by pqt_k_traverse and the one-launch kernel
struct PqtTravArgs {
  const float* Q; const float* cb1; const float* cb2;
  const float4* cb2T;  // per (p,c1): [S/4][C2] 16-byte vectors, or null when S % 4 != 0
  PqtDevParams prm;
  const uint4* heur8;  // rows of 8 x u16
  const uint32_t* heur4;  // rows of 4 x u8 (P <= 4 and W*C2 <= 256), or null: one dword per row, cheap to request ahead
  uint32_t He, Bv;
  const PqtBinEntry* table; const uint32_t* lower; uint32_t tableBits;
  const uint32_t* ids; uint32_t qn;
  float* qL1virt; uint32_t* cand; uint32_t* candPos;
  uint32_t* nCand; uint32_t* nLocal; uint32_t* nIncl; uint64_t stride;
  unsigned long long* counters; unsigned long long* tstamp;
  float* segDOut; uint32_t* segBOut;    // [q][P][WC]: sorted lists, written when He > 512 (overflow hand-over)
  uint32_t* ovList; uint32_t* ovCount;  // queries handed to pqt_k_bins (He > 512 and > 512 populated rows)
  const uint32_t* filter; uint32_t filterBits;  // presence bitmap over the bin keys, or null
  unsigned long long* runs; uint32_t* runGpos; uint32_t* nRuns;  // bin runs for the rerank (see PqtRsArgs), or null
  uint32_t runCap;  // at most this many runs are handed over (more: the plain candidate list is written)
  uint32_t* outCount;  // the caller's per-query candidate count, written here directly (saves a copy on the stream), or null
  uint32_t tdbg;  // test bits: 1 = order all rows, not just the populated ones; 2 = part lists by the one-list-at-a-time code (option exact_part_sorts)
  // rerank schedule 2: the query registers itself in the list of (XCD pool q % 8, size class of its local candidate count):
  // schedCnt[pool * 64 + class] entries so far, schedList[(pool * 64 + class) * schedCap + i] = i-th query; or null
  uint32_t* schedCnt; unsigned long long* schedList /* query | local candidates << 32 */; uint32_t schedCap;
  // list mode (pqt_query_shard_bins: the queries whose exchanged bin list overflowed are traversed here after all): wavefront i
  // takes query qlist[i], i < *qcount; or null = query i
  const uint32_t* qlist; const uint32_t* qcount;
  // query-sharded traversal (pqt_traverse_bins, SHARDED indices): stop after the cut and hand out the included populated bins
  // in shard-INDEPENDENT form -- gbins[q][gbinCap + 1]: entry i = bin id | global visiting position of its first member << 32
  // (visiting order), trailer word [gbinCap] = number of entries (0xffffffff: more than gbinCap, or the wide traversal's own
  // overflow -- the receiving shard traverses such a query itself) | global candidate count << 32; or null
  unsigned long long* gbins; uint32_t gbinCap;
  // wide enumeration (He > 512), pqt_k_traverse_f1 only: first level of the presence bitmap, folded to 2^filter1Bits bits (bit i = OR of the
  // 2^(filterBits - filter1Bits) bits of `filter` whose index starts with i) -- small enough for the LDS of a workgroup; or null
  const uint32_t* filter1; uint32_t filter1Bits;
  uint32_t f1Compact;  // pqt_k_traverse_f1: 1 = the bitmap is asked for full wavefronts of the rows that passed the first level (round 6), 0 = masked lanes of every row
};

// the whole traversal of query q by the calling wavefront; base = its private LDS slice of perWaveBytes bytes
// SHAPE: 0 = run-time shape; 1 / 2 = the two BASELINE shapes at compile time (1: d=128 p=4 c1=c2=32 w=2 lineparts=16 --
// configs[0]/[1]; 2: d=128 p=4 c1=c2=64 w=1 lineparts=32 -- configs[2]/[3]).  With the trip counts known the compiler
// unrolls the per-dimension loops and issues the centroid reads of a lane's accumulators together; with run-time SS it
// emitted a remainder loop of one 4-byte load + s_waitcnt per dimension (cfg3 shape: 128 serialized round trips per
// query in a1 alone, 56 k of the 228 k clocks of a traversal).
template <int SHAPE> struct PqtShape { static constexpr uint32_t D = 1, P = 1, C1 = 1, C2 = 1, W = 1, LP = 1; };  // run-time shape: unused
template <> struct PqtShape<1> { static constexpr uint32_t D = 128, P = 4, C1 = 32, C2 = 32, W = 2, LP = 16; };
template <> struct PqtShape<2> { static constexpr uint32_t D = 128, P = 4, C1 = 64, C2 = 64, W = 1, LP = 32; };
__host__ __device__ inline int pqt_shape_of(const PqtDevParams& d) {
  if (d.D == 128 && d.P == 4 && d.C1 == 32 && d.C2 == 32 && d.W == 2 && d.LP == 16) return 1;
  if (d.D == 128 && d.P == 4 && d.C1 == 64 && d.C2 == 64 && d.W == 1 && d.LP == 32) return 2;
  return 0;
}

template <int WCR, bool SHARDED, bool P2 /* C1, C2, W, LP, D, S, SS, R all powers of two: shifts and masks instead of
                                            runtime integer divisions (~25 VALU each) and quarter-rate multiplies */,
          int SHAPE = 0>
__device__ __forceinline__ void pqt_traverse_query(const PqtTravArgs& A, const uint32_t q, unsigned char* const base, const uint32_t perWaveBytes,
                                                   const uint32_t* const sF1 = nullptr /* the workgroup's LDS copy of A.filter1 (pqt_k_traverse_f1), or null */) {
  const float* __restrict__ Q = A.Q; const float* __restrict__ cb1 = A.cb1; const float* __restrict__ cb2 = A.cb2;
  const float4* __restrict__ cb2T = A.cb2T; const PqtDevParams& prm = A.prm; const uint4* __restrict__ heur8 = A.heur8;
  const uint32_t He = A.He, Bv = A.Bv; const PqtBinEntry* __restrict__ table = A.table; const uint32_t* __restrict__ lower = A.lower;
  const uint32_t tableBits = A.tableBits; float* __restrict__ qL1virt = A.qL1virt;
  uint32_t* __restrict__ cand = A.cand; uint32_t* __restrict__ candPos = A.candPos; uint32_t* __restrict__ nCand = A.nCand;
  uint32_t* __restrict__ nLocal = A.nLocal; uint32_t* __restrict__ nIncl = A.nIncl; const uint64_t stride = A.stride;
  unsigned long long* __restrict__ counters = A.counters; unsigned long long* __restrict__ tstamp = A.tstamp;
  float* __restrict__ segDOut = A.segDOut; uint32_t* __restrict__ segBOut = A.segBOut;
  uint32_t* __restrict__ ovList = A.ovList; uint32_t* __restrict__ ovCount = A.ovCount;
  const uint32_t* __restrict__ filter = A.filter; const uint32_t filterBits = A.filterBits; const uint32_t tdbg = A.tdbg;
  (void)lower; (void)candPos;
  // compile-time shapes: always the two-phase enumeration (the host selects them only with the packed heuristic rows, the
  // presence bitmap, no modulo hashing and without the order-all-rows debug switch), so the single-pass code is not compiled in
  constexpr bool TWO = SHAPE != 0;
  const uint32_t forceFullOrder = TWO ? 0u : (tdbg & 1u);
#define PQT_TS(i) do { if (tstamp && lane == 0) tstamp[(size_t)q * PQT_TS_WORDS + (i)] = __builtin_readcyclecounter(); } while (0)
  using SH = PqtShape<SHAPE>;
  static_assert(SHAPE == 0 || (P2 && WCR == 1), "compile-time shapes are power-of-two shapes with W*C2 == 64");
  const uint32_t D = SHAPE ? SH::D : prm.D, P = SHAPE ? SH::P : prm.P, C1 = SHAPE ? SH::C1 : prm.C1, C2 = SHAPE ? SH::C2 : prm.C2,
                 W = SHAPE ? SH::W : prm.W, LP = SHAPE ? SH::LP : prm.LP, S = SHAPE ? SH::D / SH::P : prm.S,
                 SS = SHAPE ? SH::D / SH::LP : prm.SS, R = SHAPE ? SH::LP / SH::P : prm.R, WC = SHAPE ? SH::W * SH::C2 : prm.WC;
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t shC1 = P2 ? (uint32_t)__builtin_ctz(C1) : 0u, shC2 = P2 ? (uint32_t)__builtin_ctz(C2) : 0u, shLP = P2 ? (uint32_t)__builtin_ctz(LP) : 0u,
                 shWC = P2 ? (uint32_t)__builtin_ctz(WC) : 0u, shW = P2 ? (uint32_t)__builtin_ctz(W) : 0u, shD = P2 ? (uint32_t)__builtin_ctz(D) : 0u,
                 shS = P2 ? (uint32_t)__builtin_ctz(S) : 0u, shSS = P2 ? (uint32_t)__builtin_ctz(SS) : 0u, shR = P2 ? (uint32_t)__builtin_ctz(R) : 0u;
#define PQT_MUL(x, v, sh) (P2 ? ((x) << (sh)) : ((x) * (v)))
#define PQT_DIV(x, v, sh) (P2 ? ((x) >> (sh)) : ((x) / (v)))
#define PQT_MOD(x, v) (P2 ? ((x) & ((v) - 1u)) : ((x) % (v)))
  // region0 is time-shared: L1virt + unsorted d2 (a1/a2), then the per-row bin records and the compact bin list (a4-a6)
  const uint32_t r0Bytes = perWaveBytes - 4 * (P * C1 + P * W + 2 * P * WC);
  uint64_t* sBin = (uint64_t*)base;                 // 512 : (gcount | lstart<<32) by row, later the compact bin list
  float* sVirt = (float*)base;                      // LP*C1     (dead before sBin is written)
  float* sD2 = sVirt + LP * C1;                     // P*WC unsorted staging (dead before sBin is written)
  float* sQ = sD2 + P * WC;                         // D         (read by a1 and a2 only: dead before sBin is written)
  float* sL1 = (float*)(base + r0Bytes);            // P*C1
  uint32_t* sOrd = (uint32_t*)(sL1 + P * C1);       // P*W
  float* sSegD = (float*)(sOrd + P * W);            // P*WC sorted d2
  uint32_t* sSegB = (uint32_t*)(sSegD + P * WC);    // P*WC sorted bin parts (pre-multiplied)

  PQT_TS(0);
  const uint32_t* __restrict__ heur4 = A.heur4;
  for (uint32_t i = lane; i < D; i += 64) sQ[i] = Q[(size_t)q * D + i];
  __builtin_amdgcn_wave_barrier();
  // ---- a1 (UA accumulators per lane in flight: their cb1 reads overlap; 8 sixteen-byte reads when the shape is known)
  constexpr int UA = SHAPE == 2 ? 8 : 4;
  for (uint32_t t0 = lane; t0 < C1 * LP; t0 += 64 * UA) {
    float acc[UA];
    uint32_t dst[UA];
#pragma unroll
    for (int u = 0; u < UA; ++u) {
      const uint32_t t = t0 + 64 * u;
      constexpr bool kFull = SHAPE != 0;  // compile-time shapes: C1*LP is a multiple of 64*UA, every t is in range
      const uint32_t tt = (kFull || t < C1 * LP) ? t : t0;
      const uint32_t c = PQT_DIV(tt, LP, shLP), lp = PQT_MOD(tt, LP);
      const float* cen = cb1 + (size_t)PQT_MUL(c, D, shD) + PQT_MUL(lp, SS, shSS);
      const float* qq = sQ + PQT_MUL(lp, SS, shSS);
      float s = 0.f;
      if constexpr (SHAPE != 0) {
        // SS = 4 or 8 dims = one or two 16-byte reads, requested for all 4 accumulators before the first is consumed
        constexpr uint32_t V = SH::D / SH::LP / 4;
        float4 cv[V];
#pragma unroll
        for (uint32_t v = 0; v < V; ++v) cv[v] = reinterpret_cast<const float4*>(cen)[v];
#pragma unroll
        for (uint32_t v = 0; v < V; ++v) {
          // differences and squares two dimensions at a time (v_pk_add_f32 / v_pk_mul_f32: IEEE per component, nothing fused);
          // the sum stays sequential in dimension order
          const float4 qv = reinterpret_cast<const float4*>(qq)[v];
          if constexpr (SHAPE == 2) {
            const pqt_f2 d01 = pqt_f2{qv.x, qv.y} - pqt_f2{cv[v].x, cv[v].y}, d23 = pqt_f2{qv.z, qv.w} - pqt_f2{cv[v].z, cv[v].w};
            const pqt_f2 s01 = d01 * d01, s23 = d23 * d23;
            s = s + s01.x; s = s + s01.y; s = s + s23.x; s = s + s23.y;
          } else {  // (SIFT1M shape: the packed form measured 1.1 k clocks slower in this phase)
            float df = qv.x - cv[v].x; s = s + df * df;
            df = qv.y - cv[v].y; s = s + df * df;
            df = qv.z - cv[v].z; s = s + df * df;
            df = qv.w - cv[v].w; s = s + df * df;
          }
        }
      } else {
        for (uint32_t d = 0; d < SS; ++d) { const float df = qq[d] - cen[d]; s = s + df * df; }
      }
      acc[u] = s;
      dst[u] = (kFull || t < C1 * LP) ? PQT_MUL(lp, C1, shC1) + c : 0xffffffffu;
    }
#pragma unroll
    for (int u = 0; u < UA; ++u) if (dst[u] != 0xffffffffu) sVirt[dst[u]] = acc[u];
  }
  __builtin_amdgcn_wave_barrier();
  PQT_TS(1);
  uint32_t ties = 0;
  if constexpr (SHAPE != 0) {
    // compile-time shapes: L1virt leaves in 16-byte pieces; the W nearest cells of a part come from W rounds of a
    // wave-level arg-min over the part's C1 lanes (xor butterfly over (distance key << 32 | cell)) instead of counting
    // every cell's rank against all others (C1 = 64: 1.3 k -> 0.12 k instructions per query).  Same result: keys are
    // unique, the smallest key is the nearest cell, ties go to the lower index like the stable order.
    {
      const float4* src4 = reinterpret_cast<const float4*>(sVirt);
      float4* dst4 = reinterpret_cast<float4*>(qL1virt + (size_t)q * LP * C1);
      for (uint32_t t = lane; t < LP * C1 / 4; t += 64) dst4[t] = src4[t];
    }
    constexpr uint32_t kC1 = SH::C1, kW = SH::W, kR = SH::LP / SH::P;
    static_assert(64 % kC1 == 0 && (SH::P * kC1) % 64 == 0, "a 64-lane chunk holds whole parts");
#pragma unroll
    for (uint32_t t0 = 0; t0 < SH::P * kC1; t0 += 64) {
      const uint32_t t = t0 + lane;
      const uint32_t p = t / kC1, c = t % kC1;
      float d = 0.f;
#pragma unroll
      for (uint32_t pp = 0; pp < kR; ++pp) d = d + sVirt[(p * kR + pp) * kC1 + c];
      sL1[t] = d;
      uint64_t key = ((uint64_t)pqt_f2key(d) << 32) | c;
#pragma unroll
      for (uint32_t w = 0; w < kW; ++w) {
        uint64_t m = key;
        { const uint64_t o = pqt_lane_xor_u64<1>(m); m = o < m ? o : m; }
        { const uint64_t o = pqt_lane_xor_u64<2>(m); m = o < m ? o : m; }
        { const uint64_t o = pqt_lane_xor_u64<4>(m); m = o < m ? o : m; }
        { const uint64_t o = pqt_lane_xor_u64<8>(m); m = o < m ? o : m; }
        { const uint64_t o = pqt_lane_xor_u64<16>(m); m = o < m ? o : m; }
        if constexpr (kC1 == 64) { const uint64_t o = pqt_lane_xor_u64<32>(m); m = o < m ? o : m; }
        // exact ties that touch a selected cell (statistics; every tie that can influence the result is of this kind)
        ties += (key != m && key != ~0ull && (uint32_t)(key >> 32) == (uint32_t)(m >> 32)) ? 1u : 0u;
        if (key == m) { sOrd[p * kW + w] = c; key = ~0ull; }
      }
    }
  } else {
  for (uint32_t t = lane; t < LP * C1; t += 64) qL1virt[(size_t)q * LP * C1 + t] = sVirt[t];
  for (uint32_t t = lane; t < P * C1; t += 64) {
    const uint32_t p = PQT_DIV(t, C1, shC1), c = PQT_MOD(t, C1);
    float d = 0.f;
    for (uint32_t pp = 0; pp < R; ++pp) d = d + sVirt[PQT_MUL(PQT_MUL(p, R, shR) + pp, C1, shC1) + c];
    sL1[t] = d;
  }
  __builtin_amdgcn_wave_barrier();
  if (C1 > 64 && C1 <= 256 && W <= 4) {
    // many first-level cells, few of them expanded (BASELINE configs[4]: C1 = 128, W = 1): W rounds of a wave-wide arg-min per part over
    // (distance key << 32 | cell) -- ties go to the lower cell like the stable order -- instead of every cell's rank against all others
    // (C1 = 128: ~10 k instructions per query, 41 % of a traversal at that shape: profiles/r04_cfg5_phase_clocks.txt).  The tie
    // statistic counts the exact ties that touch a selected cell (every tie that can influence the result is of this kind).
    for (uint32_t p = 0; p < P; ++p) {
      uint64_t kc[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) { const uint32_t c = lane + 64u * r; kc[r] = c < C1 ? (((uint64_t)pqt_f2key(sL1[PQT_MUL(p, C1, shC1) + c]) << 32) | c) : ~0ull; }
      for (uint32_t w = 0; w < W; ++w) {
        uint64_t m = kc[0];
#pragma unroll
        for (int r = 1; r < 4; ++r) m = kc[r] < m ? kc[r] : m;
        { const uint64_t o = pqt_lane_xor_u64<1>(m); m = o < m ? o : m; }
        { const uint64_t o = pqt_lane_xor_u64<2>(m); m = o < m ? o : m; }
        { const uint64_t o = pqt_lane_xor_u64<4>(m); m = o < m ? o : m; }
        { const uint64_t o = pqt_lane_xor_u64<8>(m); m = o < m ? o : m; }
        { const uint64_t o = pqt_lane_xor_u64<16>(m); m = o < m ? o : m; }
        { const uint64_t o = pqt_lane_xor_u64<32>(m); m = o < m ? o : m; }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          ties += (kc[r] != m && kc[r] != ~0ull && (uint32_t)(kc[r] >> 32) == (uint32_t)(m >> 32)) ? 1u : 0u;
          if (kc[r] == m) { sOrd[PQT_MUL(p, W, shW) + w] = (uint32_t)m; kc[r] = ~0ull; }
        }
      }
    }
  } else
  for (uint32_t t = lane; t < P * C1; t += 64) {
    const uint32_t p = PQT_DIV(t, C1, shC1), c = PQT_MOD(t, C1);
    const float my = sL1[t];
    uint32_t rank = 0;
    if ((C1 & 3) == 0 && (((uint32_t)(uintptr_t)(sL1 + PQT_MUL(p, C1, shC1))) & 15u) == 0) {  // 16-byte LDS reads: 4 cells per ds_read_b128
      const float4* row4 = reinterpret_cast<const float4*>(sL1 + PQT_MUL(p, C1, shC1));
      for (uint32_t o4 = 0; o4 < C1 / 4; ++o4) {
        const float4 v4 = row4[o4];
        const float vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const uint32_t o = o4 * 4 + e;
          rank += (vv[e] < my) || (vv[e] == my && o < c);
          ties += (vv[e] == my && o < c);
        }
      }
    } else {
      for (uint32_t o = 0; o < C1; ++o) {
        const float v = sL1[p * C1 + o];
        rank += (v < my) || (v == my && o < c);
        ties += (v == my && o < c);
      }
    }
    if (rank < W) sOrd[p * W + rank] = c;
  }
  }
  pqt_count_ties(&counters[0], ties);
  __builtin_amdgcn_wave_barrier();
  PQT_TS(2);
  // ---- a2 (2 entries per lane in flight)
  for (uint32_t t0 = lane; t0 < P * WC; t0 += 64 * 2) {
    float acc[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const uint32_t t = t0 + 64 * u;
      const uint32_t tt = (SHAPE != 0 || t < P * WC) ? t : t0;  // compile-time shapes: P*WC = 256, every t is in range
      const uint32_t p = PQT_DIV(tt, WC, shWC), pos = PQT_MOD(tt, WC), h1 = PQT_DIV(pos, C2, shC2), h2 = PQT_MOD(pos, C2);
      const uint32_t c1 = sOrd[PQT_MUL(p, W, shW) + h1];
      const float* qq = sQ + PQT_MUL(p, S, shS);
      float s = 0.f;
      if (cb2T) {
        // transposed tile: vector v of the C2 rows of a cell is contiguous, so the 32..64 lanes walking the rows of
        // a cell read 512..1024 contiguous bytes per instruction (row-per-lane reads of the file layout touch one
        // cache line per lane); same dims in the same order
        const float4* cen4 = cb2T + PQT_MUL(PQT_MUL((size_t)(PQT_MUL(p, C1, shC1) + c1), S / 4, shS - 2), C2, shC2) + h2;
        // 8 vectors of the row are requested together (one round trip per 32 dims instead of one per 4), then summed in
        // dimension order
        for (uint32_t v0 = 0; v0 < S / 4; v0 += 8) {
          float4 c[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const uint32_t v = v0 + e;
            c[e] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (v < S / 4) c[e] = cen4[PQT_MUL((size_t)v, C2, shC2)];
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const uint32_t v = v0 + e;
            if (v < S / 4) {
              const pqt_f2 d01 = pqt_f2{qq[4 * v], qq[4 * v + 1]} - pqt_f2{c[e].x, c[e].y}, d23 = pqt_f2{qq[4 * v + 2], qq[4 * v + 3]} - pqt_f2{c[e].z, c[e].w};
              const pqt_f2 s01 = d01 * d01, s23 = d23 * d23;  // packed, per-component IEEE; the sum stays in dimension order
              s = s + s01.x; s = s + s01.y; s = s + s23.x; s = s + s23.y;
            }
          }
        }
      } else {
        const float* cen = cb2 + PQT_MUL(PQT_MUL((size_t)(PQT_MUL(p, C1, shC1) + c1), C2, shC2) + h2, S, shS);
        for (uint32_t d = 0; d < S; ++d) { const float df = qq[d] - cen[d]; s = s + df * df; }
      }
      acc[u] = s;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) if (SHAPE != 0 || t0 + 64 * u < P * WC) sD2[t0 + 64 * u] = acc[u];
  }
  __builtin_amdgcn_wave_barrier();
  PQT_TS(3);
  ties = 0;
  bool rowSorted = false;
#ifndef PQT_NO_ROW_SORT
  if constexpr (SHAPE != 0) {
    // compile-time shapes (P = 4 lists of W*C2 = 64 entries): the four part lists are sorted TOGETHER, list p by the 16-lane row p of the
    // wavefront with 4 entries per lane (pqt_row_sort64_u32: 178 instructions for all four, the one-entry-per-lane network below takes 161
    // per list), on the same 32-bit keys -- distance key with its low 6 bits replaced by the entry's position.  That order is the exact
    // (distance, position) order unless two NEIGHBOURS of a sorted list agree in the upper 26 bits of their distance keys (two second-level
    // distances within 2^-17 of each other, relatively); a query where that happens anywhere takes the path below, which settles such
    // pairs exactly.  Lists enter and leave in 16-byte LDS accesses: entry e of list p sits at word 64 p + e = 4 lane + r.
    static_assert(SHAPE == 0 || (PqtShape<SHAPE>::P == 4 && PqtShape<SHAPE>::W * PqtShape<SHAPE>::C2 == 64), "row sort: 4 lists of 64");
    const uint32_t pr = lane >> 4, l16 = lane & 15u;
    const float4 dv = reinterpret_cast<const float4*>(sD2)[lane];
    uint32_t k[4] = {(pqt_f2key(dv.x) & ~63u) | (4u * l16), (pqt_f2key(dv.y) & ~63u) | (4u * l16 + 1u),
                     (pqt_f2key(dv.z) & ~63u) | (4u * l16 + 2u), (pqt_f2key(dv.w) & ~63u) | (4u * l16 + 3u)};
    pqt_row_sort64_u32(k);
    // neighbours with equal upper 26 bits (the successor of a lane's last entry is the next lane's first; the last lane of a row has none)
    const uint32_t nx0 = pqt_lane_down1_u32(k[0]);
    const bool close = ((k[0] ^ k[1]) < 64u) || ((k[1] ^ k[2]) < 64u) || ((k[2] ^ k[3]) < 64u) || (l16 != 15u && (k[3] ^ nx0) < 64u);
    if (__builtin_expect(__ballot(close) == 0ull && !(tdbg & 2u), 1)) {
      rowSorted = true;
      const float* dRow = sD2 + 64u * pr;
      const uint32_t pos[4] = {k[0] & 63u, k[1] & 63u, k[2] & 63u, k[3] & 63u};
      const float4 dOut = make_float4(dRow[pos[0]], dRow[pos[1]], dRow[pos[2]], dRow[pos[3]]);
      // bin part of an entry = (cell * C2 + second-level index) * (C1 C2)^p in uint32 wrap-around arithmetic; (C1 C2)^p is a power of two
      // here, or 0 once p * log2(C1 C2) reaches 32
      constexpr uint32_t shCC = (uint32_t)__builtin_ctz(SH::C1 * SH::C2);
      const uint32_t sh = pr * shCC, shMask = sh < 32u ? 0xffffffffu : 0u;
      uint32_t cellC2[2];
      cellC2[0] = sOrd[pr * SH::W] * SH::C2;
      cellC2[1] = SH::W > 1 ? sOrd[pr * SH::W + (SH::W > 1 ? 1u : 0u)] * SH::C2 : 0u;
      uint32_t part[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) part[r] = (SH::W > 1 && (pos[r] >> shC2) ? cellC2[1] : cellC2[0]) + (pos[r] & (SH::C2 - 1u));
      reinterpret_cast<float4*>(sSegD)[lane] = dOut;
      reinterpret_cast<uint4*>(sSegB)[lane] = make_uint4((part[0] << (sh & 31u)) & shMask, (part[1] << (sh & 31u)) & shMask,
                                                         (part[2] << (sh & 31u)) & shMask, (part[3] << (sh & 31u)) & shMask);
      if (He > 512) {  // wide enumeration: keep the lists for the overflow hand-over to pqt_k_bins
        reinterpret_cast<float4*>(segDOut + (size_t)q * (SH::P * 64u))[lane] = dOut;
        reinterpret_cast<uint4*>(segBOut + (size_t)q * (SH::P * 64u))[lane] = make_uint4(part[0], part[1], part[2], part[3]);
      }
    }
  }
#endif
  constexpr int PM = (4 / WCR) < 1 ? 1 : (4 / WCR);  // parts sorted together (register budget: PM * WCR keys)
  if (!rowSorted)
  for (uint32_t p0 = 0; p0 < P; p0 += PM) {
    // several parts at once: their sorting networks are independent dependency chains the scheduler interleaves
    uint64_t key[PM][WCR];
#pragma unroll
    for (int m = 0; m < PM; ++m) {
      const uint32_t p = p0 + m < P ? p0 + m : p0;
#pragma unroll
      for (int r = 0; r < WCR; ++r) {
        const uint32_t pos = lane + 64 * r;
        key[m][r] = pos < WC ? (((uint64_t)pqt_f2key(sD2[PQT_MUL(p, WC, shWC) + pos]) << 32) | pos) : ~0ull;
      }
    }
#ifndef PQT_NO_SORT32
    if constexpr (WCR == 1) {
      // 64 entries per part, one per lane: the network runs on 32-bit keys -- the distance key with its low 6 bits replaced by the entry's
      // position (unique, so min / max compare-exchanges do) -- a third of the instructions of the (distance key, position) u64 network.
      // That order equals the exact one unless two neighbours of the result agree in the upper 26 bits of their distance keys and stand
      // in the wrong (distance, position) order; every lane checks its successor, and a part where that happens is sorted again by the
      // u64 network (two second-level distances within 2^-17 of each other, relatively: rare on any data, certain for exact duplicates).
      uint32_t k32[PM];
#pragma unroll
      for (int m = 0; m < PM; ++m) k32[m] = ((uint32_t)(key[m][0] >> 32) & ~63u) | ((uint32_t)key[m][0] & 63u);
#pragma unroll
      for (int m = 0; m < PM; ++m) { uint32_t t[1] = {k32[m]}; pqt_wave_sort_u32<1>(t); k32[m] = t[0]; }
      unsigned long long bad = 0;
#pragma unroll
      for (int m = 0; m < PM; ++m) {
        const uint32_t p = p0 + m < P ? p0 + m : p0;
        const uint32_t pos = k32[m] & 63u;
        const uint64_t full = ((uint64_t)pqt_f2key(sD2[PQT_MUL(p, WC, shWC) + pos]) << 32) | pos;  // exact key of the entry now at rank `lane`
        const uint64_t nxt = pqt_lane_down1_u64(full);                                               // ... of the entry at rank lane + 1
        if (__ballot(lane < 63 && pos < WC && (uint32_t)nxt < WC && nxt < full)) bad |= 1ull << m;
        key[m][0] = (uint32_t)full < WC ? full : ~0ull;
      }
      if (bad) {  // uniform, rare
#pragma unroll
        for (int m = 0; m < PM; ++m) {
          if ((bad >> m) & 1ull) {
            const uint32_t p = p0 + m < P ? p0 + m : p0;
            key[m][0] = lane < WC ? (((uint64_t)pqt_f2key(sD2[PQT_MUL(p, WC, shWC) + lane]) << 32) | lane) : ~0ull;
            pqt_wave_sort_u64<1>(key[m]);
          }
        }
      }
    } else
#endif
#pragma unroll
    for (int m = 0; m < PM; ++m) pqt_wave_sort_u64<WCR>(key[m]);
#pragma unroll
    for (int m = 0; m < PM; ++m) {
      const uint32_t p = p0 + m;
      if (p < P) {
#pragma unroll
        for (int r = 0; r < WCR; ++r) {
          const uint32_t e = lane * WCR + r;
          if (e < WC) {
            const uint32_t pos = (uint32_t)key[m][r];
            sSegD[PQT_MUL(p, WC, shWC) + e] = sD2[PQT_MUL(p, WC, shWC) + pos];
            sSegB[PQT_MUL(p, WC, shWC) + e] = (PQT_MUL(sOrd[PQT_MUL(p, W, shW) + PQT_DIV(pos, C2, shC2)], C2, shC2) + PQT_MOD(pos, C2)) * prm.powers[p];  // pre-multiplied by (C1*C2)^p, uint32 wrap
            if (He > 512) {  // wide enumeration: keep the lists for the overflow hand-over to pqt_k_bins
              segDOut[((size_t)q * P + p) * WC + e] = sD2[p * WC + pos];
              segBOut[((size_t)q * P + p) * WC + e] = PQT_MUL(sOrd[PQT_MUL(p, W, shW) + PQT_DIV(pos, C2, shC2)], C2, shC2) + PQT_MOD(pos, C2);
            }
          }
          // exact ties between neighbours of the sorted list (statistics only)
          const uint32_t hi = (uint32_t)(key[m][r] >> 32);
          const uint32_t nx = (r + 1 < WCR) ? (uint32_t)(key[m][(r + 1) % WCR] >> 32) : __shfl_down((uint32_t)(key[m][0] >> 32), 1, 64);
          if (e + 1 < WC && hi == nx && !(r + 1 == WCR && lane == 63)) ++ties;
        }
      }
    }
  }
  pqt_count_ties(&counters[1], ties);
  __builtin_amdgcn_wave_barrier();
  PQT_TS(4);
  // ---- a4 + a5: 8 rows per lane and block of 512 rows, row h = hb + lane + 64*r
  uint64_t key[8];
  uint32_t recG[8], recL[8];  // population of the row's bin (0: empty), start of its members (sharded: table slot)
  auto rowKey = [&](const uint32_t h, const uint32_t (&dg)[8], uint32_t& globOut) -> uint64_t {
    float fine = 0.f;
    uint32_t g = 0;
#pragma unroll
    for (int p = 0; p < PQT_MAXP; ++p) {
      if ((uint32_t)p < P) {
        fine = fine + sSegD[PQT_MUL((uint32_t)p, WC, shWC) + dg[p]];
        g += sSegB[PQT_MUL((uint32_t)p, WC, shWC) + dg[p]];
      }
    }
    if (prm.hashMod) g %= prm.hashMod;
    globOut = g;
    return ((uint64_t)pqt_f2key(fine) << 32) | h;
  };
  uint32_t hw[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  auto rowBlock = [&](const uint32_t hb) {
    // packed table (one dword per row): the rows of the NEXT block of a wide enumeration are requested while this one is
    // processed (8 VGPRs; wide traversal 0.272 -> 0.247 ms).  Requesting the first block at kernel start was measured as
    // well: it costs the short enumeration 6 % (8 more registers live through a1/a2), so the first block is read here.
    uint32_t glob[8];
    uint32_t cur[8];
    if (heur4 && hb == 0) {
#pragma unroll
      for (int r = 0; r < 8; ++r) { const uint32_t h = lane + 64 * r; hw[r] = h < He ? heur4[h] : 0u; }
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) cur[r] = hw[r];
    if (heur4 && hb + 512 < He) {  // next block's rows, consumed one iteration later
#pragma unroll
      for (int r = 0; r < 8; ++r) { const uint32_t h = hb + 512 + lane + 64 * r; hw[r] = h < He ? heur4[h] : 0u; }
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const uint32_t h = hb + lane + 64 * r;
      key[r] = ~0ull;
      glob[r] = 0;
      if (h < He) {
        if (heur4) {
          const uint32_t dg[8] = {cur[r] & 0xffu, (cur[r] >> 8) & 0xffu, (cur[r] >> 16) & 0xffu, cur[r] >> 24, 0u, 0u, 0u, 0u};
          key[r] = rowKey(h, dg, glob[r]);
        } else {
          // (requesting the 16-byte rows ahead was measured: 32 more live VGPRs spill at 5 waves per SIMD and a2 slows
          // down by more than the round trip saved)
          const uint4 hv = heur8[h];  // one 16-byte read: the row's P digits
          const uint32_t dg[8] = {hv.x & 0xffffu, hv.x >> 16, hv.y & 0xffffu, hv.y >> 16, hv.z & 0xffffu, hv.z >> 16, hv.w & 0xffffu, hv.w >> 16};
          key[r] = rowKey(h, dg, glob[r]);
        }
      }
    }
    // probes: first touch of all 8 slots is issued before any is consumed
    const uint4* table4 = reinterpret_cast<const uint4*>(table);  // {key, gcount, lstart, lcount}
    // 16 independent 16-byte reads per lane in flight, then resolved: exactly one memory round trip
    // almost all enumerated rows name empty bins: one 4-byte read of the presence bitmap (a few hundred KB, L2 resident)
    // settles those, only the rows whose bit is set go on to the two 16-byte table probes (measured: -10 % on the
    // traversal at 500 rows per query, -27 % at 4096 where the probe rate is the bound)
    uint32_t maybe = 0xffu;
    if (filter) {
      uint32_t fw[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) fw[r] = filter[pqt_hash_filter(glob[r], filterBits) >> 5];
      maybe = 0;
#pragma unroll
      for (int r = 0; r < 8; ++r) maybe |= ((fw[r] >> (pqt_hash_filter(glob[r], filterBits) & 31u)) & 1u) << r;
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const uint32_t h = hb + lane + 64 * r;
      uint32_t slot = 0;
      uint4 x = make_uint4(0, 0, 0, 0);
      if ((maybe >> r) & 1u) x = pqt_table_lookup(table4, glob[r], tableBits, prm.tableSeed, &slot);
      recG[r] = h < He ? x.y : 0u;
      recL[r] = SHARDED ? slot : x.z;  // sharded: keep the slot, resolve the local fields after the cut
    }
  };
  // phase 2 of the two-phase enumeration: the work list (row | bin id << 32, nwork <= 512 entries, entry e = lane + 64 r) is
  // probed in one round trip -- blocks of 64 entries beyond the list are skipped by a uniform branch -- and the rows whose
  // bin exists get their distance key.  Leaves recG / recL / key in work-list arrangement; the list's LDS area is free after.
  auto probeWork = [&](const uint64_t* sWork, const uint32_t nwork) {
    const uint4* table4 = reinterpret_cast<const uint4*>(table);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      recG[r] = 0; recL[r] = 0; key[r] = ~0ull;
      if (64u * r < nwork) {
        const uint32_t e = lane + 64 * r;
        const uint64_t we = e < nwork ? sWork[e] : 0ull;
        const uint32_t hrow = (uint32_t)we;
        uint32_t slot = 0;
        uint4 x = make_uint4(0, 0, 0, 0);
        uint32_t wd = 0;
        if (e < nwork) { x = pqt_table_lookup(table4, (uint32_t)(we >> 32), tableBits, prm.tableSeed, &slot); wd = heur4[hrow]; }
        recG[r] = x.y;
        recL[r] = SHARDED ? slot : x.z;
        if (x.y) {
          float fine = 0.f;
#pragma unroll
          for (int p = 0; p < 4; ++p) if ((uint32_t)p < P) fine = fine + sSegD[PQT_MUL((uint32_t)p, WC, shWC) + ((wd >> (8 * p)) & 0xffu)];
          key[r] = ((uint64_t)pqt_f2key(fine) << 32) | hrow;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();  // the work list is in registers now: its LDS area can be overwritten
  };
  // Short enumeration (He <= 512) in the same two phases: every row's bin id + one bitmap word, the few "maybe" rows go to
  // a work list that aliases sBin, one round of probes for the list.  (The single-pass rowBlock computed a distance key
  // for every row and ran all eight predicated probe blocks; this kernel is issue-bound at 5 waves per SIMD.)
  const bool twoShort = TWO && He <= 512;
  if (twoShort) {
    uint64_t* sWork = sBin;
    uint32_t g[8], f[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) { const uint32_t h = lane + 64 * r; hw[r] = h < He ? heur4[h] : 0u; }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      uint32_t gg = 0;
#pragma unroll
      for (int p = 0; p < 4; ++p) if ((uint32_t)p < P) gg += sSegB[PQT_MUL((uint32_t)p, WC, shWC) + ((hw[r] >> (8 * p)) & 0xffu)];
      g[r] = gg;
      f[r] = filter[pqt_hash_filter(gg, filterBits) >> 5];
    }
    uint32_t nwork = 0;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const uint32_t h = lane + 64 * r;
      const bool bit = h < He && ((f[r] >> (pqt_hash_filter(g[r], filterBits) & 31u)) & 1u);
      uint32_t tot;
      const uint32_t rk = pqt_ballot_rank(bit, &tot);
      if (bit) sWork[nwork + rk] = (uint64_t)h | ((uint64_t)g[r] << 32);
      nwork += tot;
    }
    __builtin_amdgcn_wave_barrier();
    probeWork(sWork, nwork);
  } else if (!TWO && He <= 512) rowBlock(0);
  PQT_TS(5);

  // ---- a6, shared by the two orderings below.  skey: the sorted keys, element i = lane*R + r, low word & 0xffff = index
  // of the bin's record in sBin; cnt elements.  Scans the populations in visiting order, applies the cut, writes the
  // compact list of the included populated bins to LDS and gathers the candidates by binary search over it.
  // Returns the number of included populated bins and (by reference) the candidate total.
  // rerank schedule 2: position in the (pool, size class) list -- the atomic is issued as soon as the count is known and its
  // result is used after the candidate list is written (the round trip hides under the expansion)
  uint32_t schedSlot = 0, schedPos = 0, schedN = 0;
  auto schedDraw = [&](const uint32_t nloc) {
    schedN = nloc;
    if (A.schedCnt && lane == 0) {
      schedSlot = (q & 7u) * PQT_SCHED_CLASSES + pqt_sched_class(nloc);
      schedPos = __hip_atomic_fetch_add(&A.schedCnt[schedSlot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  auto schedCommit = [&]() {
    if (A.schedCnt && lane == 0 && schedPos < A.schedCap) A.schedList[(size_t)schedSlot * A.schedCap + schedPos] = (unsigned long long)q | ((unsigned long long)schedN << 32);
  };
  auto finish = [&](auto& skey, const uint32_t cnt, uint32_t& totCandOut) -> uint32_t {
    constexpr int R = (int)(sizeof(skey) / sizeof(skey[0]));
    uint32_t g8[R], ls8[R];
    uint32_t sum = 0;
    uint32_t tb = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const uint32_t i = lane * R + r;
      g8[r] = 0; ls8[r] = 0;
      if (i < cnt) {
        const uint64_t b = sBin[(uint32_t)skey[r] & 0xffffu];
        g8[r] = (uint32_t)b; ls8[r] = (uint32_t)(b >> 32);
        const uint32_t hi = (uint32_t)(skey[r] >> 32);
        const uint32_t nx = (r + 1 < R) ? (uint32_t)(skey[(r + 1) % R] >> 32) : __shfl_down((uint32_t)(skey[0] >> 32), 1, 64);
        if (i + 1 < cnt && hi == nx && !(r + 1 == R && lane == 63)) ++tb;
      }
      sum += g8[r];
    }
    pqt_count_ties(&counters[2], tb);
    const uint32_t incl = pqt_wave_incl_scan(sum);
    uint32_t run = incl - sum;  // exclusive prefix of this lane's first element
    uint32_t myCand = 0, myNonEmpty = 0;
    uint32_t ex8[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const uint32_t i = lane * R + r;
      const uint32_t g = g8[r];  // 0 beyond cnt
      ex8[r] = run;
      if (i < cnt && run <= Bv) { myCand += g; if (g) ++myNonEmpty; } else { g8[r] = 0; }
      run += g;
    }
    __builtin_amdgcn_wave_barrier();
    // wave totals
    uint32_t totCand = myCand, totNe = myNonEmpty;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { totCand += __shfl_xor(totCand, d, 64); totNe += __shfl_xor(totNe, d, 64); }
    totCandOut = totCand;
    if constexpr (!SHARDED) {
      const uint32_t neIncl = pqt_wave_incl_scan(myNonEmpty);
      const uint32_t m = __shfl(neIncl, 63, 64);  // non-empty included bins
      uint32_t wpos = neIncl - myNonEmpty;
      // compact list (start in candidate list | lstart<<32), ordered by start; overwrites sBin (all reads done)
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (g8[r]) { sBin[wpos] = (uint64_t)ex8[r] | ((uint64_t)ls8[r] << 32); ++wpos; }
      }
      __builtin_amdgcn_wave_barrier();
      if (lane == 0) { nCand[q] = totCand; nLocal[q] = totCand; if (A.outCount) A.outCount[q] = totCand; }
      schedDraw(totCand);
      PQT_TS(7);
      if (A.runs) {
        // hand the compact list itself to the rerank when it fits: no candidate list is written
        if (m <= A.runCap) {
          for (uint32_t i = lane; i < m; i += 64) A.runs[(size_t)q * PQT_RUNCAP + i] = sBin[i];
          if (lane == 0) A.nRuns[q] = m;
          schedCommit();
          return totNe;
        }
        if (lane == 0) A.nRuns[q] = 0xffffffffu;
      }
      uint32_t* const out = cand + (size_t)q * stride;
      if (totCand >= 32u * m) {
        // long bins (BASELINE configs[2]/[3]: hundreds of members each): the wave walks the listed bins and writes each
        // one's consecutive store positions with coalesced stores -- no search (the per-candidate binary search below was
        // 86 k of the 228 k clocks of a cfg3-shape traversal)
        for (uint32_t b0 = 0; b0 < m; b0 += 64) {
          const uint64_t mine = (b0 + lane < m) ? sBin[b0 + lane] : 0ull;
          const uint32_t st = (uint32_t)mine, ls = (uint32_t)(mine >> 32);
          const uint32_t nx = __shfl_down(st, 1, 64);
          const uint32_t en = (b0 + lane + 1 < m) ? (lane < 63 ? nx : (uint32_t)sBin[b0 + 64]) : totCand;
          const uint32_t nb = (m - b0 < 64u) ? m - b0 : 64u;
          for (uint32_t b = 0; b < nb; ++b) {
            const uint32_t s0 = (uint32_t)__builtin_amdgcn_readlane((int)st, (int)b), l0 = (uint32_t)__builtin_amdgcn_readlane((int)ls, (int)b),
                           e0 = (uint32_t)__builtin_amdgcn_readlane((int)en, (int)b);
            for (uint32_t j = s0 + lane; j < e0; j += 256) {  // 4 coalesced stores per trip
              out[j] = l0 + (j - s0);
              if (j + 64 < e0) out[j + 64] = l0 + (j + 64 - s0);
              if (j + 128 < e0) out[j + 128] = l0 + (j + 128 - s0);
              if (j + 192 < e0) out[j + 192] = l0 + (j + 192 - s0);
            }
          }
        }
      } else
      for (uint32_t j = lane; j < totCand; j += 64) {
        uint32_t lo = 0, hi = m;  // last entry with start <= j
        while (hi - lo > 1) {
          const uint32_t mid = (lo + hi) >> 1;
          if ((uint32_t)sBin[mid] <= j) lo = mid; else hi = mid;
        }
        const uint64_t b = sBin[lo];
        out[j] = (uint32_t)(b >> 32) + (j - (uint32_t)b);  // position in the bin-ordered line store
      }
    } else {
      // range shard: the cut above used the GLOBAL populations; now resolve what this device holds of the included bins
      // (one more round trip, only for included populated bins) and build the LOCAL candidate list together with each
      // candidate's global visiting position = global start of its bin + members held by lower shards + offset.
      const uint4* table4 = reinterpret_cast<const uint4*>(table);
      if (A.gbins) {
        // query-sharded traversal: the included populated bins as (bin id, global start) in visiting order; every shard resolves
        // its own members from its own table (pqt_k_tables_resolve)
        uint32_t kv[R];
#pragma unroll
        for (int r = 0; r < R; ++r) kv[r] = g8[r] ? table4[ls8[r]].x : 0u;
        const uint32_t neIncl = pqt_wave_incl_scan(myNonEmpty);
        const uint32_t m = __shfl(neIncl, 63, 64);
        uint32_t wpos = neIncl - myNonEmpty;
        unsigned long long* const row = A.gbins + (size_t)q * (A.gbinCap + 1u);
        if (m <= A.gbinCap) {
#pragma unroll
          for (int r = 0; r < R; ++r) if (g8[r]) { row[wpos] = (unsigned long long)kv[r] | ((unsigned long long)ex8[r] << 32); ++wpos; }
        }
        if (lane == 0) {
          row[A.gbinCap] = (unsigned long long)(m <= A.gbinCap ? m : 0xffffffffu) | ((unsigned long long)totCand << 32);
          nCand[q] = totCand; nLocal[q] = 0; if (A.outCount) A.outCount[q] = totCand;
        }
        return totNe;
      }
      uint32_t lc8[R], gp8[R];
      uint32_t myLocal = 0, myLocalBins = 0;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        lc8[r] = 0; gp8[r] = 0;
        if (g8[r]) {
          const uint32_t slot = ls8[r];
          const uint4 e = table4[slot];
          ls8[r] = e.z; lc8[r] = e.w; gp8[r] = ex8[r] + lower[slot];
          myLocal += e.w;
          if (e.w) ++myLocalBins;
        }
      }
      const uint32_t locIncl = pqt_wave_incl_scan(myLocal);
      const uint32_t totLocal = __shfl(locIncl, 63, 64);
      uint32_t lrun = locIncl - myLocal;
      const uint32_t nbIncl = pqt_wave_incl_scan(myLocalBins);
      const uint32_t m = __shfl(nbIncl, 63, 64);
      uint32_t wpos = nbIncl - myLocalBins;
      uint32_t* sGpos = (uint32_t*)(sBin + 512);  // 512 words: global position of the first local member of a listed bin
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (lc8[r]) { sBin[wpos] = (uint64_t)lrun | ((uint64_t)ls8[r] << 32); sGpos[wpos] = gp8[r]; ++wpos; lrun += lc8[r]; }
      }
      __builtin_amdgcn_wave_barrier();
      if (lane == 0) { nCand[q] = totCand; nLocal[q] = totLocal; if (A.outCount) A.outCount[q] = totCand; }
      schedDraw(totLocal);
      PQT_TS(7);
      if (A.runs) {
        if (m <= A.runCap) {
          for (uint32_t i = lane; i < m; i += 64) { A.runs[(size_t)q * PQT_RUNCAP + i] = sBin[i]; A.runGpos[(size_t)q * PQT_RUNCAP + i] = sGpos[i]; }
          if (lane == 0) A.nRuns[q] = m;
          schedCommit();
          return totNe;
        }
        if (lane == 0) A.nRuns[q] = 0xffffffffu;
      }
      uint32_t* const out = cand + (size_t)q * stride;
      uint32_t* const outP = candPos + (size_t)q * stride;
      if (totLocal >= 32u * m) {  // long bins: walk the listed bins, coalesced stores, no search (see the unsharded branch)
        for (uint32_t b0 = 0; b0 < m; b0 += 64) {
          const uint64_t mine = (b0 + lane < m) ? sBin[b0 + lane] : 0ull;
          const uint32_t st = (uint32_t)mine, ls = (uint32_t)(mine >> 32), gp = (b0 + lane < m) ? sGpos[b0 + lane] : 0u;
          const uint32_t nx = __shfl_down(st, 1, 64);
          const uint32_t en = (b0 + lane + 1 < m) ? (lane < 63 ? nx : (uint32_t)sBin[b0 + 64]) : totLocal;
          const uint32_t nb = (m - b0 < 64u) ? m - b0 : 64u;
          for (uint32_t b = 0; b < nb; ++b) {
            const uint32_t s0 = (uint32_t)__builtin_amdgcn_readlane((int)st, (int)b), l0 = (uint32_t)__builtin_amdgcn_readlane((int)ls, (int)b),
                           e0 = (uint32_t)__builtin_amdgcn_readlane((int)en, (int)b), g0 = (uint32_t)__builtin_amdgcn_readlane((int)gp, (int)b);
            for (uint32_t j = s0 + lane; j < e0; j += 64) { out[j] = l0 + (j - s0); outP[j] = g0 + (j - s0); }
          }
        }
      } else
      for (uint32_t j = lane; j < totLocal; j += 64) {
        uint32_t lo = 0, hi = m;
        while (hi - lo > 1) {
          const uint32_t mid = (lo + hi) >> 1;
          if ((uint32_t)sBin[mid] <= j) lo = mid; else hi = mid;
        }
        const uint64_t b = sBin[lo];
        const uint32_t off = j - (uint32_t)b;
        out[j] = (uint32_t)(b >> 32) + off;
        outP[j] = sGpos[lo] + off;
      }
    }
    schedCommit();
    return totNe;
  };

  // Only the POPULATED rows have to be ordered: empty bins add nothing to the running count, so their place in the
  // visiting order changes neither the cut nor a candidate's position.  Of the He rows a query enumerates, a few dozen
  // are populated (the index holds far fewer bins than the multi-index has cells), and the sorting network is where
  // this kernel spends its VALU instructions: 64 or 128 keys cost 7 % / 19 % of the 512-key network.
  // The populated rows are compacted in row order (the tie-break), sorted, and finished as above; more than 128 of
  // them (rare) fall back to ordering all rows.
  uint32_t npop = 0;
  uint32_t totCand = 0, totIncl = 0;  // candidates, included populated bins
  if (He > 512) {
    // Wide enumeration (512 < He <= 4096 rows): the rows go by in blocks of 512, their populated ones are appended to a
    // 512-entry list (records + keys, 8 KB of LDS), which is then ordered and finished like the short list below.
    // A query with more than 512 populated rows is handed to pqt_k_bins (workgroup per query, He-sized arena).
    uint64_t* sKeyW = sBin + 512;
    if (TWO || (heur4 && filter && !prm.hashMod)) {
      // Two phases (packed table + presence bitmap).  Block by block the old loop chained three dependent round trips (rows ->
      // bitmap -> table probes) eight times over.  Phase 1 only decides which rows MAY name an existing bin: bin id from the
      // pre-multiplied part lists (4 LDS reads), one bitmap word -- with the next block's bin ids and bitmap reads issued
      // before the current block's words are consumed -- and the few "maybe" rows (the populated ones plus ~1.6 % false
      // positives) are appended to a work list (row | bin id << 32) in LDS.  Phase 2 probes the whole list at once (one round
      // trip, 8 entries per lane), computes the distance key of the rows that exist and compacts them as before.
      uint64_t* sWork = sKeyW;  // 512 entries; read back into registers before the final lists are written
      uint32_t nwork = 0;
      uint32_t gA[8], fA[8];
      auto stageA = [&](const uint32_t hb, uint32_t (&g)[8], uint32_t (&f)[8]) {
        if (hb == 0) {
#pragma unroll
          for (int r = 0; r < 8; ++r) { const uint32_t h = lane + 64 * r; hw[r] = h < He ? heur4[h] : 0u; }
        }
        uint32_t w[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) w[r] = hw[r];
        if (hb + 512 < He) {
#pragma unroll
          for (int r = 0; r < 8; ++r) { const uint32_t h = hb + 512 + lane + 64 * r; hw[r] = h < He ? heur4[h] : 0u; }
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          uint32_t gg = 0;
#pragma unroll
          for (int p = 0; p < 4; ++p) if ((uint32_t)p < P) gg += sSegB[PQT_MUL((uint32_t)p, WC, shWC) + ((w[r] >> (8 * p)) & 0xffu)];
          g[r] = gg;
          const uint32_t hbit = pqt_hash_filter(gg, filterBits);
          if (sF1) {
            // first level in LDS: 4096 rows per query name ~1 % existing bins, and the word of the 512 KB bitmap behind every row was what
            // bounded this mode (~190 G random 4-byte reads per second through the texture path); a clear first-level bit proves the word's
            // bit clear, so only the rows that pass (the set fraction of the folded bitmap) still ask for it
            const uint32_t i1 = hbit >> (filterBits - A.filter1Bits);
            f[r] = ((sF1[i1 >> 5] >> (i1 & 31u)) & 1u) ? filter[hbit >> 5] : 0u;
          } else
          f[r] = filter[hbit >> 5];
        }
      };
      if (sF1 && A.f1Compact) {
        // Round 6: the first level in LDS masked LANES of the 64 bitmap gathers a 4096-row query issues, but the
        // texture path is bounded by gather INSTRUCTIONS here -- per-CU throughput did not move.  So the rows that pass the first level (the
        // set fraction of the folded bitmap, 12-24 % of the rows) are appended to a small ring in LDS (the record area is free in this phase)
        // and the bitmap is asked only for FULL wavefronts of them: 8-16 gather instructions per query instead of 64.  A batch's words are
        // requested one step before they are tested (the next batch's request, or the end of the enumeration, stands between).
        uint64_t* const sRing = sBin;  // 128 entries (row | bin id << 32)
        uint32_t c = 0, pendN = 0, pendW = 0;
        uint64_t pendE = 0;
        auto consume = [&]() {
          if (pendN) {  // (uniform)
            const bool bit = lane < pendN && ((pendW >> (pqt_hash_filter((uint32_t)(pendE >> 32), filterBits) & 31u)) & 1u);
            uint32_t tot;
            const uint32_t rk = pqt_ballot_rank(bit, &tot);
            if (bit && nwork + rk < 512) sWork[nwork + rk] = pendE;
            nwork += tot;
          }
        };
        auto issue = [&](const uint32_t take) {  // the first `take` (<= 64) ring entries leave: their bitmap words are requested, the rest moves down
          const uint64_t e = lane < take ? sRing[lane] : 0ull;
          const uint32_t wv = lane < take ? filter[pqt_hash_filter((uint32_t)(e >> 32), filterBits) >> 5] : 0u;
          const uint64_t mv = 64u + lane < c ? sRing[64u + lane] : 0ull;
          __builtin_amdgcn_wave_barrier();
          if (64u + lane < c) sRing[lane] = mv;
          __builtin_amdgcn_wave_barrier();
          consume();  // the batch requested one step ago
          pendE = e; pendW = wv; pendN = take;
          c -= take;
        };
        for (uint32_t hb = 0; hb < He; hb += 512) {
          if (hb == 0) {
#pragma unroll
            for (int r = 0; r < 8; ++r) { const uint32_t h = lane + 64 * r; hw[r] = h < He ? heur4[h] : 0u; }
          }
          uint32_t w[8];
#pragma unroll
          for (int r = 0; r < 8; ++r) w[r] = hw[r];
          if (hb + 512 < He) {
#pragma unroll
            for (int r = 0; r < 8; ++r) { const uint32_t h = hb + 512 + lane + 64 * r; hw[r] = h < He ? heur4[h] : 0u; }
          }
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            const uint32_t h = hb + lane + 64 * r;
            uint32_t gg = 0;
#pragma unroll
            for (int p = 0; p < 4; ++p) if ((uint32_t)p < P) gg += sSegB[PQT_MUL((uint32_t)p, WC, shWC) + ((w[r] >> (8 * p)) & 0xffu)];
            const uint32_t i1 = pqt_hash_filter(gg, filterBits) >> (filterBits - A.filter1Bits);
            const bool pass1 = h < He && ((sF1[i1 >> 5] >> (i1 & 31u)) & 1u);
            uint32_t tot;
            const uint32_t rk = pqt_ballot_rank(pass1, &tot);
            if (pass1) sRing[c + rk] = (uint64_t)h | ((uint64_t)gg << 32);
            c += tot;
            __builtin_amdgcn_wave_barrier();
            if (c >= 64u) issue(64u);  // (uniform)
          }
        }
        if (c) issue(c);
        consume();
      } else {
      stageA(0, gA, fA);
      for (uint32_t hb = 0; hb < He; hb += 512) {
        uint32_t gB[8], fB[8];
        const bool more = hb + 512 < He;
        if (more) stageA(hb + 512, gB, fB);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const uint32_t h = hb + lane + 64 * r;
          const bool bit = h < He && ((fA[r] >> (pqt_hash_filter(gA[r], filterBits) & 31u)) & 1u);
          uint32_t tot;
          const uint32_t rk = pqt_ballot_rank(bit, &tot);
          if (bit && nwork + rk < 512) sWork[nwork + rk] = (uint64_t)h | ((uint64_t)gA[r] << 32);
          nwork += tot;
        }
        if (more) {
#pragma unroll
          for (int r = 0; r < 8; ++r) { gA[r] = gB[r]; fA[r] = fB[r]; }
        }
      }
      }
      __builtin_amdgcn_wave_barrier();
      if (nwork > 512) {  // more "maybe" rows than the list holds: workgroup-per-query kernel with the full-size arena
        if (A.gbins) { if (lane == 0) { A.gbins[(size_t)q * (A.gbinCap + 1u) + A.gbinCap] = 0xffffffffull; nCand[q] = 0; nLocal[q] = 0; nIncl[q] = 0; } return; }
        if (lane == 0) { ovList[atomicAdd(ovCount, 1u)] = q; nCand[q] = 0; nLocal[q] = 0; nIncl[q] = 0; if (A.nRuns) A.nRuns[q] = 0xffffffffu; }
        return;
      }
      probeWork(sWork, nwork);
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        uint32_t tot;
        const uint32_t rk = pqt_ballot_rank(recG[r] != 0, &tot);
        const uint32_t e = npop + rk;
        if (recG[r]) {  // npop <= nwork <= 512
          sBin[e] = (uint64_t)recG[r] | ((uint64_t)recL[r] << 32);
          sKeyW[e] = (key[r] & 0xffffffff00000000ull) | ((key[r] & 0xffffull) << 16) | e;  // (distance, row, record)
        }
        npop += tot;
      }
    } else
    for (uint32_t hb = 0; hb < He; hb += 512) {
      rowBlock(hb);
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        uint32_t tot;
        const uint32_t rk = pqt_ballot_rank(recG[r] != 0, &tot);
        const uint32_t e = npop + rk;
        if (recG[r] && e < 512) {
          sBin[e] = (uint64_t)recG[r] | ((uint64_t)recL[r] << 32);
          sKeyW[e] = (key[r] & 0xffffffff00000000ull) | ((key[r] & 0xffffull) << 16) | e;  // (distance, row, record)
        }
        npop += tot;
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (npop > 512) {
      if (A.gbins) { if (lane == 0) { A.gbins[(size_t)q * (A.gbinCap + 1u) + A.gbinCap] = 0xffffffffull; nCand[q] = 0; nLocal[q] = 0; nIncl[q] = 0; } return; }
      if (lane == 0) { ovList[atomicAdd(ovCount, 1u)] = q; nCand[q] = 0; nLocal[q] = 0; nIncl[q] = 0; if (A.nRuns) A.nRuns[q] = 0xffffffffu; }
      return;
    }
    if (npop <= 128) {
      uint64_t k2[2];
#pragma unroll
      for (int r = 0; r < 2; ++r) k2[r] = lane * 2 + r < npop ? sKeyW[lane * 2 + r] : ~0ull;
      pqt_wave_sort_u64<2>(k2);
      PQT_TS(6);
      totIncl = finish(k2, npop, totCand);
    } else {
#pragma unroll
      for (int r = 0; r < 8; ++r) key[r] = lane * 8 + r < npop ? sKeyW[lane * 8 + r] : ~0ull;
      pqt_wave_sort_u64<8>(key);
      PQT_TS(6);
      totIncl = finish(key, npop, totCand);
    }
    if (lane == 0) nIncl[q] = totIncl;
    PQT_TS(8);
    return;
  }
#pragma unroll
  for (int r = 0; r < 8; ++r) npop += (uint32_t)__popcll(__ballot(recG[r] != 0));  // scalar; the ranks are taken where they are used
  if (npop <= 128 && !forceFullOrder) {
    uint64_t* sKeyC = sBin + 128;
    uint32_t base = 0;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      uint32_t tot;
      const uint32_t e = base + pqt_ballot_rank(recG[r] != 0, &tot);
      if (recG[r]) {
        sBin[e] = (uint64_t)recG[r] | ((uint64_t)recL[r] << 32);
        sKeyC[e] = (key[r] & 0xffffffff00000000ull) | ((key[r] & 0xffffull) << 16) | e;  // (distance, row, record)
      }
      base += tot;
    }
    __builtin_amdgcn_wave_barrier();
    if (npop <= 64) {
      uint64_t k1[1] = {lane < npop ? sKeyC[lane] : ~0ull};
      pqt_wave_sort_u64<1>(k1);
      PQT_TS(6);
      totIncl = finish(k1, npop, totCand);
    } else {
      uint64_t k2[2];
#pragma unroll
      for (int r = 0; r < 2; ++r) k2[r] = lane * 2 + r < npop ? sKeyC[lane * 2 + r] : ~0ull;
      pqt_wave_sort_u64<2>(k2);
      PQT_TS(6);
      totIncl = finish(k2, npop, totCand);
    }
  } else if (twoShort) {
    // more than 128 populated rows: records compacted like above, their keys (distance, row, record) stay in registers
    uint32_t base = 0;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      uint32_t tot;
      const uint32_t e = base + pqt_ballot_rank(recG[r] != 0, &tot);
      if (recG[r]) {
        sBin[e] = (uint64_t)recG[r] | ((uint64_t)recL[r] << 32);
        key[r] = (key[r] & 0xffffffff00000000ull) | ((key[r] & 0xffffull) << 16) | e;
      }
      base += tot;
    }
    __builtin_amdgcn_wave_barrier();
    pqt_wave_sort_u64<8>(key);
    PQT_TS(6);
    totIncl = finish(key, npop, totCand);
  } else {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const uint32_t h = lane + 64 * r;
      if (h < He) sBin[h] = (uint64_t)recG[r] | ((uint64_t)recL[r] << 32);
    }
    __builtin_amdgcn_wave_barrier();
    pqt_wave_sort_u64<8>(key);
    PQT_TS(6);
    totIncl = finish(key, He, totCand);
  }
  if (lane == 0) nIncl[q] = totIncl;
  PQT_TS(8);
  if (tstamp && lane == 0) tstamp[(size_t)q * PQT_TS_WORDS + 15] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (uint32_t)__builtin_amdgcn_s_getreg((31 << 11) | 4);
#undef PQT_TS
#undef PQT_MUL
#undef PQT_DIV
#undef PQT_MOD
}

#ifndef PQT_TR_WPS
#define PQT_TR_WPS 5   // waves per SIMD the register allocator must leave room for
#endif
template <int NW, int WCR, bool SHARDED, bool P2, int SHAPE = 0>
__global__ __launch_bounds__(NW * 64, PQT_TR_WPS) void pqt_k_traverse(const PqtTravArgs A /* kernel-argument segment: scalar loads */, uint32_t perWaveBytes) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const uint32_t wave = threadIdx.x >> 6;
  uint32_t q = blockIdx.x * NW + wave;
  if (q >= A.qn) return;
  if (A.qlist) { if (q >= *A.qcount) return; q = A.qlist[q]; }
  pqt_traverse_query<WCR, SHARDED, P2, SHAPE>(A, q, smem_raw + (size_t)wave * perWaveBytes, perWaveBytes);
}

// Wide enumeration with the first level of the presence bitmap in LDS: NW wavefronts (= queries) per workgroup share one copy of
// A.filter1 in front of their private slices; everything else is pqt_k_traverse.
template <int NW, int WCR, bool SHARDED, bool P2, int SHAPE = 0>
__global__ __launch_bounds__(NW * 64, 2) void pqt_k_traverse_f1(const PqtTravArgs A, uint32_t perWaveBytes) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const uint32_t wave = threadIdx.x >> 6;
  const uint32_t f1Words = 1u << (A.filter1Bits - 5u);
  uint32_t* const sF1 = reinterpret_cast<uint32_t*>(smem_raw);
  for (uint32_t t = threadIdx.x; t < f1Words / 4; t += NW * 64) reinterpret_cast<uint4*>(sF1)[t] = reinterpret_cast<const uint4*>(A.filter1)[t];
  __syncthreads();
  uint32_t q = blockIdx.x * NW + wave;
  if (q >= A.qn) return;
  if (A.qlist) { if (q >= *A.qcount) return; q = A.qlist[q]; }
  pqt_traverse_query<WCR, SHARDED, P2, SHAPE>(A, q, smem_raw + (size_t)f1Words * 4 + (size_t)wave * perWaveBytes, perWaveBytes, sF1);
}

// ---------------------------------------------------------------------------------------------------
// One launch for the whole query (SIFT1M shape: compile-time traversal shape 1, rerank with the coarse table in LDS): a wavefront
// draws a query, traverses it (pqt_traverse_query) and reranks it (pqt_rs_query) before it draws the next.  Two launches kept the
// latency-bound traversal (VALU mostly idle) and the VALU-bound rerank apart; here the wavefronts of a SIMD are in different phases
// at any time.  The traversal arena and the rerank's key/table slots of a wavefront share one LDS region (used one after the
// other).  The traversal's outputs (candidate list, counts, L1virt) go through global memory exactly as between the two launches:
// the wavefront that wrote them reads them back after s_waitcnt vmcnt(0) (same CU, write-through L1).
// Schedule: workgroup b owns the queries b, b + G, ...; its wavefronts draw them in index order through an LDS ticket (the candidate
// counts that the two-launch rerank orders by are not known before the traversal).
// MEASURED AND NOT THE DEFAULT: 0.191 ms per 10 k queries against 0.065 + 0.005 + 0.101 = 0.167 ms for the two launches (same box, same
// results).  The coarse table (64 KB) + 6.7 KB per wavefront cap a workgroup at 12 wavefronts = 3 per SIMD, where the stand-alone traversal
// runs 5 per SIMD to hide its dependent round trips; the merged body needs 168 VGPRs (39 spilled at the 12-wavefront budget; 8 wavefronts
// with 218 VGPRs and no spills: 0.222 ms), is 110 KB of code for a 64 KB instruction cache, and loses the longest-first order.  Kept as
// the opt-in "one_launch" option with its parity test; the L1virt round trip it was meant to save is 2 KB per query out of L2.
// ---------------------------------------------------------------------------------------------------
template <int NW, int LPV, int UREQ, int C1M, int SHAPE>
__global__ __launch_bounds__(NW * 64) void pqt_k_query_fused(const PqtTravArgs TA, const PqtRsArgs RA, uint32_t perWaveBytes) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr uint32_t LP = LPV * 4, C1 = 1u << C1M, nCoarse = LP * C1 * C1;
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* sCoarse = (float*)smem_raw;  // offset 0: pqt_rs_query's compile-time-stride addressing assumes it
  unsigned char* const wbase = smem_raw + (size_t)nCoarse * 4 + (size_t)wave * perWaveBytes;
  uint64_t* const sKeys = (uint64_t*)wbase;
  float* const sVirt = (float*)(wbase + (size_t)(PQT_RS_BEST + PQT_RS_PEND) * 8);
  uint32_t* const sTicket = reinterpret_cast<uint32_t*>(smem_raw + (size_t)nCoarse * 4 + (size_t)NW * perWaveBytes);  // [0] ticket, [1] ties
  if (threadIdx.x < 2) sTicket[threadIdx.x] = 0;
  if (blockIdx.x == 0 && threadIdx.x < 8 && RA.zero8) RA.zero8[threadIdx.x] = 0;
  for (uint32_t t = threadIdx.x; t < nCoarse; t += NW * 64) sCoarse[t] = RA.coarse[t];
  __syncthreads();
  const uint32_t G = gridDim.x, qn = RA.qn;
  const uint32_t L = blockIdx.x < qn ? (qn - blockIdx.x + G - 1) / G : 0u;
  const uint32_t slot = blockIdx.x * NW + wave;
  uint32_t tiesAcc = 0;
  for (;;) {
    uint32_t t = 0;
    if (lane == 0) t = atomicAdd(sTicket, 1u);
    t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
    if (t >= L) break;
    const uint32_t q = blockIdx.x + t * G;
    pqt_traverse_query<1, false, true, SHAPE>(TA, q, wbase, perWaveBytes);
    // the wavefront's own stores (candidate list, nLocal, L1virt) before its own loads of them
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    const uint32_t n = RA.nLocal[q];
    uint32_t nN = 0;
    pqt_rs_query<LPV, UREQ, true, false, C1M, 0, false>(RA, q, n, sKeys, sVirt, sCoarse, 0xffffffffu, nN, slot, tiesAcc, nullptr);
    __builtin_amdgcn_wave_barrier();
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) tiesAcc += (uint32_t)__shfl_xor((int)tiesAcc, d, 64);
  if (lane == 0 && tiesAcc) atomicAdd(&sTicket[1], tiesAcc);
  __syncthreads();
  if (threadIdx.x == 0) { const uint32_t tt = *(volatile uint32_t*)&sTicket[1]; if (tt) atomicAdd(&RA.counters[3], (unsigned long long)tt); }
}

// ---------------------------------------------------------------------------------------------------
// Query-sharded traversal, receiving side (pqt_query_shard_bins).  The traversal of a query ran on ANOTHER shard
// (pqt_traverse_bins) and arrives as its list of included populated bins (bin id | global start << 32, visiting order).
//   pqt_l1virt_block   a1 only: L1virt[lp][c] of every query (the rerank's distance table), same sums as the traversal.
//   pqt_resolve_block  one wavefront per query: looks every listed bin up in THIS shard's table (local start, local members,
//                      members on lower shards), scans the local populations, and leaves exactly what the SHARDED traversal's
//                      finish step leaves -- bin runs (first local visiting position | first store row << 32, global position of
//                      the first local member) or the expanded candidate list, nLocal, the schedule registration.  A query whose
//                      list overflowed at the sender (trailer count 0xffffffff) is appended to tvList and traversed here by the
//                      list-mode traversal kernel.
// ---------------------------------------------------------------------------------------------------
// a1 for QB queries per workgroup (the distance table L1virt[lp][c] of the rerank, same sums as the traversal).
// 4-dim line parts (configs[2]/[3]): a thread owns 2 x 4 consecutive table entries, keeps their centroid pieces (read coalesced from
// the line-part-major copy cb1L[lp][c][4]) in registers across the QB queries and stores float4 results straight to HBM -- no LDS
// transpose, no barrier in the loop.  History of this kernel per 10 k queries: (lp, c)-indexed reads of the row-major codebook 0.15 ms
// (one cache line per lane: texture addresser); LDS transpose per query 0.045 ms, of which 0.035 the LDS pipe (8-way conflicting
// ds_write_b32 + one ds_read_b128 per entry), measured by skipping the compute / the stores in turn; now the 80 MB of stores bound it.
// Other shapes: one query at a time through an LDS transpose, reading the row-major codebook contiguously (t = c*LP + lp).
template <int QB>
__device__ __forceinline__ void pqt_l1virt_block(const float* __restrict__ Q, const float* __restrict__ cb1, const float* __restrict__ cb1L,
                                                 const PqtDevParams& prm, float* __restrict__ qL1virt, uint32_t qn, uint32_t blk, float* smem) {
  const uint32_t D = prm.D, C1 = prm.C1, LP = prm.LP, SS = prm.SS, N = LP * C1;
  float* sQ = smem;            // QB * D
  float* sV = smem + QB * D;   // N (generic path)
  const uint32_t q0 = blk * QB, tid = threadIdx.x;
  if (q0 >= qn) return;
  const uint32_t nq = qn - q0 < (uint32_t)QB ? qn - q0 : (uint32_t)QB;
  for (uint32_t i = tid; i < nq * D; i += PQT_BLOCK) sQ[i] = Q[(size_t)q0 * D + i];
  if (cb1L && SS == 4 && (C1 & 3u) == 0 && N <= 8 * PQT_BLOCK) {
    float4 cv[2][4];
    uint32_t qo[2];
    bool on[2];
#pragma unroll
    for (uint32_t i = 0; i < 2; ++i) {
      const uint32_t t0 = (tid + i * PQT_BLOCK) * 4;
      on[i] = t0 < N;
      const uint32_t tt = on[i] ? t0 : 0u;
      qo[i] = (tt / C1) * 4;
#pragma unroll
      for (uint32_t j = 0; j < 4; ++j) cv[i][j] = reinterpret_cast<const float4*>(cb1L)[tt + j];
    }
    __syncthreads();
    for (uint32_t qi = 0; qi < nq; ++qi) {
      float* outq = qL1virt + (size_t)(q0 + qi) * N;
#pragma unroll
      for (uint32_t i = 0; i < 2; ++i) {
        if (on[i]) {
          const float4 qq = *reinterpret_cast<const float4*>(sQ + qi * D + qo[i]);
          float r[4];
#pragma unroll
          for (uint32_t j = 0; j < 4; ++j) {
            float s = 0.f;
            float df = qq.x - cv[i][j].x; s = s + df * df;
            df = qq.y - cv[i][j].y; s = s + df * df;
            df = qq.z - cv[i][j].z; s = s + df * df;
            df = qq.w - cv[i][j].w; s = s + df * df;
            r[j] = s;
          }
          *reinterpret_cast<float4*>(outq + (tid + i * PQT_BLOCK) * 4) = make_float4(r[0], r[1], r[2], r[3]);
        }
      }
    }
    return;
  }
  for (uint32_t qi = 0; qi < nq; ++qi) {
    const float* qv = sQ + qi * D;
    __syncthreads();  // sQ staged / the previous query's table read out
    for (uint32_t t = tid; t < N; t += PQT_BLOCK) {
      const uint32_t c = t / LP, lp = t % LP;
      const float* cen = cb1 + (size_t)c * D + lp * SS;
      const float* qq = qv + lp * SS;
      float s = 0.f;
      if ((SS & 3u) == 0) {
        for (uint32_t d = 0; d < SS; d += 4) {
          const float4 c4 = *reinterpret_cast<const float4*>(cen + d);
          float df = qq[d] - c4.x; s = s + df * df;
          df = qq[d + 1] - c4.y; s = s + df * df;
          df = qq[d + 2] - c4.z; s = s + df * df;
          df = qq[d + 3] - c4.w; s = s + df * df;
        }
      } else {
        for (uint32_t d = 0; d < SS; ++d) { const float df = qq[d] - cen[d]; s = s + df * df; }
      }
      sV[lp * C1 + c] = s;
    }
    __syncthreads();
    float* outq = qL1virt + (size_t)(q0 + qi) * N;
    for (uint32_t t = tid; t < N; t += PQT_BLOCK) outq[t] = sV[t];
  }
}
#define PQT_L1V_QB 8

struct PqtResolveArgs {
  const unsigned long long* gbins; uint32_t gbinCap;  // [qn][gbinCap + 1], see PqtTravArgs
  const PqtBinEntry* table; const uint32_t* lower; uint32_t tableBits, tableSeed;
  uint32_t qn;
  uint32_t* cand; uint32_t* candPos; uint64_t stride;
  uint32_t* nCand; uint32_t* nLocal; uint32_t* nIncl;
  unsigned long long* runs; uint32_t* runGpos; uint32_t* nRuns; uint32_t runCap;
  uint32_t* outCount;
  uint32_t* tvList; uint32_t* tvCount;
  uint32_t* schedCnt; unsigned long long* schedList; uint32_t schedCap;
};
#define PQT_GBIN_MAX 256  // largest per-query list pqt_resolve_block accepts (EPL = 4 entries per lane; 2 for lists of <= 128)
template <int NW, int EPL>
__device__ __forceinline__ void pqt_resolve_block(const PqtResolveArgs& A, uint32_t blk) {
  __shared__ unsigned long long sBinAll[NW][64 * EPL];
  __shared__ uint32_t sGposAll[NW][64 * EPL];
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t q = blk * NW + wave;
  if (q >= A.qn) return;
  unsigned long long* const sBin = sBinAll[wave];
  uint32_t* const sGpos = sGposAll[wave];
  const unsigned long long* row = A.gbins + (size_t)q * (A.gbinCap + 1u);
  const unsigned long long trailer = row[A.gbinCap];
  const uint32_t m = (uint32_t)trailer, cnt = (uint32_t)(trailer >> 32);
  if (m == 0xffffffffu) {  // the sender could not list this query's bins: traverse it here
    if (lane == 0) { A.tvList[atomicAdd(A.tvCount, 1u)] = q; A.nCand[q] = 0; A.nLocal[q] = 0; A.nIncl[q] = 0; if (A.nRuns) A.nRuns[q] = 0xffffffffu; }
    return;
  }
  const uint4* table4 = reinterpret_cast<const uint4*>(A.table);
  uint32_t ls[EPL], lc[EPL], gp[EPL];
  uint32_t myLocal = 0, myBins = 0;
#pragma unroll
  for (int r = 0; r < EPL; ++r) {
    const uint32_t e = lane * EPL + r;  // blocked: the wave scans below run in entry (= visiting) order
    ls[r] = 0; lc[r] = 0; gp[r] = 0;
    if (e < m) {
      const unsigned long long ent = row[e];
      uint32_t slot = 0;
      const uint4 x = pqt_table_lookup(table4, (uint32_t)ent, A.tableBits, A.tableSeed, &slot);
      if (x.y) { ls[r] = x.z; lc[r] = x.w; gp[r] = (uint32_t)(ent >> 32) + A.lower[slot]; }
    }
    myLocal += lc[r];
    myBins += lc[r] ? 1u : 0u;
  }
  const uint32_t locIncl = pqt_wave_incl_scan(myLocal);
  const uint32_t totLocal = __shfl(locIncl, 63, 64);
  uint32_t lrun = locIncl - myLocal;
  const uint32_t nbIncl = pqt_wave_incl_scan(myBins);
  const uint32_t mL = __shfl(nbIncl, 63, 64);  // listed bins with members on this shard
  uint32_t wpos = nbIncl - myBins;
#pragma unroll
  for (int r = 0; r < EPL; ++r) {
    if (lc[r]) { sBin[wpos] = (unsigned long long)lrun | ((unsigned long long)ls[r] << 32); sGpos[wpos] = gp[r]; ++wpos; lrun += lc[r]; }
  }
  __builtin_amdgcn_wave_barrier();
  uint32_t schedSlot = 0, schedPos = 0;
  if (lane == 0) {
    A.nCand[q] = cnt; A.nLocal[q] = totLocal; A.nIncl[q] = m;
    if (A.outCount) A.outCount[q] = cnt;
    if (A.schedCnt) {
      schedSlot = (q & 7u) * PQT_SCHED_CLASSES + pqt_sched_class(totLocal);
      schedPos = __hip_atomic_fetch_add(&A.schedCnt[schedSlot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (schedPos < A.schedCap) A.schedList[(size_t)schedSlot * A.schedCap + schedPos] = (unsigned long long)q | ((unsigned long long)totLocal << 32);
    }
  }
  if (A.runs) {
    if (mL <= A.runCap) {
      for (uint32_t i = lane; i < mL; i += 64) { A.runs[(size_t)q * PQT_RUNCAP + i] = sBin[i]; A.runGpos[(size_t)q * PQT_RUNCAP + i] = sGpos[i]; }
      if (lane == 0) A.nRuns[q] = mL;
      return;
    }
    if (lane == 0) A.nRuns[q] = 0xffffffffu;
  }
  uint32_t* const out = A.cand + (size_t)q * A.stride;
  uint32_t* const outP = A.candPos + (size_t)q * A.stride;
  if (totLocal >= 32u * mL) {  // long bins: walk the listed bins with coalesced stores
    for (uint32_t b = 0; b < mL; ++b) {
      const unsigned long long be = sBin[b];
      const uint32_t s0 = (uint32_t)be, l0 = (uint32_t)(be >> 32), g0 = sGpos[b];
      const uint32_t e0 = b + 1 < mL ? (uint32_t)sBin[b + 1] : totLocal;
      for (uint32_t j = s0 + lane; j < e0; j += 64) { out[j] = l0 + (j - s0); outP[j] = g0 + (j - s0); }
    }
  } else {
    for (uint32_t j = lane; j < totLocal; j += 64) {
      uint32_t lo = 0, hi = mL;  // last entry with start <= j
      while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if ((uint32_t)sBin[mid] <= j) lo = mid; else hi = mid;
      }
      const unsigned long long be = sBin[lo];
      const uint32_t off = j - (uint32_t)be;
      out[j] = (uint32_t)(be >> 32) + off;
      outP[j] = sGpos[lo] + off;
    }
  }
}
// one launch for both halves of the receiving side: the first nTab workgroups compute the distance tables (store-bound, fewer than the
// chip holds at once), the rest resolve the bin lists (latency chains: table look-up, scans, run write-out) and start beside them --
// two independent jobs that used to queue behind each other
template <int NW, int EPL>
__global__ __launch_bounds__(NW * 64) void pqt_k_tables_resolve(const PqtResolveArgs A, uint32_t nTab, const float* __restrict__ Q,
                                                                 const float* __restrict__ cb1, const float* __restrict__ cb1L,
                                                                 const PqtDevParams prm, float* __restrict__ qL1virt) {
  static_assert(NW * 64 == PQT_BLOCK, "the table half strides by PQT_BLOCK");
  extern __shared__ __attribute__((aligned(16))) float smem[];  // PQT_L1V_QB * D + LP * C1 floats
  if (blockIdx.x < nTab) pqt_l1virt_block<PQT_L1V_QB>(Q, cb1, cb1L, prm, qL1virt, A.qn, blockIdx.x, smem);
  else pqt_resolve_block<NW, EPL>(A, blockIdx.x - nTab);
}

// marks every query of a pqt_traverse_bins request as "traverse it yourself" (shapes the fused traversal does not cover)
#ifdef PQT_MAIN_TU
__global__ __launch_bounds__(256) void pqt_k_gbins_overflow(unsigned long long* __restrict__ gbins, uint32_t cap, uint32_t qn) {
  const uint32_t q = blockIdx.x * 256 + threadIdx.x;
  if (q < qn) gbins[(size_t)q * (cap + 1u) + cap] = 0xffffffffull;
}
#endif  // PQT_MAIN_TU

// ---------------------------------------------------------------------------------------------------
// offline ("next" row 8f-3): E step of the reference's Lloyd iterations (productquantizer.hpp:40-66,
// vectorquantizer.hpp:33-53): for every row the nearest of ncen centroids over `dim` dims, squared distance
// summed left to right, first minimum wins (strict '<').  lane = one row; centroids staged in LDS.
// rows (optional) selects/gathers the rows (group(), treequantizer.hpp:140-147).
// ---------------------------------------------------------------------------------------------------
#ifdef PQT_MAIN_TU
__global__ __launch_bounds__(256) void pqt_k_kmeans_assign(
    const float* __restrict__ x, uint64_t n, uint32_t dim, uint32_t ld, const uint32_t* __restrict__ rows,
    const float* __restrict__ cen, uint32_t ncen, uint32_t cenLd, uint32_t* __restrict__ outAssign,
    float* __restrict__ outDist) {
  extern __shared__ __attribute__((aligned(16))) float sCen[];  // ncen * dim
  for (uint32_t t = threadIdx.x; t < ncen * dim; t += 256) sCen[t] = cen[(size_t)(t / dim) * cenLd + t % dim];
  __syncthreads();
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float* xr = x + (size_t)(rows ? rows[i] : i) * ld;
  uint32_t best = 0;
  float bd = __uint_as_float(0x7f800000u);  // HUGE_VAL
  for (uint32_t c = 0; c < ncen; ++c) {
    float s = 0.f;
    for (uint32_t d = 0; d < dim; ++d) { const float df = xr[d] - sCen[c * dim + d]; s = s + df * df; }
    if (s < bd) { bd = s; best = c; }
  }
  outAssign[i] = best;
  outDist[i] = bd;
}
#endif  // PQT_MAIN_TU

// ---------------------------------------------------------------------------------------------------
// PMC calibration probe (MI355X_MICROARCH.md HBM: "calibrate on a known byte count in your own access pattern"):
// `gathers` random row reads of ROWV 16-byte vectors each (lane = one row, exactly the access shape of the rerank
// kernels) from a table far larger than the 256 MiB Infinity Cache.  Known bytes = gathers * ROWV * 16.
// ---------------------------------------------------------------------------------------------------
template <int ROWV>
__global__ __launch_bounds__(256) void pqt_k_calib_gather(const uint4* __restrict__ table, uint64_t tableRows, uint64_t gathers,
                                                           unsigned long long* __restrict__ sink) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= gathers) return;
  // bijective scramble of i over the table (odd multiplier modulo a power of two) -> every row at most once
  const uint64_t row = (i * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull) & (tableRows - 1);
  uint32_t acc = 0;
#pragma unroll
  for (int v = 0; v < ROWV; ++v) { const uint4 x = table[row * ROWV + v]; acc ^= x.x ^ x.y ^ x.z ^ x.w; }
  if (acc == 0x12345678u) atomicAdd(sink, 1ull);  // keeps the loads alive
}

// read-only streaming probe: every 16-byte piece of a buffer exactly once, four independent loads per lane in flight (the
// access shape of the group-major rerank: 64 lanes = one contiguous KB).  bench.py reports its GB/s beside the nominal peak.
#ifdef PQT_MAIN_TU
// U 16-byte loads in flight per lane.  CHUNK = false: grid-stride (consecutive workgroups read consecutive 4 KB pieces, a workgroup's next
// piece is gridDim * 4 KB further); CHUNK = true: a workgroup walks its own contiguous share.
template <int U, bool CHUNK>
__global__ __launch_bounds__(256) void pqt_k_stream_read(const uint4* __restrict__ p, uint64_t n16, unsigned long long* __restrict__ sink) {
  uint32_t acc = 0;
  if constexpr (CHUNK) {
    const uint64_t per = (n16 + gridDim.x - 1) / gridDim.x;
    const uint64_t lo = (uint64_t)blockIdx.x * per, hi = lo + per < n16 ? lo + per : n16;
    uint64_t i = lo + threadIdx.x;
    for (; i + (uint64_t)(U - 1) * 256 < hi; i += (uint64_t)U * 256) {
      uint4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = p[i + (uint64_t)u * 256];
#pragma unroll
      for (int u = 0; u < U; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    for (; i < hi; i += 256) { const uint4 a = p[i]; acc ^= a.x ^ a.y ^ a.z ^ a.w; }
  } else {
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (uint64_t)(U - 1) * stride < n16; i += (uint64_t)U * stride) {
      uint4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = p[i + (uint64_t)u * stride];
#pragma unroll
      for (int u = 0; u < U; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    for (; i < n16; i += stride) { const uint4 a = p[i]; acc ^= a.x ^ a.y ^ a.z ^ a.w; }
  }
  if (acc == 0x12345678u) atomicAdd(sink, 1ull);  // keeps the loads alive
}
#endif  // PQT_MAIN_TU

// the same gather with ROWV lanes per row: lane c of a group reads piece c, so a row is one contiguous ROWV*16-byte access of
// adjacent lanes instead of ROWV separate 16-byte accesses of one lane at 16-byte steps
template <int ROWV>
__global__ __launch_bounds__(256) void pqt_k_calib_gather_coop(const uint4* __restrict__ table, uint64_t tableRows, uint64_t gathers,
                                                                unsigned long long* __restrict__ sink) {
  const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const uint64_t i = t / ROWV;
  if (i >= gathers) return;
  const uint64_t row = (i * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull) & (tableRows - 1);
  const uint4 x = table[row * ROWV + (t % ROWV)];
  const uint32_t acc = x.x ^ x.y ^ x.z ^ x.w;
  if (acc == 0x12345678u) atomicAdd(sink, 1ull);
}

// ---------------------------------------------------------------------------------------------------
// one-off at load time: the line store is permuted into BIN ORDER (row pos holds the code of vector ids[pos]), so
// the candidates of a bin are consecutive rows: rerank reads become short sequential runs instead of one random
// row per candidate, the id indirection disappears from the rerank chain, and the traversal emits positions
// without touching ids[].  lane = one 16-byte (or 4-byte) piece of a row; writes coalesced, reads gathered.
// ---------------------------------------------------------------------------------------------------
#ifdef PQT_MAIN_TU
__global__ __launch_bounds__(256) void pqt_k_reorder_lines(const uint32_t* __restrict__ codes, uint64_t idBase, uint64_t nCodes,
                                                            const uint32_t* __restrict__ ids, uint64_t nIds, uint32_t LP,
                                                            uint32_t* __restrict__ out, unsigned long long* __restrict__ bad) {
  const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (LP % 4 == 0) {
    const uint32_t V = LP / 4;
    if (t >= nIds * V) return;
    const uint64_t pos = t / V, v = t % V;
    const uint64_t r = (uint64_t)ids[pos] - idBase;
    if (r >= nCodes) { atomicAdd(bad, 1ull); return; }
    reinterpret_cast<uint4*>(out)[pos * V + v] = reinterpret_cast<const uint4*>(codes)[r * V + v];
  } else {
    if (t >= nIds * LP) return;
    const uint64_t pos = t / LP, v = t % LP;
    const uint64_t r = (uint64_t)ids[pos] - idBase;
    if (r >= nCodes) { atomicAdd(bad, 1ull); return; }
    out[pos * LP + v] = codes[r * LP + v];
  }
}
#endif  // PQT_MAIN_TU

// ---------------------------------------------------------------------------------------------------
// "next" row 8f-4: exact re-rank of the first k results against the raw database vectors
// (CUDA rerankBIGKernelPerfect PerturbationProTree.cu:5532 / queryBIGKNNRerankPerfect :8703; cpu_version/Readme.md:
// "You may resort the k-th best vectors from this list exactly").
//   one wavefront per query, lane = candidate(s): squared L2 over D dims summed left to right, then the
//   (distance, previous rank) keys are sorted by the in-register network.  RAW_U8: rows are uint8 (.umem / bvecs).
// ---------------------------------------------------------------------------------------------------
template <int NW, int RR, bool RAW_U8>
__global__ __launch_bounds__(NW * 64) void pqt_k_rerank_exact(
    const float* __restrict__ Q, uint32_t qn, uint32_t D, uint32_t k, const uint32_t* __restrict__ inIdx,
    const void* __restrict__ raw, uint64_t rawIdBase, uint64_t rawRows, uint32_t* __restrict__ outIdx,
    float* __restrict__ outDist) {
  extern __shared__ __attribute__((aligned(16))) float smemf[];
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t q = blockIdx.x * NW + wave;
  if (q >= qn) return;
  float* sQ = smemf + (size_t)wave * D;
  for (uint32_t i = lane; i < D; i += 64) sQ[i] = Q[(size_t)q * D + i];
  __builtin_amdgcn_wave_barrier();
  uint64_t key[RR];
#pragma unroll
  for (int r = 0; r < RR; ++r) {
    const uint32_t pos = lane + 64 * r;
    key[r] = ~0ull;
    if (pos < k) {
      const uint32_t id = inIdx[(size_t)q * k + pos];
      const uint64_t row = (uint64_t)id - rawIdBase;
      if (id != 0xffffffffu && row < rawRows) {
        float s = 0.f;
        if (RAW_U8) {
          const uint8_t* x = reinterpret_cast<const uint8_t*>(raw) + row * D;
          for (uint32_t d = 0; d < D; ++d) { const float df = sQ[d] - (float)x[d]; s = s + df * df; }
        } else {
          const float* x = reinterpret_cast<const float*>(raw) + row * D;
          for (uint32_t d = 0; d < D; ++d) { const float df = sQ[d] - x[d]; s = s + df * df; }
        }
        key[r] = ((uint64_t)pqt_f2key(s) << 32) | pos;
      }
    }
  }
  pqt_wave_sort_u64<RR>(key);
#pragma unroll
  for (int r = 0; r < RR; ++r) {
    const uint32_t e = lane * RR + r;
    if (e < k) {
      const bool ok = key[r] != ~0ull;
      outIdx[(size_t)q * k + e] = ok ? inIdx[(size_t)q * k + (uint32_t)key[r]] : 0xffffffffu;
      outDist[(size_t)q * k + e] = ok ? pqt_key2f((uint32_t)(key[r] >> 32)) : __uint_as_float(0x7f800000u);
    }
  }
}

// ===================================================================================================
// Fused stage a7 + a8 for LARGE first-level codebooks (coarse[LP][C1][C1] > 64 KB: BASELINE cfg3/cfg4/cfg5 shapes,
// C1 = 64..128, lineparts = 32): one WORKGROUP (8 wavefronts) per query.
//
//   The 4th look-up of every ADC term, coarse[p][A][B], cannot live in LDS as a whole (512 KB .. 2 MB) and as a
//   4-byte gather through the texture path it costs ~64 clk per wave instruction -- the bound of pqt_k_rerank_select
//   at these shapes.  Here the table is staged into LDS one LINE-PART GROUP at a time (G line parts = 64 KB) and the
//   whole candidate tile (up to 8192 candidates, 16 per thread, accumulators in registers) is advanced by those G
//   terms before the next group is staged: 512 KB of L2->LDS traffic per query tile instead of one gather per term.
//   The sum per candidate still runs p = 0..LP-1 in order (groups ascend, terms ascend inside a group): bit-exact.
//   Selection: per wavefront as in pqt_k_rerank_select (tau filter, pending buffer, in-register sort), then wave 0
//   merges the 8 sorted best lists (3 sorts of 512).
// LDS: G*C1*C1*4 (<= 64 KB) + LP*C1*4 + NW * 512 * 8 bytes.
// ===================================================================================================
#define PQT_RS2_NW 8
#ifndef PQT_RS2_CPT
#define PQT_RS2_CPT 16  // candidates per thread per tile -> tile = 8192
#endif
#ifndef PQT_RS2_KEYS
#define PQT_RS2_KEYS 512  // key slots per wavefront: [best 128 | pending]; 512 -> sort<8>, 256 -> sort<4>
#endif
#ifndef PQT_RS2_PF
#define PQT_RS2_PF 6    // code-word prefetch distance (candidates per thread in flight; 4..8 measure the same)
#endif
#ifndef PQT_RS2_WPS
#define PQT_RS2_WPS 4    // waves per SIMD the register allocator must leave room for
#endif

template <int G, bool SHARDED, int C1M /* 0: any C1, 1: power of two, >= 2: C1 == 1 << C1M at compile time */>
__global__ __launch_bounds__(PQT_RS2_NW * 64, PQT_RS2_WPS) void pqt_k_rerank_select_wg(
    const uint32_t* __restrict__ codesGrp /* bin-ordered, group-major: [LP/G][nIds][G] */, uint64_t nIds,
    const uint32_t* __restrict__ ids, const float* __restrict__ qL1virt,
    const float* __restrict__ coarse, const uint32_t* __restrict__ cand, const uint32_t* __restrict__ candPos,
    const uint32_t* __restrict__ nLocal, uint64_t stride, uint32_t k, PqtDevParams prm,
    uint32_t* __restrict__ outIdx, float* __restrict__ outDist, uint32_t* __restrict__ outPos,
    unsigned long long* __restrict__ counters, uint32_t dbg /* ablation bits (wrong results): 1 no sorts, 8 no ADC arithmetic, 16 cache-resident rows */) {
  constexpr int NW = PQT_RS2_NW, NT = NW * 64, CPT = PQT_RS2_CPT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr bool C1P2 = C1M != 0;
  const uint32_t C1 = C1M >= 2 ? (1u << C1M) : prm.C1, LP = prm.LP;
  const uint32_t c1sh = C1M >= 2 ? (uint32_t)C1M : (C1P2 ? (uint32_t)__builtin_ctz(C1) : 0u);  // power-of-two C1: shifts instead of quarter-rate multiplies
  const uint32_t chunkFloats = G * C1 * C1;
  float* sChunk = (float*)smem_raw;
  float* sVirt = sChunk + chunkFloats;
  const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  uint64_t* sKeysAll = (uint64_t*)(sVirt + LP * C1);
  constexpr int KR = PQT_RS2_KEYS / 64;  // keys per lane in the sorting network
  uint64_t* sKeys = sKeysAll + (size_t)wave * PQT_RS2_KEYS;
  const uint32_t q = blockIdx.x;
  const uint32_t n = nLocal[q];
  const uint32_t* cid = cand + (size_t)q * stride;
  for (uint32_t t = tid; t < LP * C1; t += NT) sVirt[t] = qL1virt[(size_t)q * LP * C1 + t];

  uint64_t tau = ~0ull;
  uint32_t npend = 0, off0 = 0;
  auto flush = [&](const bool final) {
    // as in pqt_rs_query: more than 128 keys are cut down to the k smallest by the exact radix select, only the final
    // survivors of the wavefront are sorted (the merge below wants sorted, ~0-padded lists of 128)
    uint32_t have = off0 + npend;
    if (have > PQT_RS_BEST) {
      uint64_t key[KR];
#pragma unroll
      for (int r = 0; r < KR; ++r) {
        const uint32_t e = r * 64 + lane;
        key[r] = (e < have) ? sKeys[e] : ~0ull;
      }
      __builtin_amdgcn_wave_barrier();
      tau = pqt_wave_kth_u64<KR>(key, k, reinterpret_cast<uint32_t*>(sKeys + PQT_RS_BEST));
      uint32_t cnt = 0;
#pragma unroll
      for (int r = 0; r < KR; ++r) {
        uint32_t tot;
        const uint32_t rk = pqt_ballot_rank(key[r] <= tau, &tot);
        if (key[r] <= tau) sKeys[cnt + rk] = key[r];
        cnt += tot;
      }
      have = k;
      __builtin_amdgcn_wave_barrier();
    }
    if (final) {
      uint64_t key[2];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const uint32_t e = lane * 2 + r;
        key[r] = (e < have) ? sKeys[e] : ~0ull;
      }
      if (!(dbg & 1)) pqt_wave_sort_u64<2>(key);
#pragma unroll
      for (int r = 0; r < 2; ++r) sKeys[lane * 2 + r] = key[r];
      __builtin_amdgcn_wave_barrier();
    }
    npend = 0;
    off0 = have;
  };

  for (uint32_t tile0 = 0; tile0 < n; tile0 += NT * CPT) {
    const uint32_t tn = (n - tile0 < NT * CPT) ? n - tile0 : NT * CPT;
    uint32_t pos[CPT];
    float acc[CPT];
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const uint32_t j = tid + NT * i;
      pos[i] = cid[tile0 + (j < tn ? j : 0)];
      if (dbg & 16) pos[i] &= 4095u;
      acc[i] = 0.f;
    }
    for (uint32_t g = 0; g < LP / G; ++g) {
      // the G code words of this group for the first PF candidates of the thread are requested before the table slice is
      // staged (their round trip overlaps the staging and its barriers), the others PF candidates ahead of the
      // arithmetic that consumes them.  Group-major store: the G words of consecutive candidates (consecutive positions inside a
      // bin) are contiguous, so a wavefront's 64 reads coalesce into one 1 KB (G = 4) transaction.
      constexpr int PF = CPT < PQT_RS2_PF ? CPT : PQT_RS2_PF;  // candidates whose words are in flight ahead of the arithmetic
      uint32_t wBuf[PF][G];
      auto loadWords = [&](const int i) {
        const uint32_t* row = codesGrp + ((size_t)g * nIds + pos[i]) * G;
        if (G == 4) { const uint4 v = *reinterpret_cast<const uint4*>(row); wBuf[i % PF][0] = v.x; wBuf[i % PF][1 % G] = v.y; wBuf[i % PF][2 % G] = v.z; wBuf[i % PF][3 % G] = v.w; }
        else if (G == 2) { const uint2 v = *reinterpret_cast<const uint2*>(row); wBuf[i % PF][0] = v.x; wBuf[i % PF][1 % G] = v.y; }
        else { wBuf[i % PF][0] = row[0]; }
      };
#pragma unroll
      for (int i = 0; i < PF; ++i) loadWords(i);
      __syncthreads();  // everyone is done with the previous group's table (and sVirt is loaded)
      {
        const float4* src = reinterpret_cast<const float4*>(coarse + (size_t)g * chunkFloats);
        float4* dst = reinterpret_cast<float4*>(sChunk);
        for (uint32_t t = tid; t < chunkFloats / 4; t += NT) dst[t] = src[t];
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < CPT; ++i) {
        const uint32_t j = tid + NT * i;
        if (j < tn) {
          uint32_t w[G];
#pragma unroll
          for (int x = 0; x < G; ++x) w[x] = wBuf[i % PF][x];
          float a = acc[i];
          if (dbg & 8) { uint32_t xx = 0;
#pragma unroll
            for (int x = 0; x < G; ++x) xx ^= w[x];
            a = a + __uint_as_float(xx & 0x3fffffffu); } else
#pragma unroll
          for (int x = 0; x < G; ++x) {
            const uint32_t p = g * G + x;
            const uint32_t A = w[x] & 0xffu, B = (w[x] >> 8) & 0xffu;
            const float lam = __builtin_fmaf((float)(w[x] >> 16), 8.f / 65536.f, -4.f);
            float sb, sa, sc;
            if constexpr (C1M >= 2) {
              // compile-time strides: the table offsets fold into one uniform base and the DS immediate fields
              const uint32_t B4 = B << 2;
              const uint32_t vOffP = (uint32_t)(G << (2 * C1M + 2)) + ((p << C1M) << 2);  // byte offset of L1virt row p (uniform)
              sb = *reinterpret_cast<const float*>(smem_raw + ((A << 2) + vOffP));
              sa = *reinterpret_cast<const float*>(smem_raw + (B4 + vOffP));
              sc = *reinterpret_cast<const float*>(smem_raw + ((A << (2 + C1M)) + B4) + ((uint32_t)x << (2 * C1M + 2)));
            } else {
              const uint32_t pv = C1P2 ? (p << c1sh) : p * C1;
              sb = sVirt[pv + A];
              sa = sVirt[pv + B];
              sc = sChunk[C1P2 ? ((((uint32_t)x << c1sh) + A) << c1sh) + B : (x * C1 + A) * C1 + B];
            }
            a = a + pqt_extract_distance(sa, sb, sc, lam);
          }
          acc[i] = a;
        }
        if (i + PF < CPT) loadWords(i + PF);  // refill the slot just consumed
      }
    }
    // selection: wave-synchronous append of the survivors
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const uint32_t j = tid + NT * i;
      const bool valid = j < tn;
      const uint64_t key = ((uint64_t)pqt_f2key(acc[i]) << 32) | (tile0 + j);
      const bool pass = valid && key < tau;
      uint32_t tot;
      const uint32_t rk = pqt_ballot_rank(pass, &tot);
      if (pass) sKeys[off0 + npend + rk] = key;
      npend += tot;
      __builtin_amdgcn_wave_barrier();
      if (off0 + npend + 64 > PQT_RS2_KEYS) flush(false);
    }
  }
  flush(true);
  __syncthreads();
  // merge the NW sorted best lists (each at sKeysAll[w*512 .. +128)) in wave 0
  if (wave == 0) {
    for (uint32_t first = 1; first < NW; first += 3) {
      uint64_t key[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const uint32_t e = lane * 8 + r;
        const uint32_t w = e < PQT_RS_BEST ? 0u : first + (e / PQT_RS_BEST - 1);
        key[r] = w < NW ? sKeysAll[(size_t)w * PQT_RS2_KEYS + (e % PQT_RS_BEST)] : ~0ull;
      }
      pqt_wave_sort_u64<8>(key);
      if (lane < PQT_RS_BEST / 8) {
#pragma unroll
        for (int r = 0; r < 8; ++r) sKeys[lane * 8 + r] = key[r];
      }
      __builtin_amdgcn_wave_barrier();
    }
    const uint32_t kk = n < k ? n : k;
    uint32_t ties = 0;
    for (uint32_t i = lane; i < k; i += 64) {
      const size_t o = (size_t)q * k + i;
      if (i < kk) {
        const uint64_t key = sKeys[i];
        const uint32_t j = (uint32_t)key;
        outIdx[o] = ids[cid[j]];
        outDist[o] = pqt_key2f((uint32_t)(key >> 32));
        if (SHARDED) outPos[o] = candPos[(size_t)q * stride + j];
        if (i + 1 < kk && (uint32_t)(sKeys[i + 1] >> 32) == (uint32_t)(key >> 32)) ++ties;
      } else {
        outIdx[o] = 0xffffffffu;
        outDist[o] = __uint_as_float(0x7f800000u);
        if (SHARDED) outPos[o] = 0xffffffffu;
      }
    }
    pqt_count_ties(&counters[3], ties);
  }
}


// ===================================================================================================
// Fused stage a7 + a8 for 128 < k <= 4096 (the reference front-end's own call is queryKNN(.., 4096): tool_query.cpp:155,
// PerturbationProTree.cu:8187-8218): one WORKGROUP (8 wavefronts) per query, the distances never leave the chip.
//   lane = one candidate (reference association, p ascending: bit-exact like pqt_rs_query); a key (f32 key << 32 | visiting
//   position) that beats tau is appended to ONE key array in LDS (capacity kcap >= 2k); when the array could overflow, an
//   exact block-wide radix select (range-adaptive 8-bit digits over the unique u64 keys, as pqt_wave_kth_u64) finds the k-th
//   key, the array is compacted to the k smallest and tau drops to it; at the end the <= k survivors are sorted by the block
//   bitonic network in LDS.  Replaces the staged pqt_k_rerank -> candDist in HBM -> pqt_k_select pair for these k.
// LDS: [coarse LP*C1*C1*4 when it fits] + LP*C1*4 + kcap*8 + 1088 bytes.
// ===================================================================================================
#define PQT_RSB_NT 512
template <int NT>
__device__ __forceinline__ uint64_t pqt_block_kth_u64(const uint64_t* keys, uint32_t cnt, uint32_t kth, uint32_t* hist /*256*/,
                                                       unsigned long long* slot /*4, 8-byte aligned*/) {
  const uint32_t tid = threadIdx.x;
  if (tid == 0) { slot[0] = ~0ull; slot[1] = 0; }
  __syncthreads();
  uint64_t mn = ~0ull, mx = 0;
  for (uint32_t i = tid; i < cnt; i += NT) { const uint64_t v = keys[i]; mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
  if (mn != ~0ull || mx != 0) { atomicMin(&slot[0], (unsigned long long)mn); atomicMax(&slot[1], (unsigned long long)mx); }
  __syncthreads();
  uint64_t lo = slot[0], hi = slot[1];
  uint64_t tau = lo;
  for (int pass = 0; pass < 9; ++pass) {
    const uint64_t range = hi - lo;
    if (range == 0) { tau = lo; break; }
    const int msb = 63 - __builtin_clzll(range);
    const uint32_t sh = msb > 7 ? (uint32_t)(msb - 7) : 0u;
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    for (uint32_t i = tid; i < cnt; i += NT) {
      const uint64_t v = keys[i];
      if (v >= lo && v <= hi) atomicAdd(&hist[(uint32_t)((v - lo) >> sh)], 1u);
    }
    __syncthreads();
    if (tid < 64) {  // wavefront 0: 4 counters per lane + one wave scan, as in pqt_wave_kth_u64
      const uint4 h = reinterpret_cast<const uint4*>(hist)[tid];
      const uint32_t sum = h.x + h.y + h.z + h.w;
      const uint32_t incl = pqt_wave_incl_scan(sum);
      uint32_t before = incl - sum;
      if (before < kth && kth <= incl) {
        uint32_t b = tid * 4, m = h.x;
        if (before + h.x < kth) { before += h.x; b += 1; m = h.y;
          if (before + h.y < kth) { before += h.y; b += 1; m = h.z;
            if (before + h.z < kth) { before += h.z; b += 1; m = h.w; } } }
        slot[2] = ((unsigned long long)b << 32) | m;
        slot[3] = before;
      }
    }
    __syncthreads();
    const uint32_t b = (uint32_t)(slot[2] >> 32), m = (uint32_t)slot[2];
    kth -= (uint32_t)slot[3];
    lo = lo + ((uint64_t)b << sh);
    const uint64_t top = lo + ((1ull << sh) - 1ull);
    hi = top < hi ? top : hi;
    __syncthreads();
    if (m == 1) {
      for (uint32_t i = tid; i < cnt; i += NT) { const uint64_t v = keys[i]; if (v >= lo && v <= hi) slot[2] = v; }
      __syncthreads();
      tau = slot[2];
      break;
    }
  }
  __syncthreads();
  return tau;
}

template <bool COARSE_LDS, bool SHARDED, int VEC /* 4: 16-byte code reads (LP % 4 == 0), 1: scalar */>
__global__ __launch_bounds__(PQT_RSB_NT) void pqt_k_rerank_select_big(
    const uint32_t* __restrict__ codes /* bin-ordered */, const uint32_t* __restrict__ ids, const float* __restrict__ qL1virt,
    const float* __restrict__ coarse, const uint32_t* __restrict__ cand, const uint32_t* __restrict__ candPos,
    const uint32_t* __restrict__ nLocal, uint64_t stride, uint32_t k, uint32_t kP2, uint32_t kcap, uint32_t nq, PqtDevParams prm,
    uint32_t* __restrict__ outIdx, float* __restrict__ outDist, uint32_t* __restrict__ outPos, unsigned long long* __restrict__ counters,
    const uint32_t* __restrict__ qlist, const uint32_t* __restrict__ qcount /* the queries to process (those pqt_k_rerank_sort_small left), or null = all */) {
  constexpr int NT = PQT_RSB_NT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const uint32_t C1 = prm.C1, LP = prm.LP;
  const uint32_t nCoarse = COARSE_LDS ? LP * C1 * C1 : 0;
  float* sCoarse = (float*)smem_raw;
  float* sVirt = sCoarse + nCoarse;
  uint64_t* sKeys = (uint64_t*)(sVirt + LP * C1);  // LP*C1 is even for every accepted shape (8-byte alignment)
  uint32_t* sHist = (uint32_t*)(sKeys + kcap);
  unsigned long long* sSlot = (unsigned long long*)(sHist + 256);
  uint32_t* sCnt = (uint32_t*)(sSlot + 4);
  const uint32_t tid = threadIdx.x;
  (void)kP2;
  // persistent workgroups: the LDS copy of the coarse table is loaded once and serves every query of this workgroup
  if (COARSE_LDS) for (uint32_t t = tid; t < nCoarse; t += NT) sCoarse[t] = coarse[t];
  const float* cz = COARSE_LDS ? sCoarse : coarse;
  const uint32_t nList = qlist ? *qcount : nq;
  for (uint32_t qe = blockIdx.x; qe < nList; qe += gridDim.x) {
  const uint32_t q = qlist ? qlist[qe] : qe;
  const uint32_t n = nLocal[q];
  const uint32_t* cid = cand + (size_t)q * stride;
  __syncthreads();  // the previous query's readers of sVirt / sKeys are done
  for (uint32_t t = tid; t < LP * C1; t += NT) sVirt[t] = qL1virt[(size_t)q * LP * C1 + t];
  if (tid == 0) { sCnt[0] = 0; sCnt[1] = 0; }
  __syncthreads();
  uint64_t tau = ~0ull;
  // keep the k smallest of the sCnt[0] keys held, tau = the k-th (block-wide, exact)
  auto shrink = [&]() {
    const uint32_t cnt = sCnt[0];
    tau = pqt_block_kth_u64<NT>(sKeys, cnt, k, sHist, sSlot);
    constexpr int PER = 16;  // kcap <= 8192 = NT * 16
    uint64_t mine[PER];
#pragma unroll
    for (int r = 0; r < PER; ++r) { const uint32_t i = tid + NT * r; mine[r] = i < cnt ? sKeys[i] : ~0ull; }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < PER; ++r) if (mine[r] <= tau) sKeys[atomicAdd(&sCnt[1], 1u)] = mine[r];
    __syncthreads();
    if (tid == 0) { sCnt[0] = sCnt[1]; sCnt[1] = 0; }
    __syncthreads();
  };
  for (uint32_t base = 0; base < n; base += NT) {
    const uint32_t j = base + tid;
    if (j < n) {
      const uint32_t pos = cid[j];
      const uint32_t* row = codes + (size_t)pos * LP;
      float acc = 0.f;
      if (VEC == 4) {
        const uint4* row4 = reinterpret_cast<const uint4*>(row);
        for (uint32_t p4 = 0; p4 < LP / 4; ++p4) {
          const uint4 v = row4[p4];
          const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const uint32_t p = p4 * 4 + u;
            const uint32_t A = w[u] & 0xffu, B = (w[u] >> 8) & 0xffu;
            const float lam = pqt_lambda_decode(w[u] >> 16);
            acc = acc + pqt_extract_distance(sVirt[p * C1 + B], sVirt[p * C1 + A], cz[((size_t)p * C1 + A) * C1 + B], lam);
          }
        }
      } else {
        for (uint32_t p = 0; p < LP; ++p) {
          const uint32_t w = row[p];
          const uint32_t A = w & 0xffu, B = (w >> 8) & 0xffu;
          const float lam = pqt_lambda_decode(w >> 16);
          acc = acc + pqt_extract_distance(sVirt[p * C1 + B], sVirt[p * C1 + A], cz[((size_t)p * C1 + A) * C1 + B], lam);
        }
      }
      const uint64_t key = ((uint64_t)pqt_f2key(acc) << 32) | j;
      if (key < tau) sKeys[atomicAdd(&sCnt[0], 1u)] = key;
    }
    __syncthreads();
    if (sCnt[0] + NT > kcap) shrink();  // uniform: sCnt[0] is read after the barrier by every thread
  }
  if (sCnt[0] > k) shrink();
  const uint32_t have = sCnt[0];
  uint32_t sortN = 2;  // the network runs over the keys really held (lists are often far shorter than k)
  while (sortN < have) sortN <<= 1;
  for (uint32_t i = have + tid; i < sortN; i += NT) sKeys[i] = ~0ull;
  __syncthreads();
  pqt_bitonic_sort_u64<NT>(sKeys, sortN);
  const uint32_t kk = n < k ? n : k;
  uint32_t ties = 0;
  for (uint32_t i = tid; i < k; i += NT) {
    const size_t o = (size_t)q * k + i;
    if (i < kk) {
      const uint64_t key = sKeys[i];
      const uint32_t j = (uint32_t)key;
      outIdx[o] = ids[cid[j]];
      outDist[o] = pqt_key2f((uint32_t)(key >> 32));
      if (SHARDED) outPos[o] = candPos[(size_t)q * stride + j];
      if (i + 1 < kk && (uint32_t)(sKeys[i + 1] >> 32) == (uint32_t)(key >> 32)) ++ties;
    } else {
      outIdx[o] = 0xffffffffu;
      outDist[o] = __uint_as_float(0x7f800000u);
      if (SHARDED) outPos[o] = 0xffffffffu;
    }
  }
  pqt_count_ties(&counters[3], ties);
  }
}

// ===================================================================================================
// 128 < k <= 4096 with SHORT candidate lists (n <= 1024: every list of the reference front-end's own call, queryKNN(.., 4096)
// at maxBins = 4096, holds a few hundred candidates): one WAVEFRONT per query, no tau filter, no select -- every candidate is
// evaluated (reference association, p ascending: bit-exact), the <= 1024 keys (f32 key << 32 | visiting position) are sorted
// by the in-register network of the smallest fitting size (R = 2, 4, 8, 16 keys per lane) and the n results + padding are
// written.  pqt_k_rerank_select_big spent 27 us per query on such lists (one 8-wavefront workgroup per CU, a barrier-separated
// chain of ~6 exposed memory round trips); here NW independent wavefronts per workgroup share the LDS copy of the coarse table
// and overlap each other's round trips.  Queries with longer lists are appended to bigList for pqt_k_rerank_select_big.
// LDS: [coarse LP*C1*C1*4 when it fits] + NW * LP*C1*4 + 16 bytes.
// ===================================================================================================
#define PQT_RSS_MAXN 1024
// (few instantiations on purpose -- the sorting networks dominate the compile time: C1 and the sharded outputs are run-time.)
// Key slots are 32 bits: phase 1 stores only the distance key, in the slot of the candidate's visiting position; the network works on
// (distance key << 32 | position) in registers; the results leave through the same slots in two rounds (positions, then distance keys),
// read back in coalesced order.  Half the LDS per wavefront of the 64-bit slots of round 3: 12 wavefronts (3 per SIMD) instead of 8 around
// the 64 KB coarse table for lists of <= 1024 candidates, 8 (2 per SIMD) instead of 4 for the second pass.
// MAXN = 2048, LIST = true is the second pass: the lists of 1025..2048 candidates the first pass set aside (A.qlist / A.qcount); each half is
// sorted by the 16-keys-per-lane network and the halves are joined by one bitonic merge level with 32 keys per lane (the second half read
// back in reverse).  What is longer still goes to bigList.
// A.padDone != 0: the padding behind the results is written by pqt_k_pad_rows on another stream (the rows are 4096 slots of which a few
// hundred are filled: 27 KB of stores per query that the compute wavefronts need not issue).
template <int NW, int LPV, bool COARSE_LDS, int MAXN = PQT_RSS_MAXN, bool LIST = false>
__global__ __launch_bounds__(NW * 64) void pqt_k_rerank_sort_small(const PqtRsArgs A, uint32_t* __restrict__ bigList, uint32_t* __restrict__ bigCount) {
  static_assert(MAXN == 1024 || MAXN == 2048, "key slots per wavefront");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr uint32_t LP = LPV * 4;
  const uint32_t C1 = A.prm.C1;
  const bool C1P2 = C1 > 1 && (C1 & (C1 - 1)) == 0;  // uniform
  const uint32_t c1sh = C1P2 ? (uint32_t)__builtin_ctz(C1) : 0u;
  const bool SHARDED = A.outPos != nullptr;
  const uint32_t nCoarse = COARSE_LDS ? LP * C1 * C1 : 0;
  float* sCoarse = (float*)smem_raw;
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  uint32_t* sK32 = (uint32_t*)(smem_raw + (size_t)nCoarse * 4) + (size_t)wave * MAXN;
  float* sVirt = (float*)(smem_raw + (size_t)nCoarse * 4 + (size_t)NW * MAXN * 4) + (size_t)wave * LP * C1;
  uint32_t* sTicket = reinterpret_cast<uint32_t*>(smem_raw + (size_t)nCoarse * 4 + (size_t)NW * MAXN * 4 + (size_t)NW * LP * C1 * 4);
  if (threadIdx.x == 0) sTicket[0] = 0;
  if (COARSE_LDS) for (uint32_t t = threadIdx.x; t < nCoarse; t += NW * 64) sCoarse[t] = A.coarse[t];
  __syncthreads();
  const float* cz = COARSE_LDS ? sCoarse : A.coarse;
  const uint32_t G = gridDim.x, k = A.k;
  const uint32_t nWork = LIST ? *A.qcount : A.qn;
  const uint32_t L = blockIdx.x < nWork ? (nWork - blockIdx.x + G - 1) / G : 0u;  // this workgroup's queries (list entries): b, b + G, ...
  uint32_t tiesAcc = 0;
  for (;;) {
    uint32_t t = 0;
    if (lane == 0) t = atomicAdd(sTicket, 1u);
    t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
    if (t >= L) break;
    const uint32_t q = LIST ? A.qlist[blockIdx.x + t * G] : blockIdx.x + t * G;
    const uint32_t n = A.nLocal[q];
    // (no `continue` in this loop: with one, hipcc 7.2 emitted a self-branch spinning on the loop-invariant `n <= 1024` mask for
    // the long-list path -- a hang on the first query with more than 1024 candidates)
    if (n > (uint32_t)MAXN) {  // long list: the next pass / the block-wide kernel takes it
      if (lane == 0) bigList[atomicAdd(bigCount, 1u)] = q;
    } else {
    if ((LP * C1) % 4 == 0) {
      const float4* src4 = reinterpret_cast<const float4*>(A.qL1virt + (size_t)q * LP * C1);
      float4* dst4 = reinterpret_cast<float4*>(sVirt);
      for (uint32_t i = lane; i < LP * C1 / 4; i += 64) dst4[i] = src4[i];
    } else {
      for (uint32_t i = lane; i < LP * C1; i += 64) sVirt[i] = A.qL1virt[(size_t)q * LP * C1 + i];
    }
    __builtin_amdgcn_wave_barrier();
    const uint32_t* cid = A.cand + (size_t)q * A.stride;
    // ---- phase 1: every candidate's distance key (reference association, p ascending: bit-exact) -> the slot of its visiting position
    constexpr int U = 4;  // candidates per lane whose rows are in flight together
    if (A.dbg & 1u) { for (uint32_t j = lane; j < n; j += 64) sK32[j] = n - j; }
    else
    for (uint32_t base = 0; base < n; base += 64 * U) {
      uint32_t pos[U];
#pragma unroll
      for (int u = 0; u < U; ++u) { const uint32_t j = base + u * 64 + lane; pos[u] = j < n ? cid[j] : 0u; if ((A.dbg & 8u) && pos[u] >= A.nIds) pos[u] = 0; }
      uint4 rows[U][LPV];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint4* row4 = reinterpret_cast<const uint4*>(A.codes + (size_t)pos[u] * LP);
#pragma unroll
        for (int v = 0; v < LPV; ++v) rows[u][v] = row4[v];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t j = base + u * 64 + lane;
        float acc = 0.f;
#pragma unroll
        for (int v = 0; v < LPV; ++v) {
          const uint32_t w[4] = {rows[u][v].x, rows[u][v].y, rows[u][v].z, rows[u][v].w};
#pragma unroll
          for (int x = 0; x < 4; ++x) {
            const uint32_t p = v * 4 + x;
            const uint32_t Aa = w[x] & 0xffu, Bb = (w[x] >> 8) & 0xffu;
            const float lam = __builtin_fmaf((float)(w[x] >> 16), 8.f / 65536.f, -4.f);  // == pqt_lambda_decode: the product is exact
            const uint32_t pv = C1P2 ? (p << c1sh) : p * C1;
            const float sb = sVirt[pv + Aa], sa = sVirt[pv + Bb];
            const float sc = cz[C1P2 ? (((pv + Aa) << c1sh) + Bb) : ((pv + Aa) * C1 + Bb)];
            acc = acc + pqt_extract_distance(sa, sb, sc, lam);
          }
        }
        if (j < n) sK32[j] = pqt_f2key(acc);
      }
    }
    __builtin_amdgcn_wave_barrier();
    const uint32_t kk = n < k ? n : k;
    // results leave in two rounds through the 32-bit slots: round 0 = visiting positions -> ids (+ global positions), round 1 = distances
    auto emit = [&](const int round) {
      __builtin_amdgcn_wave_barrier();
      if (A.dbg & 4u) return;
      if (round == 0) {
        for (uint32_t i0 = 0; i0 < kk; i0 += 256) {
          uint32_t sp[4], gp[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const uint32_t i = i0 + u * 64 + lane;
            uint32_t jj = i < kk ? sK32[i] : 0u;
            if ((A.dbg & 8u) && jj >= n) jj = 0;
            sp[u] = i < kk ? cid[jj] : 0u;
            gp[u] = (SHARDED && i < kk) ? A.candPos[(size_t)q * A.stride + jj] : 0xffffffffu;
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) { const uint32_t i = i0 + u * 64 + lane; if ((A.dbg & 8u) && sp[u] >= A.nIds) sp[u] = 0; sp[u] = i < kk ? A.ids[sp[u]] : 0xffffffffu; }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const uint32_t i = i0 + u * 64 + lane;
            if (i < kk) { const size_t o = (size_t)q * k + i; A.outIdx[o] = sp[u]; if (SHARDED) A.outPos[o] = gp[u]; }
          }
        }
      } else {
        for (uint32_t i = lane; i < kk; i += 64) {
          const uint32_t dk = sK32[i];
          if (i + 1 < kk && sK32[i + 1] == dk) ++tiesAcc;
          A.outDist[(size_t)q * k + i] = pqt_key2f(dk);
        }
      }
      __builtin_amdgcn_wave_barrier();
    };
    // sorts the first 64 * R slots (padded with +inf keys behind n) and emits them
    auto sortEmit = [&](auto rTag) {
      constexpr int R = decltype(rTag)::value;
      uint64_t key[R];
#pragma unroll
      for (int r = 0; r < R; ++r) { const uint32_t e = lane * R + r; key[r] = e < n ? (((uint64_t)sK32[e] << 32) | e) : ~0ull; }
      __builtin_amdgcn_wave_barrier();
      if (!(A.dbg & 2u)) pqt_wave_sort_u64<R>(key);
#pragma unroll
      for (int r = 0; r < R; ++r) sK32[lane * R + r] = (uint32_t)key[r];
      emit(0);
#pragma unroll
      for (int r = 0; r < R; ++r) sK32[lane * R + r] = (uint32_t)(key[r] >> 32);
      emit(1);
    };
    if (n <= 128) sortEmit(std::integral_constant<int, 2>{});
    else if (n <= 256) sortEmit(std::integral_constant<int, 4>{});
    else if (n <= 512) sortEmit(std::integral_constant<int, 8>{});
    else if (n <= 1024) sortEmit(std::integral_constant<int, 16>{});
    else if constexpr (MAXN > 1024) {
      // two sorted halves (16 keys per lane each), then ONE merge level over [first half ascending | second half descending]
      uint64_t ka[16], kb[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const uint32_t e = lane * 16 + r;
        ka[r] = ((uint64_t)sK32[e] << 32) | e;
        kb[r] = 1024u + e < n ? (((uint64_t)sK32[1024u + e] << 32) | (1024u + e)) : ~0ull;
      }
      __builtin_amdgcn_wave_barrier();
      pqt_wave_sort_u64<16>(ka);
      pqt_wave_sort_u64<16>(kb);
      uint64_t key[32];
      auto bit = [](const uint32_t e) -> uint32_t { return e < 1024u ? e : 3071u - e; };
#pragma unroll
      for (int r = 0; r < 16; ++r) { sK32[lane * 16 + r] = (uint32_t)(ka[r] >> 32); sK32[1024 + lane * 16 + r] = (uint32_t)(kb[r] >> 32); }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int r = 0; r < 32; ++r) key[r] = (uint64_t)sK32[bit(lane * 32 + r)] << 32;
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int r = 0; r < 16; ++r) { sK32[lane * 16 + r] = (uint32_t)ka[r]; sK32[1024 + lane * 16 + r] = (uint32_t)kb[r]; }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int r = 0; r < 32; ++r) key[r] |= sK32[bit(lane * 32 + r)];
      __builtin_amdgcn_wave_barrier();
      pqt_sort_merge<32, 2048, 1024>(key, (int)lane);
#pragma unroll
      for (int r = 0; r < 32; ++r) sK32[lane * 32 + r] = (uint32_t)key[r];
      emit(0);
#pragma unroll
      for (int r = 0; r < 32; ++r) sK32[lane * 32 + r] = (uint32_t)(key[r] >> 32);
      emit(1);
    }
    if (!A.padDone) {
      for (uint32_t i = ((A.dbg & 4u) ? 0u : kk) + lane; i < k; i += 64) {
        const size_t o = (size_t)q * k + i;
        A.outIdx[o] = 0xffffffffu;
        A.outDist[o] = __uint_as_float(0x7f800000u);
        if (SHARDED) A.outPos[o] = 0xffffffffu;
      }
    }
    }
    __builtin_amdgcn_wave_barrier();
  }
  pqt_count_ties(&A.counters[3], tiesAcc);
}

// padding of the result rows [min(n, k), k) of every query: ids 0xffffffff, distances +inf (positions 0xffffffff); launched on a side
// stream beside the k > 128 rerank kernels, which then write only the entries that exist
#ifdef PQT_MAIN_TU
__global__ __launch_bounds__(256) void pqt_k_pad_rows(const uint32_t* __restrict__ nLocal, uint32_t qn, uint32_t k, uint32_t* __restrict__ outIdx,
                                                       float* __restrict__ outDist, uint32_t* __restrict__ outPos) {
  const uint32_t q = blockIdx.x;
  if (q >= qn) return;
  const uint32_t n = nLocal[q] < k ? nLocal[q] : k;
  for (uint32_t i = n + threadIdx.x; i < k; i += 256) {
    const size_t o = (size_t)q * k + i;
    outIdx[o] = 0xffffffffu;
    outDist[o] = __uint_as_float(0x7f800000u);
    if (outPos) outPos[o] = 0xffffffffu;
  }
}
#endif  // PQT_MAIN_TU

// opt-in "adc_bias" mode: bias[pos] = sum_p (l*l*c - l*c), c = coarse[p][A][B], of the row at position pos of the
// bin-ordered store, summed in p order (f32, separate multiply and add).  lane = one row.
#ifdef PQT_MAIN_TU
__global__ __launch_bounds__(256) void pqt_k_adc_bias(const uint32_t* __restrict__ codesBin, uint64_t nIds, const float* __restrict__ coarse,
                                                       PqtDevParams prm, float* __restrict__ bias) {
  const uint64_t pos = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (pos >= nIds) return;
  const uint32_t* row = codesBin + pos * prm.LP;
  float s = 0.f;
  for (uint32_t p = 0; p < prm.LP; ++p) {
    const uint32_t w = row[p];
    const uint32_t A = w & 0xffu, B = (w >> 8) & 0xffu;
    const float l = pqt_lambda_decode(w >> 16);
    const float c = coarse[((size_t)p * prm.C1 + A) * prm.C1 + B];
    s = s + (l * l * c - l * c);
  }
  bias[pos] = s;
}
#endif  // PQT_MAIN_TU

// X-code copy of the bin-ordered line store for the exact rerank with the coarse table in LDS (pqt_rs_query XC): every 4-byte code
// {u8 A; u8 B; u16 lambda} (cpu_version/helper.hpp:39-90) becomes {u16 A*4*C1 + B*4; u16 lambda} -- the byte offset of
// coarse[p][A][B] inside the part's [C1][C1] table, from which the two L1virt offsets A*4 and B*4 are one bit-field operation each.
#ifdef PQT_MAIN_TU
__global__ __launch_bounds__(256) void pqt_k_xcode(const uint32_t* __restrict__ in, uint64_t nWords, uint32_t c1Shift, uint32_t* __restrict__ out) {
  const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= nWords) return;
  const uint32_t w = in[t];
  const uint32_t A = w & 0xffu, B = (w >> 8) & 0xffu;
  out[t] = (w & 0xffff0000u) | (A << (2 + c1Shift)) | (B << 2);
}
#endif  // PQT_MAIN_TU

// group-major copy of the bin-ordered line store for pqt_k_rerank_select_wg: out[g][pos][x] = in[pos][g*G + x]
#ifdef PQT_MAIN_TU
__global__ __launch_bounds__(256) void pqt_k_group_major(const uint32_t* __restrict__ in, uint64_t nIds, uint32_t LP, uint32_t G,
                                                          uint32_t* __restrict__ out) {
  const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;  // one word each; consecutive t -> consecutive output words
  if (t >= nIds * LP) return;
  const uint64_t g = t / (nIds * G), r = t % (nIds * G), pos = r / G, x = r % G;
  out[t] = in[pos * LP + g * G + x];
}
#endif  // PQT_MAIN_TU

// ===================================================================================================
// Self-checks of the sort / select / scan primitives on the patterns the reference's own self-tests use
// (pqt/bitonicSort.cuh:213-252: sortTestLarge sorts the values N - tid with payload tid and expects payload[tid] == N - tid - 1;
// scanTestLarge scans a vector of ones and expects the exclusive prefix tid), N = 1024 / 2048 / 4096.  One launch per (primitive, N);
// out[N] receives what the reference checks.  mode: 0 = in-register wave network pqt_wave_sort_u64<R> (N = 64 R <= 2048, one wavefront),
// 1 = block-wide bitonic network in LDS (pqt_bitonic_sort_u64), 2 = wave radix select pqt_wave_kth_u64 (out[j] = payload of the
// (j+1)-th smallest key, every rank j asked for in turn), 3 = block radix select pqt_block_kth_u64, 4 = wave scan pqt_wave_incl_scan of
// ones (exclusive = inclusive - 1, per wavefront of the block, offset by the wave's base), 5 = block scan pqt_block_excl_scan of N ones
// held N / 256 per thread.  Keys are built like the product's: pqt_f2key(value) << 32 | payload.
// ===================================================================================================
#ifdef PQT_MAIN_TU
template <int R>
__device__ __forceinline__ void pqt_dbg_wave_sort(uint32_t n, uint32_t* out) {
  const uint32_t lane = threadIdx.x & 63;
  uint64_t key[R];
#pragma unroll
  for (int r = 0; r < R; ++r) { const uint32_t e = lane * R + r; key[r] = ((uint64_t)pqt_f2key((float)(n - e)) << 32) | e; }
  pqt_wave_sort_u64<R>(key);
#pragma unroll
  for (int r = 0; r < R; ++r) out[lane * R + r] = (uint32_t)key[r];
}
template <int R>
__device__ __forceinline__ void pqt_dbg_wave_kth(uint32_t n, uint32_t* out, uint32_t* hist) {
  const uint32_t lane = threadIdx.x & 63;
  uint64_t key[R];
#pragma unroll
  for (int r = 0; r < R; ++r) { const uint32_t e = r * 64 + lane; key[r] = ((uint64_t)pqt_f2key((float)(n - e)) << 32) | e; }
  for (uint32_t j = 0; j < n; ++j) {
    const uint64_t tau = pqt_wave_kth_u64<R>(key, j + 1, hist);
    if (lane == 0) out[j] = (uint32_t)tau;
  }
}
__global__ __launch_bounds__(256) void pqt_k_debug_sortscan(uint32_t mode, uint32_t n, uint32_t* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  uint64_t* sKey = reinterpret_cast<uint64_t*>(smem_raw);
  const uint32_t tid = threadIdx.x;
  if (mode == 0) {
    if (tid >= 64) return;
    if (n == 64) pqt_dbg_wave_sort<1>(n, out); else if (n == 128) pqt_dbg_wave_sort<2>(n, out); else if (n == 256) pqt_dbg_wave_sort<4>(n, out);
    else if (n == 512) pqt_dbg_wave_sort<8>(n, out); else if (n == 1024) pqt_dbg_wave_sort<16>(n, out); else if (n == 2048) pqt_dbg_wave_sort<32>(n, out);
  } else if (mode == 1) {
    for (uint32_t e = tid; e < n; e += 256) sKey[e] = ((uint64_t)pqt_f2key((float)(n - e)) << 32) | e;
    __syncthreads();
    pqt_bitonic_sort_u64<256>(sKey, n);
    for (uint32_t e = tid; e < n; e += 256) out[e] = (uint32_t)sKey[e];
  } else if (mode == 2) {
    if (tid >= 64) return;
    uint32_t* hist = reinterpret_cast<uint32_t*>(smem_raw);  // 256 u32 + 4 u64
    if (n == 512) pqt_dbg_wave_kth<8>(n, out, hist); else if (n == 1024) pqt_dbg_wave_kth<16>(n, out, hist);
  } else if (mode == 3) {
    uint32_t* hist = reinterpret_cast<uint32_t*>(sKey + n);
    unsigned long long* slot = reinterpret_cast<unsigned long long*>(hist + 256);
    for (uint32_t e = tid; e < n; e += 256) sKey[e] = ((uint64_t)pqt_f2key((float)(n - e)) << 32) | e;
    __syncthreads();
    for (uint32_t j = 0; j < n; j += (n >> 6)) {  // 64 ranks spread over the range (each costs a few block-wide passes)
      const uint64_t tau = pqt_block_kth_u64<256>(sKey, n, j + 1, hist, slot);
      if (tid == 0) out[j] = (uint32_t)tau;
      __syncthreads();
    }
  } else if (mode == 4) {
    for (uint32_t e = tid; e < n; e += 256) {  // every wavefront scans its 64 ones; element e sits in lane e % 64 of round e / 64
      const uint32_t incl = pqt_wave_incl_scan(1u);
      out[e] = (e & ~63u) + incl - 1u;
    }
  } else if (mode == 6) {
    // the lane exchanges of the sorting networks: out[LMi * 64 + lane] = value held by lane ^ LM, LM = 1, 2, 4, 8, 16, 32 (first wavefront)
    if (tid >= 64) return;
    const uint32_t v = 0x9e3779b9u * (tid + 1u);
    out[0 * 64 + tid] = pqt_lane_xor_u32<1>(v); out[1 * 64 + tid] = pqt_lane_xor_u32<2>(v); out[2 * 64 + tid] = pqt_lane_xor_u32<4>(v);
    out[3 * 64 + tid] = pqt_lane_xor_u32<8>(v); out[4 * 64 + tid] = pqt_lane_xor_u32<16>(v); out[5 * 64 + tid] = pqt_lane_xor_u32<32>(v);
    out[6 * 64 + tid] = pqt_lane_down1_u32(v);  // value of lane + 1 (lane 63: its own)
    { uint32_t t[1] = {(0x9e3779b9u * (tid + 1u)) | 1u}; pqt_wave_sort_u32<1>(t); out[7 * 64 + tid] = t[0]; }  // the u32 network on 64 distinct keys
  } else if (mode == 8) {
    // four 64-key row sorts in one pass (pqt_row_sort64_u32): row p sorts the distinct keys hash(64 p + e) | 1, e = 4 (lane & 15) + r
    if (tid >= 64) return;
    uint32_t k[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) k[r] = (0x9e3779b9u * (4u * tid + (uint32_t)r + 1u)) | 1u;
    pqt_row_sort64_u32(k);
#pragma unroll
    for (int r = 0; r < 4; ++r) out[4 * tid + r] = k[r];
  } else if (mode == 7) {
    // inclusive wave scan of irregular values (v = (lane * 2654435761) >> 24), first wavefront
    if (tid >= 64) return;
    out[tid] = pqt_wave_incl_scan((tid * 2654435761u) >> 24);
  } else if (mode == 5) {
    uint32_t* sPart = reinterpret_cast<uint32_t*>(smem_raw);
    const uint32_t per = n / 256;  // ones held per thread (contiguous elements tid*per ..)
    uint32_t total = 0;
    const uint32_t base = pqt_block_excl_scan<256>(per, sPart, &total);
    for (uint32_t i = 0; i < per; ++i) out[tid * per + i] = base + i;
    if (tid == 0) out[n] = total;
  }
}
#endif  // PQT_MAIN_TU


// ===================================================================================================
// Compaction of padded result rows for the host hand-over (pqt_compact_results): the front-end's queryKNN(.., 4096) returns rows of
// 4096 slots of which a few hundred are filled; only the filled prefix of every row should cross PCIe.
//   pqt_k_row_offsets: offsets[q] = sum_{i<q} min(count[i], k), offsets[qn] = total (one workgroup, chunks of 1024 rows)
//   pqt_k_compact_rows: one wavefront per row copies its first min(count, k) (id, distance) pairs to the packed arrays
// ===================================================================================================
#ifdef PQT_MAIN_TU
__global__ __launch_bounds__(1024) void pqt_k_row_offsets(const uint32_t* __restrict__ count, uint32_t qn, uint32_t k, uint32_t* __restrict__ offsets) {
  __shared__ uint32_t sPart[1024 / 64 + 1];
  uint32_t run = 0;
  for (uint32_t b = 0; b < qn; b += 1024) {
    const uint32_t q = b + threadIdx.x;
    const uint32_t c = q < qn ? (count[q] < k ? count[q] : k) : 0u;
    uint32_t total = 0;
    const uint32_t ex = pqt_block_excl_scan<1024>(c, sPart, &total);
    if (q < qn) offsets[q] = run + ex;
    run += total;
  }
  if (threadIdx.x == 0) offsets[qn] = run;
}
__global__ __launch_bounds__(256) void pqt_k_compact_rows(const uint32_t* __restrict__ idx, const float* __restrict__ dist, const uint32_t* __restrict__ offsets,
                                                            uint32_t qn, uint32_t k, uint32_t* __restrict__ outIdx, float* __restrict__ outDist) {
  const uint32_t q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (q >= qn) return;
  const uint32_t o = offsets[q], c = offsets[q + 1] - o;
  for (uint32_t i = lane; i < c; i += 64) { outIdx[o + i] = idx[(size_t)q * k + i]; outDist[o + i] = dist[(size_t)q * k + i]; }
}
#endif  // PQT_MAIN_TU
