// pqt_shared_rows.h -- the shared-row pass of the filtered rerank (pqt_rs_query MODE 2 with bin runs; BASELINE configs[2]/[3] shape).
//
// What it removes.  At this shape (C1*C2 = 2^12, P = 4) the reference's uint32 bin id drops part 3 altogether ((C1*C2)^3 = 2^36 wraps to 0,
// treequantizer.hpp:45-49,572; cpu_version/matlab/readme.md:26): the tuples of a query that differ only there name the SAME bin, the
// reference visits it once per tuple and its candidate list holds the bin's vectors once per visit (rerankVectors appends whatever
// _bins[globIdx] holds, :462-476).  Measured on the 100 M bench index at (20000, 500): 229.9 M candidate reads per 10 k-query batch are
// 133.9 M distinct (query, row) pairs (1.72 visits per pair) over 64.0 M distinct rows (2.09 queries per row); a query's candidates sit
// in one to three bins of 4 k - 25 k rows.  The wave-per-query rerank reads and evaluates every visit; it runs at the stream rate of the
// device, so only fewer bytes make it faster.
//
// How.  The filter distance d1 (MODE 2: sum_p (b + l (a - b)) + bias[row], pqt_rs_query) depends on (query, row) only:
//   1. pqt_k_sr_visits   one wavefront per query: its runs (= bin visits) that name the same first store row are one PAIR (query, bin);
//                        the pair registers itself in a per-batch open-addressing table keyed by that row (count per bin = queries that
//                        include it), no same-address atomics across bins.
//   2. pqt_k_sr_scan / _scan2   items = sum over bins of tiles(bin) x queries(bin): exclusive scan over the table slots (two levels, no atomics).
//   3. pqt_k_sr_items    every pair writes its items (query, run, tile) at base(bin) + tile * queries(bin) + its rank: the items of one
//                        tile of one bin are ADJACENT, whichever queries they belong to.
//   4. pqt_k_sr_adc      one wavefront per item: L1virt of the query into LDS (8 KB), the tile's rows streamed from the group-major store
//                        (1 KB per plane per wave instruction), d1 stored at EVERY visiting position of the pair (candDist[q][j0_visit + row]).
//                        Adjacent items run in adjacent wavefronts of a workgroup at the same time: the tile's rows come from DRAM once and
//                        from the CU's vector cache / the XCD's L2 for the other queries.
//   5. pqt_k_rerank_select<.., PRE>   the wave-per-query selection as before, reading d1 instead of rows: tau filter, exact radix select of the
//                        256 smallest keys (d1, visiting position), reference association for the band, sort, ids.
// d1 is computed by the same instruction sequence as in pqt_rs_query (same association, -ffp-contract=off): the keys, the band, the
// fallbacks and the results are the same bits as without the pass (tests/test_gpu_parity.py runs both).
// Queries whose runs did not fit the hand-over (nRuns = 0xffffffff) or whose pairs did not fit the table keep evaluating their rows in step 5.
#pragma once
#include "pqt_kernels.h"

#ifndef PQT_SR_TILE
#define PQT_SR_TILE 2048u   // rows per item
#endif

struct PqtSrArgs {
  const unsigned long long* runs; const uint32_t* nRuns; const uint32_t* nLocal; uint32_t qn;
  uint32_t* keys; uint32_t* cnt; uint32_t* len; uint32_t* base; uint32_t slotBits;  // per-batch bin table: 2^slotBits slots (keys, cnt zeroed / 0xff-filled per batch)
  uint32_t* pairSlot; uint32_t* pairIdx;   // [qn][64]: table slot of the pair run r of query q is the canonical visit of (0xffffffff: none), its rank among the bin's queries
  uint32_t* preOk;                         // [qn]
  uint32_t* blockSum; uint32_t nBlocks;    // per 1024 slots: items of the block, then (after _scan2) their exclusive prefix
  uint32_t* total;                         // [0] items of the batch
  unsigned long long* items; uint64_t itemCap;  // query | run << 32 | tile << 40
  const uint4* codesGrp4; uint64_t nIds; const float* bias; const float* qL1virt; float* dist; uint64_t stride;
};

// 1. one wavefront per query
__global__ __launch_bounds__(256) void pqt_k_sr_visits(const PqtSrArgs A) {
  const uint32_t lane = threadIdx.x & 63, q = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= A.qn) return;
  const uint32_t m = A.nRuns[q], n = A.nLocal[q];
  const bool listed = m != 0xffffffffu && m <= 64u;
  const bool act = listed && lane < m;
  const unsigned long long rr = act ? A.runs[(size_t)q * PQT_RUNCAP + lane] : ~0ull;
  const uint32_t j0 = (uint32_t)rr, s = (uint32_t)(rr >> 32);
  const uint32_t jn = __shfl_down(j0, 1, 64);
  const uint32_t len = act ? ((lane + 1 < m ? jn : n) - j0) : 0u;
  bool canon = act && len > 0u;
  const uint32_t mu = listed ? (uint32_t)__builtin_amdgcn_readfirstlane((int)m) : 0u;
  for (uint32_t i = 0; i < mu; ++i) {  // canonical visit of a bin = its first run in visiting order
    const uint32_t si = (uint32_t)__builtin_amdgcn_readlane((int)s, (int)i);
    if (si == s && i < lane) canon = false;
  }
  uint32_t slot = 0xffffffffu, rank = 0;
  if (canon) {
    const uint32_t mask = (1u << A.slotBits) - 1u;
    uint32_t h = (s * 2654435761u) >> (32u - A.slotBits);
    for (uint32_t probe = 0; probe < 128u; ++probe) {
      const uint32_t prev = atomicCAS(&A.keys[h], 0xffffffffu, s);
      if (prev == 0xffffffffu || prev == s) { slot = h; break; }
      h = (h + 1u) & mask;
    }
    if (slot != 0xffffffffu) { rank = atomicAdd(&A.cnt[slot], 1u); A.len[slot] = len; }
  }
  const bool failed = canon && slot == 0xffffffffu;
  const bool ok = listed && __ballot(failed) == 0ull;
  A.pairSlot[(size_t)q * 64 + lane] = slot;
  A.pairIdx[(size_t)q * 64 + lane] = rank;
  if (lane == 0) A.preOk[q] = ok ? 1u : 0u;
}

// 2a. items per table slot, exclusive scan inside blocks of 1024 slots
__global__ __launch_bounds__(1024) void pqt_k_sr_scan(const PqtSrArgs A) {
  __shared__ uint32_t sWave[16];
  const uint32_t slot = blockIdx.x * 1024 + threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t c = A.cnt[slot];
  const uint32_t need = c ? c * ((A.len[slot] + PQT_SR_TILE - 1u) / PQT_SR_TILE) : 0u;
  const uint32_t incl = pqt_wave_incl_scan(need);
  if (lane == 63) sWave[wave] = incl;
  __syncthreads();
  uint32_t off = 0;
  for (uint32_t w = 0; w < wave; ++w) off += sWave[w];
  A.base[slot] = off + incl - need;
  if (threadIdx.x == 1023) A.blockSum[blockIdx.x] = off + incl;
}
// 2b. exclusive scan of the block sums (one workgroup), total
__global__ __launch_bounds__(1024) void pqt_k_sr_scan2(const PqtSrArgs A) {
  __shared__ uint32_t sWave[16];
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t per = (A.nBlocks + 1023u) / 1024u;
  uint32_t sum = 0;
  for (uint32_t i = 0; i < per; ++i) { const uint32_t b = threadIdx.x * per + i; if (b < A.nBlocks) sum += A.blockSum[b]; }
  const uint32_t incl = pqt_wave_incl_scan(sum);
  if (lane == 63) sWave[wave] = incl;
  __syncthreads();
  uint32_t off = 0;
  for (uint32_t w = 0; w < wave; ++w) off += sWave[w];
  uint32_t run = off + incl - sum;
  for (uint32_t i = 0; i < per; ++i) {
    const uint32_t b = threadIdx.x * per + i;
    if (b < A.nBlocks) { const uint32_t v = A.blockSum[b]; A.blockSum[b] = run; run += v; }
  }
  if (threadIdx.x == 1023) A.total[0] = off + incl;
}

// 3. one thread per (query, run): the items of a canonical visit
__global__ __launch_bounds__(256) void pqt_k_sr_items(const PqtSrArgs A) {
  const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= (uint64_t)A.qn * 64) return;
  const uint32_t slot = A.pairSlot[t];
  if (slot == 0xffffffffu) return;
  const uint32_t q = (uint32_t)(t >> 6), r = (uint32_t)(t & 63u);
  // (a query whose other pairs did not fit the table still owns the slots it got: its items are written and evaluated, the selection
  // kernel ignores them -- no holes in the item list)
  const uint32_t c = A.cnt[slot], rank = A.pairIdx[t];
  const uint32_t tiles = (A.len[slot] + PQT_SR_TILE - 1u) / PQT_SR_TILE;
  const uint64_t b = (uint64_t)A.base[slot] + A.blockSum[slot >> 10];
  for (uint32_t ti = 0; ti < tiles; ++ti) {
    const uint64_t o = b + (uint64_t)ti * c + rank;
    if (o < A.itemCap) A.items[o] = (unsigned long long)q | ((unsigned long long)r << 32) | ((unsigned long long)ti << 40);
  }
}

// 4. filter distances of one tile of one bin for one query, by one wavefront
template <int NW, int LPV, int C1M, int U>
__global__ __launch_bounds__(NW * 64) void pqt_k_sr_adc(const PqtSrArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr uint32_t LP = LPV * 4, C1 = 1u << C1M;
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* const sVirt = reinterpret_cast<float*>(smem_raw) + (size_t)wave * LP * C1;
  uint64_t total = A.total[0];
  if (total > A.itemCap) total = A.itemCap;
  for (uint64_t i = (uint64_t)blockIdx.x * NW + wave; i < total; i += (uint64_t)gridDim.x * NW) {
    const unsigned long long it = A.items[i];
    const uint32_t q = (uint32_t)it, r = (uint32_t)(it >> 32) & 0xffu, ti = (uint32_t)(it >> 40);
    const uint32_t m = A.nRuns[q], n = A.nLocal[q];
    const unsigned long long rr = lane < m ? A.runs[(size_t)q * PQT_RUNCAP + lane] : ~0ull;
    // the query's table, requested whole before the first piece is stored
    constexpr uint32_t NV = LP * C1 / 4, IT = (NV + 63) / 64;
    {
      const float4* src4 = reinterpret_cast<const float4*>(A.qL1virt + (size_t)q * LP * C1);
      float4* dst4 = reinterpret_cast<float4*>(sVirt);
      float4 tmp[IT];
#pragma unroll
      for (uint32_t x = 0; x < IT; ++x) { const uint32_t t = lane + 64 * x; tmp[x] = src4[t < NV ? t : 0]; }
#pragma unroll
      for (uint32_t x = 0; x < IT; ++x) { const uint32_t t = lane + 64 * x; if (t < NV) dst4[t] = tmp[x]; }
    }
    const uint32_t j0 = (uint32_t)rr, s = (uint32_t)(rr >> 32);
    const uint32_t sr = (uint32_t)__builtin_amdgcn_readlane((int)s, (int)r), j0r = (uint32_t)__builtin_amdgcn_readlane((int)j0, (int)r);
    const uint32_t jn = r + 1 < m ? (uint32_t)__builtin_amdgcn_readlane((int)j0, (int)(r + 1)) : n;
    const uint32_t lenr = jn - j0r;
    const unsigned long long visits = __ballot(lane < m && s == sr);  // every visit of this bin by the query (run r is the first)
    const uint32_t row0 = ti * PQT_SR_TILE, row1 = lenr < row0 + PQT_SR_TILE ? lenr : row0 + PQT_SR_TILE;
    float* const drow = A.dist + (size_t)q * A.stride;
    __builtin_amdgcn_wave_barrier();
    for (uint32_t b = row0; b < row1; b += 64 * U) {
      uint4 rows[U][LPV];
      float rbias[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t o = b + u * 64 + lane;
        const size_t pos = (size_t)sr + (o < row1 ? o : row1 - 1u);
#pragma unroll
        for (int v = 0; v < LPV; ++v) rows[u][v] = A.codesGrp4[(size_t)v * A.nIds + pos];
        rbias[u] = A.bias[pos];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t o = b + u * 64 + lane;
        float acc = 0.f;
#pragma unroll
        for (int v = 0; v < LPV; ++v) {
          const uint32_t w[4] = {rows[u][v].x, rows[u][v].y, rows[u][v].z, rows[u][v].w};
#pragma unroll
          for (int x = 0; x < 4; x += 2) {  // the instruction sequence of pqt_rs_query MODE 2 (same association: same bits)
            pqt_f2 sb2, sa2, lam2;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const uint32_t p = v * 4 + x + h;
              const uint32_t ww = w[x + h];
              const uint32_t Aa = ww & 0xffu, Bb = (ww >> 8) & 0xffu;
              lam2[h] = (float)(ww >> 16);
              sb2[h] = sVirt[(p << C1M) + Aa];
              sa2[h] = sVirt[(p << C1M) + Bb];
            }
            const pqt_f2 kScale = {8.f / 65536.f, 8.f / 65536.f}, kOff = {-4.f, -4.f};
            lam2 = lam2 * kScale + kOff;
            const pqt_f2 d2 = sb2 + lam2 * (sa2 - sb2);
            acc = acc + d2[0];
            acc = acc + d2[1];
          }
        }
        acc = acc + rbias[u];
        unsigned long long vm = visits;
        while (vm) {  // uniform: one (coalesced) store per visit of the bin
          const uint32_t d = (uint32_t)__builtin_ctzll(vm);
          vm &= vm - 1ull;
          const uint32_t jd = (uint32_t)__builtin_amdgcn_readlane((int)j0, (int)d);
          if (o < row1) drow[jd + o] = acc;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();  // the next item overwrites the table
  }
}
