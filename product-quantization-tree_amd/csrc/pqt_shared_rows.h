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

#ifndef PQT_SR_ABL
#define PQT_SR_ABL 0   // development ablations of pqt_k_sr_adc (results wrong): 1 no stores, 2 no table look-ups, 4 cache-resident rows, 8 one visit only
#endif
#ifndef PQT_SR_TILE
#define PQT_SR_TILE 2048u   // rows per item
#endif

#ifndef PQT_SR_OCC
#define PQT_SR_OCC 4   // wavefronts per SIMD pqt_k_sr_adc is compiled for (register budget 512 / PQT_SR_OCC)
#endif
#ifndef PQT_SR_DB
#define PQT_SR_DB 1    // 1: two register sets of rows (next batch in flight under the current one's arithmetic), 0: one
#endif
#ifndef PQT_SR_B64
#define PQT_SR_B64 0   // 1: table look-ups as ds_read_b64 of entry pairs (experiment, see pqt_k_sr_adc)
#endif
#ifndef PQT_SR_QC
#define PQT_SR_QC 8   // queries of a bin evaluated per item against one copy of the rows (their L1virt tables side by side in LDS: 8 KB each)
#endif

struct PqtSrArgs {
  const unsigned long long* runs; const uint32_t* nRuns; const uint32_t* nLocal; uint32_t qn;
  uint32_t* keys; uint32_t* cnt; uint32_t* len; uint32_t* base; uint32_t* lbase; uint32_t slotBits;  // per-batch bin table: 2^slotBits slots (keys 0xff-filled, cnt zeroed per batch)
  uint32_t* pairSlot; uint32_t* pairIdx;   // [qn][64]: table slot of the pair run r of query q is the canonical visit of (0xffffffff: none), its rank among the bin's queries
  uint32_t* preOk; float* qmax;            // [qn] covered by the pass; largest entry of the query's L1virt table (the selection's error bound)
  uint32_t* blockSum; uint32_t nBlocks;    // [nBlocks][2] per 1024 slots: items / list entries of the block, then (after _scan2) their exclusive prefixes
  uint32_t* total;                         // [0] items of the batch, [1] list entries, [2] != 0: they did not fit itemCap / listCap (every query is handed back)
  uint32_t maxProbes;                      // table probes before a pair gives up (128; tests: fewer)
  unsigned long long* stat;                // pqt_k_sr_stats only: 8 counters, zeroed before its launch
  unsigned long long* items; uint64_t itemCap;  // slot | tile << 32 | chunk << 48
  unsigned long long* binList; uint64_t listCap;  // the queries of a bin, side by side: query | run << 32
  const uint4* codesGrp4; uint64_t nIds; const float* bias; const float* qL1virt; float* dist; uint64_t stride; uint32_t tableFloats /* LP * C1 */;
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
    for (uint32_t probe = 0; probe < A.maxProbes; ++probe) {
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
  // largest entry of the query's table (pqt_rs_query MODE 2 takes it from its LDS copy; the selection kernel of this pass keeps none)
  float qmax = 0.f;
  {
    const uint32_t nv = A.tableFloats / 4;
    const float4* src4 = reinterpret_cast<const float4*>(A.qL1virt + (size_t)q * A.tableFloats);
    for (uint32_t t = lane; t < nv; t += 64) {
      const float4 v = src4[t];
      const float m01 = v.x > v.y ? v.x : v.y, m23 = v.z > v.w ? v.z : v.w, mm = m01 > m23 ? m01 : m23;
      qmax = mm > qmax ? mm : qmax;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { const float o = __shfl_xor(qmax, d, 64); qmax = o > qmax ? o : qmax; }
  }
  if (lane == 0) { A.preOk[q] = ok ? 1u : 0u; A.qmax[q] = qmax; }
}

// 2a. items and list entries per table slot, exclusive scans inside blocks of 1024 slots
__global__ __launch_bounds__(1024) void pqt_k_sr_scan(const PqtSrArgs A) {
  __shared__ uint32_t sWave[2][16];
  const uint32_t slot = blockIdx.x * 1024 + threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t c = A.cnt[slot];
  const uint32_t need = c ? ((c + PQT_SR_QC - 1u) / PQT_SR_QC) * ((A.len[slot] + PQT_SR_TILE - 1u) / PQT_SR_TILE) : 0u;
  const uint32_t incl = pqt_wave_incl_scan(need), inclL = pqt_wave_incl_scan(c);
  if (lane == 63) { sWave[0][wave] = incl; sWave[1][wave] = inclL; }
  __syncthreads();
  uint32_t off = 0, offL = 0;
  for (uint32_t w = 0; w < wave; ++w) { off += sWave[0][w]; offL += sWave[1][w]; }
  A.base[slot] = off + incl - need;
  A.lbase[slot] = offL + inclL - c;
  if (threadIdx.x == 1023) { A.blockSum[2 * blockIdx.x] = off + incl; A.blockSum[2 * blockIdx.x + 1] = offL + inclL; }
}
// 2b. exclusive scans of the block sums (one workgroup), total
__global__ __launch_bounds__(1024) void pqt_k_sr_scan2(const PqtSrArgs A) {
  __shared__ uint32_t sWave[2][16];
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t per = (A.nBlocks + 1023u) / 1024u;
  uint32_t sum = 0, sumL = 0;
  for (uint32_t i = 0; i < per; ++i) { const uint32_t b = threadIdx.x * per + i; if (b < A.nBlocks) { sum += A.blockSum[2 * b]; sumL += A.blockSum[2 * b + 1]; } }
  const uint32_t incl = pqt_wave_incl_scan(sum), inclL = pqt_wave_incl_scan(sumL);
  if (lane == 63) { sWave[0][wave] = incl; sWave[1][wave] = inclL; }
  __syncthreads();
  uint32_t off = 0, offL = 0;
  for (uint32_t w = 0; w < wave; ++w) { off += sWave[0][w]; offL += sWave[1][w]; }
  uint32_t run = off + incl - sum, runL = offL + inclL - sumL;
  for (uint32_t i = 0; i < per; ++i) {
    const uint32_t b = threadIdx.x * per + i;
    if (b < A.nBlocks) { const uint32_t v = A.blockSum[2 * b], vL = A.blockSum[2 * b + 1]; A.blockSum[2 * b] = run; A.blockSum[2 * b + 1] = runL; run += v; runL += vL; }
  }
  if (threadIdx.x == 1023) {
    // The capacities are worst-case bounds (a run is a whole bin: at most stride / TILE + 64 items and 64 list entries per query), so the
    // flag cannot rise today; if a later change of the run emission or of the caps ever breaks that, the guards of pqt_k_sr_items /
    // pqt_k_sr_adc would drop work silently -- instead every query of the batch is handed back to the exact kernels
    const uint32_t items = off + incl, entries = offL + inclL;
    A.total[0] = items; A.total[1] = entries;
    A.total[2] = ((uint64_t)items > A.itemCap || (uint64_t)entries > A.listCap) ? 1u : 0u;
  }
}

// 3. one thread per (query, run): a canonical visit enters its bin's query list; the first query of every chunk of PQT_SR_QC writes the chunk's items
__global__ __launch_bounds__(256) void pqt_k_sr_items(const PqtSrArgs A) {
  const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= (uint64_t)A.qn * 64) return;
  const uint32_t slot = A.pairSlot[t];
  if (slot == 0xffffffffu) return;
  const uint32_t q = (uint32_t)(t >> 6), r = (uint32_t)(t & 63u);
  // (a query whose other pairs did not fit the table still owns the slots it got: it is listed and evaluated, the selection kernel
  // ignores what was written for it -- no holes in the lists)
  const uint32_t c = A.cnt[slot], rank = A.pairIdx[t];
  const uint64_t lo = (uint64_t)A.lbase[slot] + A.blockSum[2 * (slot >> 10) + 1] + rank;
  if (lo < A.listCap) A.binList[lo] = (unsigned long long)q | ((unsigned long long)r << 32);
  if (rank % PQT_SR_QC) return;
  const uint32_t chunks = (c + PQT_SR_QC - 1u) / PQT_SR_QC, chunk = rank / PQT_SR_QC;
  const uint32_t tiles = (A.len[slot] + PQT_SR_TILE - 1u) / PQT_SR_TILE;
  const uint64_t b = (uint64_t)A.base[slot] + A.blockSum[2 * (slot >> 10)];
  for (uint32_t ti = 0; ti < tiles; ++ti) {
    const uint64_t o = b + (uint64_t)ti * chunks + chunk;
    if (o < A.itemCap) A.items[o] = (unsigned long long)slot | ((unsigned long long)ti << 32) | ((unsigned long long)chunk << 48);
  }
}

// 4. filter distances of one tile of one bin for up to PQT_SR_QC of its queries, by one workgroup: every row is read ONCE (a wavefront holds
// 64 of them in registers while it walks the queries' tables)
template <int NW, int LPV, int C1M>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(PQT_SR_OCC, PQT_SR_OCC))) void pqt_k_sr_adc(const PqtSrArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr uint32_t LP = LPV * 4, C1 = 1u << C1M, QC = PQT_SR_QC;
  constexpr uint32_t TB = LP * C1 * 4;  // bytes of one table
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  uint32_t* const sJ0 = reinterpret_cast<uint32_t*>(smem_raw + (size_t)QC * TB);  // [QC][64] first visiting positions of the query's visits of this bin
  uint32_t* const sNv = sJ0 + QC * 64;                                               // [QC] their number
  uint32_t* const sQ = sNv + QC;                                                     // [QC] the query
  uint64_t total = A.total[0];
  if (total > A.itemCap) total = A.itemCap;
  for (uint64_t i = blockIdx.x; i < total; i += gridDim.x) {
    const unsigned long long it = A.items[i];
    const uint32_t slot = (uint32_t)it, ti = (uint32_t)(it >> 32) & 0xffffu, chunk = (uint32_t)(it >> 48);
    const uint32_t sr = A.keys[slot], lenr = A.len[slot], c = A.cnt[slot];
    const uint64_t lo = (uint64_t)A.lbase[slot] + A.blockSum[2 * (slot >> 10) + 1] + (uint64_t)chunk * QC;
    const uint32_t nq = c - chunk * QC < QC ? c - chunk * QC : QC;
    __syncthreads();  // the previous item's tables are no longer read
    // the queries' tables, side by side
    for (uint32_t e = wave; e < nq; e += NW) {
      const unsigned long long ent = A.binList[lo + e];
      const uint32_t q = (uint32_t)ent;
      const float4* src4 = reinterpret_cast<const float4*>(A.qL1virt + (size_t)q * LP * C1);
      float4* dst4 = reinterpret_cast<float4*>(smem_raw + (size_t)e * TB);
      constexpr uint32_t NV = LP * C1 / 4, IT = (NV + 63) / 64;
      float4 tmp[IT];
#pragma unroll
      for (uint32_t x = 0; x < IT; ++x) { const uint32_t t = lane + 64 * x; tmp[x] = src4[t < NV ? t : 0]; }
      // every visit of this bin by the query (the listed run is the first): first visiting positions, compacted
      const uint32_t m = A.nRuns[q];
      const unsigned long long rr = lane < m ? A.runs[(size_t)q * PQT_RUNCAP + lane] : ~0ull;
      const bool vis = lane < m && (uint32_t)(rr >> 32) == sr;
      uint32_t nv;
      const uint32_t rk = pqt_ballot_rank(vis, &nv);
      if (vis) sJ0[e * 64 + rk] = (uint32_t)rr;
      if (lane == 0) { sNv[e] = nv; sQ[e] = q; }
#pragma unroll
      for (uint32_t x = 0; x < IT; ++x) { const uint32_t t = lane + 64 * x; if (t < NV) dst4[t] = tmp[x]; }
    }
    __syncthreads();
    const uint32_t row0 = ti * PQT_SR_TILE, row1 = lenr < row0 + PQT_SR_TILE ? lenr : row0 + PQT_SR_TILE;
    // the rows of this wavefront's next batch are requested before the current one is evaluated (two register sets, alternating).
    // (Without the scheduling barriers the compiler sinks the requests between the terms of the other set to save registers.)
    auto request = [&](uint4 (&rows)[LPV], float& rbias, const uint32_t b) {
      const uint32_t o = b + lane;
      size_t pos = (size_t)sr + (o < row1 ? o : row1 - 1u);
      if (PQT_SR_ABL & 4) pos &= 4095u;
#pragma unroll
      for (int v = 0; v < LPV; ++v) rows[v] = A.codesGrp4[(size_t)v * A.nIds + pos];
      rbias = A.bias[pos];
    };
    auto evaluate = [&](const uint4 (&rows)[LPV], const float rbias, const uint32_t b) {
      const uint32_t o = b + lane;
      for (uint32_t e = 0; e < nq; ++e) {
        const float* const sVirt = reinterpret_cast<const float*>(smem_raw + (size_t)e * TB);
        float acc = 0.f;
#pragma unroll
        for (int v = 0; v < LPV; ++v) {
          const uint32_t w[4] = {rows[v].x, rows[v].y, rows[v].z, rows[v].w};
#pragma unroll
          for (int x = 0; x < 4; x += 2) {  // the instruction sequence of pqt_rs_query MODE 2 (same association: same bits)
            pqt_f2 sb2, sa2, lam2;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const uint32_t p = v * 4 + x + h;
              const uint32_t ww = w[x + h];
              const uint32_t Aa = ww & 0xffu, Bb = (ww >> 8) & 0xffu;
              lam2[h] = (float)(ww >> 16);
              if (PQT_SR_ABL & 2) { sb2[h] = __uint_as_float(Aa); sa2[h] = __uint_as_float(Bb); } else {
#if PQT_SR_B64
              // experiment (default off, not yet measured): a 64-entry table row read as 32 eight-byte pairs -- ds_read_b64 banks are
              // (a / 4) mod 64, conflict free for any 64 lanes inside one 256-byte row, where ds_read_b32 ((a / 4) mod 32) is 2-way; one
              // select per look-up more
              const float2 pa = reinterpret_cast<const float2*>(sVirt)[(p << (C1M - 1)) + (Aa >> 1)], pb = reinterpret_cast<const float2*>(sVirt)[(p << (C1M - 1)) + (Bb >> 1)];
              sb2[h] = (Aa & 1u) ? pa.y : pa.x;
              sa2[h] = (Bb & 1u) ? pb.y : pb.x;
#else
              sb2[h] = sVirt[(p << C1M) + Aa];
              sa2[h] = sVirt[(p << C1M) + Bb];
#endif
              }
            }
            const pqt_f2 kScale = {8.f / 65536.f, 8.f / 65536.f}, kOff = {-4.f, -4.f};
            lam2 = lam2 * kScale + kOff;
            const pqt_f2 d2 = sb2 + lam2 * (sa2 - sb2);
            acc = acc + d2[0];
            acc = acc + d2[1];
          }
        }
        acc = acc + rbias;
        float* const drow = A.dist + (size_t)sQ[e] * A.stride;
        const uint32_t nv = (PQT_SR_ABL & 8) ? 1u : sNv[e];
        if (PQT_SR_ABL & 1) { if (acc == 12345.678f) drow[o] = acc; }
        else
        for (uint32_t vi = 0; vi < nv; ++vi) {  // uniform: one (coalesced) store per visit of the bin
          const uint32_t jd = sJ0[e * 64 + vi];
          if (o < row1) drow[jd + o] = acc;
        }
      }
    };
    const uint32_t first = row0 + wave * 64;
#if PQT_SR_DB
    uint4 rowsA[LPV], rowsB[LPV];
    float biasA = 0.f, biasB = 0.f;
    if (first < row1) request(rowsA, biasA, first);
    for (uint32_t b = first; b < row1; b += 2 * NW * 64) {
      const bool haveB = b + NW * 64 < row1;
      if (haveB) request(rowsB, biasB, b + NW * 64);
      __builtin_amdgcn_sched_barrier(0);
      evaluate(rowsA, biasA, b);
      __builtin_amdgcn_sched_barrier(0);
      if (!haveB) break;
      if (b + 2 * NW * 64 < row1) request(rowsA, biasA, b + 2 * NW * 64);
      __builtin_amdgcn_sched_barrier(0);
      evaluate(rowsB, biasB, b + NW * 64);
      __builtin_amdgcn_sched_barrier(0);
    }
#else
    uint4 rowsA[LPV];
    float biasA = 0.f;
    for (uint32_t b = first; b < row1; b += NW * 64) {
      request(rowsA, biasA, b);
      __builtin_amdgcn_sched_barrier(0);
      evaluate(rowsA, biasA, b);
      __builtin_amdgcn_sched_barrier(0);
    }
#endif
  }
}

// statistics of the pass for the batch just prepared (on request only: option "sr_stats"; bench.py prices the roofline of pqt_k_sr_adc with
// them).  stat[0] bins in the table, [1] (query, bin) pairs, [2] distinct rows (sum of the bins' lengths), [3] rows the evaluating kernel
// reads (a bin's rows once per chunk of PQT_SR_QC queries), [4] items, [5] queries with candidates that the pass does not cover, [6] distances
// it writes for the covered ones (their candidates: one per visit and row), [7] capacity flag (total[2]).
__global__ __launch_bounds__(1024) void pqt_k_sr_stats(const PqtSrArgs A) {
  const uint32_t t = blockIdx.x * 1024 + threadIdx.x, lane = threadIdx.x & 63;
  unsigned long long v[7] = {0, 0, 0, 0, 0, 0, 0};
  if (t < (1u << A.slotBits)) {
    const uint32_t c = A.cnt[t];
    if (c) {
      const uint32_t len = A.len[t], chunks = (c + PQT_SR_QC - 1u) / PQT_SR_QC;
      v[0] = 1; v[1] = c; v[2] = len; v[3] = (unsigned long long)len * chunks; v[4] = (unsigned long long)chunks * ((len + PQT_SR_TILE - 1u) / PQT_SR_TILE);
    }
  }
  if (t < A.qn) {
    const uint32_t n = A.nLocal[t];
    if (n && !A.preOk[t]) v[5] = 1;
    if (A.preOk[t]) v[6] = n;
  }
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    unsigned long long x = v[i];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) x += __shfl_xor(x, d, 64);
    if (lane == 0 && x) atomicAdd(&A.stat[i], x);
  }
  if (t == 0) A.stat[7] = A.total[2];
}

// 4'. the same launch with the tables of TWO queries interleaved and the row decode outside the query loop (round 6; option "sr_kernel" = 2).
// What pqt_k_sr_adc spends per (64 rows, query): 64 ds_read_b32 (2-way bank conflicts inside a 64-entry row: 4 LDS cycles each) and ~280
// VALU instructions, of which the extraction of A, B, lambda from the code words and the two look-up addresses per term (~190) do not depend
// on the query.  Here
//   * the tables of queries (2k, 2k + 1) of a chunk sit interleaved in LDS as float2 [LP][C1]{q_2k, q_2k+1}: ONE ds_read_b64 at byte
//     (p * C1 + A) * 8 returns the entry of both queries -- half the LDS instructions per (row, query), and their addresses are the decoded
//     byte offset A * 8 of the row plus an immediate (pair * 16 KB + p * 512 B < 64 KB: the DS offset field), no address arithmetic at all;
//   * a 16-byte piece of a row (4 terms) is decoded ONCE (A * 8, B * 8, lambda: 12 registers) and then walks the <= 4 query pairs, each pair
//     advancing its two accumulators by the piece's 4 terms with packed f32 operations over the PAIR (v_pk_add / v_pk_mul: the two queries
//     are the two halves) -- 4 packed instructions per term and pair instead of ~8.6 per term and query.
// Per query the operations and their order are those of pqt_k_sr_adc (t = a - b; u = lambda * t; d = b + u; acc += d for p = 0 .. LP - 1, then
// + bias; separate IEEE multiply and add, -ffp-contract=off): the same bits.  A chunk with an odd number of queries evaluates its last table
// against a copy of itself (results of the second half dropped).
template <int NW, int LPV, int C1M>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(PQT_SR_OCC, PQT_SR_OCC))) void pqt_k_sr_adc2(const PqtSrArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr uint32_t LP = LPV * 4, C1 = 1u << C1M, QC = 8, NP = QC / 2;
  constexpr uint32_t TF = LP * C1;      // floats of one table
  constexpr uint32_t PB = TF * 8;       // bytes of one interleaved pair of tables
  static_assert(NW == 8, "one wavefront per query of a chunk sets the chunk up; two wavefronts interleave one pair of tables");
  static_assert((NP - 1) * PB + (LP - 1) * C1 * 8 + (C1 - 1) * 8 + 8 <= 65536, "look-up offsets must fit the DS instructions' 16-bit offset field");
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  typedef __attribute__((address_space(3))) const pqt_f2* lds_f2p;
  if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) const unsigned char*)smem_raw != 0u) __builtin_trap();  // (uniform; cannot happen: the kernel has no static LDS)
  uint32_t* const sJ0 = reinterpret_cast<uint32_t*>(smem_raw + (size_t)NP * PB);  // [QC][64] first visiting positions of the query's visits of this bin
  uint32_t* const sNv = sJ0 + QC * 64;                                               // [QC] their number
  uint32_t* const sQ = sNv + QC;                                                     // [QC] the query
  uint64_t total = A.total[0];
  if (total > A.itemCap) total = A.itemCap;
  for (uint64_t i = blockIdx.x; i < total; i += gridDim.x) {
    const unsigned long long it = A.items[i];
    const uint32_t slot = (uint32_t)it, ti = (uint32_t)(it >> 32) & 0xffffu, chunk = (uint32_t)(it >> 48);
    const uint32_t sr = A.keys[slot], lenr = A.len[slot], c = A.cnt[slot];
    const uint64_t lo = (uint64_t)A.lbase[slot] + A.blockSum[2 * (slot >> 10) + 1] + (uint64_t)chunk * QC;
    const uint32_t nq = c - chunk * QC < QC ? c - chunk * QC : QC;
    const uint32_t npairs = (nq + 1u) >> 1;
    __syncthreads();  // the previous item's tables are no longer read
    if (wave < nq) {
      // wavefront e: the visits of this bin by query e (the listed run is the first) -- first visiting positions, compacted
      const uint32_t e = wave;
      const uint32_t q = (uint32_t)A.binList[lo + e];
      const uint32_t m = A.nRuns[q];
      const unsigned long long rr = lane < m ? A.runs[(size_t)q * PQT_RUNCAP + lane] : ~0ull;
      const bool vis = lane < m && (uint32_t)(rr >> 32) == sr;
      uint32_t nv;
      const uint32_t rk = pqt_ballot_rank(vis, &nv);
      if (vis) sJ0[e * 64 + rk] = (uint32_t)rr;
      if (lane == 0) { sNv[e] = nv; sQ[e] = q; }
    }
    if ((wave >> 1) < npairs) {
      // wavefronts 2k and 2k + 1: one half each of the interleaved tables of pair k (4-byte loads side by side, 8-byte stores: conflict free)
      const uint32_t k = wave >> 1, h = wave & 1u;
      const uint32_t q0 = (uint32_t)A.binList[lo + 2 * k];
      const uint32_t q1 = (2 * k + 1 < nq) ? (uint32_t)A.binList[lo + 2 * k + 1] : q0;
      const float* const t0 = A.qL1virt + (size_t)q0 * TF + h * (TF / 2);
      const float* const t1 = A.qL1virt + (size_t)q1 * TF + h * (TF / 2);
      float2* const dst = reinterpret_cast<float2*>(smem_raw + (size_t)k * PB) + h * (TF / 2);
      constexpr uint32_t IT = TF / 2 / 64;
      float v0[IT], v1[IT];
#pragma unroll
      for (uint32_t x = 0; x < IT; ++x) { v0[x] = t0[x * 64 + lane]; v1[x] = t1[x * 64 + lane]; }
#pragma unroll
      for (uint32_t x = 0; x < IT; ++x) dst[x * 64 + lane] = make_float2(v0[x], v1[x]);
    }
    __syncthreads();
    const uint32_t row0 = ti * PQT_SR_TILE, row1 = lenr < row0 + PQT_SR_TILE ? lenr : row0 + PQT_SR_TILE;
    auto request = [&](uint4 (&rows)[LPV], float& rbias, const uint32_t b) {
      const uint32_t o = b + lane;
      const size_t pos = (size_t)sr + (o < row1 ? o : row1 - 1u);
#pragma unroll
      for (int v = 0; v < LPV; ++v) rows[v] = A.codesGrp4[(size_t)v * A.nIds + pos];
      rbias = A.bias[pos];
    };
    auto evaluate = [&](const uint4 (&rows)[LPV], const float rbias, const uint32_t b) {
      const uint32_t o = b + lane;
      pqt_f2 acc[NP];
#pragma unroll
      for (uint32_t k = 0; k < NP; ++k) acc[k] = pqt_f2{0.f, 0.f};
#pragma unroll
      for (int v = 0; v < LPV; ++v) {
        const uint32_t w[4] = {rows[v].x, rows[v].y, rows[v].z, rows[v].w};
        // the piece's four terms, decoded once for every query of the chunk
        uint32_t offA[4], offB[4];
        pqt_f2 lam01, lam23;
        lam01[0] = (float)(w[0] >> 16); lam01[1] = (float)(w[1] >> 16); lam23[0] = (float)(w[2] >> 16); lam23[1] = (float)(w[3] >> 16);
        const pqt_f2 kScale = {8.f / 65536.f, 8.f / 65536.f}, kOff = {-4.f, -4.f};
        lam01 = lam01 * kScale + kOff;
        lam23 = lam23 * kScale + kOff;
        const float lam[4] = {lam01[0], lam01[1], lam23[0], lam23[1]};
#pragma unroll
        for (int x = 0; x < 4; ++x) { offA[x] = (w[x] & 0xffu) << 3; offB[x] = ((w[x] >> 8) & 0xffu) << 3; }
#pragma unroll
        for (uint32_t k = 0; k < NP; ++k) {
          if (k < npairs) {  // (uniform)
#pragma unroll
            for (int x = 0; x < 4; ++x) {
              const uint32_t partBase = k * PB + (uint32_t)(v * 4 + x) * (C1 * 8);  // compile-time: the DS offset field
              // absolute LDS addresses (the dynamic segment starts at 0: no static LDS in this kernel, checked once below) -- through smem_raw the
              // compiler adds the segment's link-time base to every address (one v_add per look-up, 18 % of this kernel's VALU instructions)
              const pqt_f2 sb = *(lds_f2p)(uintptr_t)(partBase + offA[x]);
              const pqt_f2 sa = *(lds_f2p)(uintptr_t)(partBase + offB[x]);
              const pqt_f2 l2 = {lam[x], lam[x]};
              const pqt_f2 d2 = sb + l2 * (sa - sb);
              acc[k] = acc[k] + d2;
            }
          }
        }
      }
      const pqt_f2 b2 = {rbias, rbias};
#pragma unroll
      for (uint32_t k = 0; k < NP; ++k) {
        if (k < npairs) {  // (uniform)
          const pqt_f2 r2 = acc[k] + b2;
#pragma unroll
          for (uint32_t hq = 0; hq < 2; ++hq) {
            const uint32_t e = 2 * k + hq;
            if (e < nq) {  // (uniform)
              float* const drow = A.dist + (size_t)sQ[e] * A.stride;
              const uint32_t nv = sNv[e];
              for (uint32_t vi = 0; vi < nv; ++vi) {  // uniform: one (coalesced) store per visit of the bin
                const uint32_t jd = sJ0[e * 64 + vi];
                if (o < row1) drow[jd + o] = r2[hq];
              }
            }
          }
        }
      }
    };
    const uint32_t first = row0 + wave * 64;
    uint4 rowsA[LPV], rowsB[LPV];
    float biasA = 0.f, biasB = 0.f;
    if (first < row1) request(rowsA, biasA, first);
    for (uint32_t b = first; b < row1; b += 2 * NW * 64) {
      const bool haveB = b + NW * 64 < row1;
      if (haveB) request(rowsB, biasB, b + NW * 64);
      __builtin_amdgcn_sched_barrier(0);
      evaluate(rowsA, biasA, b);
      __builtin_amdgcn_sched_barrier(0);
      if (!haveB) break;
      if (b + 2 * NW * 64 < row1) request(rowsA, biasA, b + 2 * NW * 64);
      __builtin_amdgcn_sched_barrier(0);
      evaluate(rowsB, biasB, b + NW * 64);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// 5a. the scan of the split selection: the <= 256 smallest filter keys (f32 key << 32 | visiting position) of query q out of the n distances
// pqt_k_sr_adc wrote, by one wavefront.  pqt_rs_query's batch loop spends ~75 instructions per 64 candidates (64-bit keys, two candidates per
// lane and round trip, its bookkeeping for rows it does not fetch here) and the selection launch was at the instruction-issue ceiling (37 k
// instructions per 23 k-candidate query); here a lane takes FOUR consecutive distances per 16-byte request (four requests in flight), rejects
// on the 32-bit key alone (anything above the current 256-th smallest key's high word cannot belong), and only the survivors get a 64-bit key
// and a slot.  Same result: every key below the final 256-th smallest is kept, the exact radix select and the final sort are pqt_rs_query's.
// sKeys: NSLOT 8-byte slots of this wavefront ([best <= 256 | pending]).
template <int NSLOT>
__device__ __forceinline__ void pqt_sr_scan_query(const PqtRsArgs& A, const uint32_t q, const uint32_t n, uint64_t* const sKeys) {
  const uint32_t lane = threadIdx.x & 63;
  constexpr uint32_t BESTN = 256, SLOTS = NSLOT;
  constexpr int RK = NSLOT / 64;
  static_assert(BESTN + 256 <= SLOTS / 2 + 256 && SLOTS >= 768, "a block of 256 appended keys must fit behind the best list and a half-full pending area");
  if (n && (A.preOk[q] == 0u || (A.preFlags && A.preFlags[2]))) {  // not covered by the pass (or the pass ran out of item / list space): handed back like a query whose near-tie band overflows
    if (lane == 0) { A.fbList[atomicAdd(A.fbCount, 1u)] = q; A.preCnt[q] = 0xffffffffu; }
    return;
  }
  const float* const row = A.preDist + (size_t)q * A.stride;  // 256-byte aligned (the stride is a multiple of 64 floats)
  uint32_t off0 = 0, npend = 0, tauHi = 0xffffffffu;
  auto flush = [&](const bool final) {
    uint32_t have = off0 + npend;
    if (have > BESTN) {
      uint64_t key[RK];
#pragma unroll
      for (int r = 0; r < RK; ++r) { const uint32_t e = r * 64 + lane; key[r] = (e < have) ? sKeys[e] : ~0ull; }
      __builtin_amdgcn_wave_barrier();
      const uint64_t tau = pqt_wave_kth_u64<RK>(key, BESTN, reinterpret_cast<uint32_t*>(sKeys + BESTN));
      tauHi = (uint32_t)(tau >> 32);
      uint32_t cnt = 0;
#pragma unroll
      for (int r = 0; r < RK; ++r) {
        uint32_t tot;
        const uint32_t rk = pqt_ballot_rank(key[r] <= tau, &tot);
        if (key[r] <= tau) sKeys[cnt + rk] = key[r];
        cnt += tot;
      }
      have = BESTN;
      __builtin_amdgcn_wave_barrier();
    }
    if (final) {
      uint64_t key[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) { const uint32_t e = lane * 4 + r; key[r] = (e < have) ? sKeys[e] : ~0ull; }
      pqt_wave_sort_u64<4>(key);
#pragma unroll
      for (int r = 0; r < 4; ++r) sKeys[lane * 4 + r] = key[r];
      __builtin_amdgcn_wave_barrier();
    }
    npend = 0;
    off0 = have;
  };
  constexpr int QD = 4;
  float4 qv[QD];
  const uint32_t lastQuad = n ? ((n - 1u) & ~3u) : 0u;
#pragma unroll
  for (int d = 0; d < QD; ++d) { const uint32_t j = ((uint32_t)d * 64u + lane) * 4u; qv[d] = *reinterpret_cast<const float4*>(row + (j < n ? j : lastQuad)); }
  for (uint32_t base = 0; base < n; base += 256) {
    const float4 v = qv[0];
#pragma unroll
    for (int d = 0; d + 1 < QD; ++d) qv[d] = qv[d + 1];
    {  // (unconditional, clamped: the compiler must be able to count the requests in flight)
      const uint32_t j = base + (uint32_t)QD * 256u + lane * 4u;
      qv[QD - 1] = *reinterpret_cast<const float4*>(row + (j < n ? j : lastQuad));
    }
    const float c4[4] = {v.x, v.y, v.z, v.w};
    const uint32_t j0 = base + lane * 4u;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const uint32_t k32 = pqt_f2key(c4[c]);
      const bool pass = j0 + c < n && k32 <= tauHi;
      const unsigned long long m = __ballot(pass);
      if (m) {  // (uniform)
        const uint32_t rk = (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (pass) sKeys[off0 + npend + rk] = ((uint64_t)k32 << 32) | (j0 + c);
        npend += (uint32_t)__popcll(m);
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (off0 + npend + 256 > SLOTS) flush(false);
  }
  flush(true);
#pragma unroll
  for (int r = 0; r < 4; ++r) { const uint32_t e = r * 64 + lane; if (e < off0) A.preKeys[(size_t)q * 256 + e] = sKeys[e]; }
  if (lane == 0) A.preCnt[q] = off0;
}

// 5a'. the scan over POSITION RANGES (round 6, opt-in: option "sr_scan_split" = 2 | 4, "sr_scan_depth" = 4 | 8).  Why: the scan launch moves
// 4 bytes per candidate at half the stream rate (0.26-0.31 ms for 0.92 GB).  One wavefront per query means 10 k items of very different
// length (20 k .. 45 k candidates) on 4096 resident wavefronts -- 2.4 items per wavefront, so the last third of the launch runs on a
// fraction of the device -- and each wavefront has 4 KB in flight.  Here an item is one of SEG position ranges of a query (multiples of 256
// candidates), QD requests of 16 bytes per lane are in flight, every range leaves its own <= 256 best keys, and pqt_k_sr_merge reduces the
// SEG lists of a query to the list the band launch expects (exact radix select of the 256 smallest of <= SEG * 256 unique keys, sorted).
// The final best list is the same SET of keys as the one-wavefront scan's (every key below the 256-th smallest of the query is among the 256
// smallest of its range), sorted the same way: same bits downstream.
template <int NSLOT, int QD>
__device__ __forceinline__ uint32_t pqt_sr_scan_range(const float* const row, const uint32_t jb, const uint32_t je, uint64_t* const sKeys) {
  const uint32_t lane = threadIdx.x & 63;
  constexpr uint32_t BESTN = 256, SLOTS = NSLOT;
  constexpr int RK = NSLOT / 64;
  static_assert(SLOTS >= 768, "a block of 256 appended keys must fit behind the best list and a half-full pending area");
  uint32_t off0 = 0, npend = 0, tauHi = 0xffffffffu;
  auto flush = [&](const bool final) {
    uint32_t have = off0 + npend;
    if (have > BESTN) {
      uint64_t key[RK];
#pragma unroll
      for (int r = 0; r < RK; ++r) { const uint32_t e = r * 64 + lane; key[r] = (e < have) ? sKeys[e] : ~0ull; }
      __builtin_amdgcn_wave_barrier();
      const uint64_t tau = pqt_wave_kth_u64<RK>(key, BESTN, reinterpret_cast<uint32_t*>(sKeys + BESTN));
      tauHi = (uint32_t)(tau >> 32);
      uint32_t cnt = 0;
#pragma unroll
      for (int r = 0; r < RK; ++r) {
        uint32_t tot;
        const uint32_t rk = pqt_ballot_rank(key[r] <= tau, &tot);
        if (key[r] <= tau) sKeys[cnt + rk] = key[r];
        cnt += tot;
      }
      have = BESTN;
      __builtin_amdgcn_wave_barrier();
    }
    (void)final;
    npend = 0;
    off0 = have;
  };
  if (jb >= je) return 0u;
  float4 qv[QD];
  const uint32_t lastQuad = (je - 1u) & ~3u;
#pragma unroll
  for (int d = 0; d < QD; ++d) { const uint32_t j = jb + ((uint32_t)d * 64u + lane) * 4u; qv[d] = *reinterpret_cast<const float4*>(row + (j < je ? j : lastQuad)); }
  for (uint32_t base = jb; base < je; base += 256) {
    const float4 v = qv[0];
#pragma unroll
    for (int d = 0; d + 1 < QD; ++d) qv[d] = qv[d + 1];
    {  // (unconditional, clamped: the compiler must be able to count the requests in flight)
      const uint32_t j = base + (uint32_t)QD * 256u + lane * 4u;
      qv[QD - 1] = *reinterpret_cast<const float4*>(row + (j < je ? j : lastQuad));
    }
    const float c4[4] = {v.x, v.y, v.z, v.w};
    const uint32_t j0 = base + lane * 4u;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const uint32_t k32 = pqt_f2key(c4[c]);
      const bool pass = j0 + c < je && k32 <= tauHi;
      const unsigned long long m = __ballot(pass);
      if (m) {  // (uniform)
        const uint32_t rk = (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (pass) sKeys[off0 + npend + rk] = ((uint64_t)k32 << 32) | (j0 + c);
        npend += (uint32_t)__popcll(m);
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (off0 + npend + 256 > SLOTS) flush(false);
  }
  flush(true);
  return off0;  // the <= 256 smallest keys of the range sit in sKeys[0 .. off0), unsorted
}

// item = (query, range): segKeys[(q * SEG + s) * 256 ..], segCnt[q * SEG + s]; range 0 of a query also says whether the query is handed back
// (preCnt[q] = 0xffffffff, fbList) or not (preCnt[q] = 0: pqt_k_sr_merge fills it in)
template <int NW, int SEG, int QD>
__global__ __launch_bounds__(NW * 64) void pqt_k_sr_scan_seg(const PqtRsArgs A, unsigned long long* const segKeys, uint32_t* const segCnt) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int NSLOT = 1024;
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  uint64_t* const sKeys = reinterpret_cast<uint64_t*>(smem_raw) + (size_t)wave * NSLOT;
  // (what the evaluating kernel does for the next call: statistics block and the schedule's registration block zeroed)
  if (blockIdx.x == 0 && threadIdx.x < 8 && A.zero8) A.zero8[threadIdx.x] = 0;
  if (blockIdx.x == 0 && A.poolNext) for (uint32_t t = threadIdx.x; t < 16u + 8u * PQT_SCHED_CLASSES; t += NW * 64) A.poolNext[t] = 0;
  const uint64_t items = (uint64_t)A.qn * SEG;
  for (uint64_t it = (uint64_t)blockIdx.x * NW + wave; it < items; it += (uint64_t)gridDim.x * NW) {
    const uint32_t q = (uint32_t)(it / SEG), sg = (uint32_t)(it % SEG);
    const uint32_t n = A.nLocal[q];
    const bool back = n && (A.preOk[q] == 0u || (A.preFlags && A.preFlags[2]));
    if (sg == 0 && lane == 0) {
      if (back) A.fbList[atomicAdd(A.fbCount, 1u)] = q;
      A.preCnt[q] = back ? 0xffffffffu : 0u;
    }
    uint32_t got = 0;
    if (!back && n) {
      const uint32_t per = ((n + (uint32_t)SEG * 256u - 1u) / ((uint32_t)SEG * 256u)) * 256u;
      const uint32_t jb = sg * per, je = n < jb + per ? n : jb + per;
      got = pqt_sr_scan_range<NSLOT, QD>(A.preDist + (size_t)q * A.stride, jb < n ? jb : n, je, sKeys);
#pragma unroll
      for (int r = 0; r < 4; ++r) { const uint32_t e = r * 64 + lane; if (e < got) segKeys[(size_t)it * 256 + e] = sKeys[e]; }
    }
    if (lane == 0) segCnt[it] = got;
    __builtin_amdgcn_wave_barrier();
  }
}

// the SEG best lists of a query -> its <= 256 smallest keys, ascending, in preKeys[q][256] / preCnt[q] (what the band launch reads)
template <int NW, int SEG>
__global__ __launch_bounds__(NW * 64) void pqt_k_sr_merge(const PqtRsArgs A, const unsigned long long* const segKeys, const uint32_t* const segCnt) {
  __shared__ __attribute__((aligned(16))) uint64_t sAll[NW][256 + 136];  // per wavefront: 256 compaction slots + the radix select's 1056 bytes
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t q = blockIdx.x * NW + wave;
  if (q >= A.qn) return;
  if (A.preCnt[q] == 0xffffffffu) return;  // handed back by the scan
  uint64_t* const sK = sAll[wave];
  constexpr int RK = SEG * 4;
  uint64_t key[RK];
  uint32_t have = 0;
#pragma unroll
  for (int sg = 0; sg < SEG; ++sg) {
    const uint32_t c = segCnt[(size_t)q * SEG + sg];
#pragma unroll
    for (int r = 0; r < 4; ++r) { const uint32_t e = r * 64 + lane; key[sg * 4 + r] = e < c ? segKeys[((size_t)q * SEG + sg) * 256 + e] : ~0ull; }
    have += c;
  }
  uint64_t tau = ~0ull - 1ull;
  if (have > 256u) tau = pqt_wave_kth_u64<RK>(key, 256u, reinterpret_cast<uint32_t*>(sK + 256));
  uint32_t cnt = 0;
#pragma unroll
  for (int r = 0; r < RK; ++r) {
    uint32_t tot;
    const bool keep = key[r] <= tau;  // (absent entries are ~0: never kept)
    const uint32_t rk = pqt_ballot_rank(keep, &tot);
    if (keep) sK[cnt + rk] = key[r];
    cnt += tot;
  }
  __builtin_amdgcn_wave_barrier();
  uint64_t k4[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { const uint32_t e = lane * 4 + r; k4[r] = e < cnt ? sK[e] : ~0ull; }
  pqt_wave_sort_u64<4>(k4);
#pragma unroll
  for (int r = 0; r < 4; ++r) { const uint32_t e = lane * 4 + r; if (e < cnt) A.preKeys[(size_t)q * 256 + e] = k4[r]; }
  if (lane == 0) A.preCnt[q] = cnt;
}

// ---- cooperative filter scan (round 6; opt-in: option "coop_rerank" = 1) ------------------------------------------
// The wave-per-query filter kernel (pqt_k_rerank_select MODE 2) keeps a private 8 KB copy of the query's table per wavefront: 12 wavefronts
// per CU (3 per SIMD), and at a few hundred to a few thousand candidates per query (range shards) a wavefront sits in s_waitcnt for half of
// its cycles with nothing else to issue (DESIGN_HISTORY round 4: "what would move it is a fourth wavefront per SIMD, which the 8 KB copy per
// wavefront rules out").  Here a workgroup is 8 query slots x 2 wavefronts: the two wavefronts of a slot work on the SAME query around ONE
// table copy (8 x 8 KB + 16 x (4 KB of key slots + run list) = 140 KB: 16 wavefronts per CU, 4 per SIMD), take alternate batches of its
// candidates and each keep the 256 smallest filter keys of their share; pqt_k_sr_merge<.., 2> reduces the two lists to the query's best list
// and the band launch (pqt_k_sr_select PHASE 3) finishes it exactly as after the shared-row pass.  Every candidate's filter key is computed
// by pqt_rs_query's own MODE 2 instruction sequence (COOP only changes which batches a wavefront takes): same keys, same band, same results.
// The two wavefronts of a slot meet once per query, before the table is overwritten ("my partner no longer reads the previous table"); both
// then write the WHOLE new table (identical values to identical addresses: a wavefront reads only entries it has written itself).  The
// meeting is a counter in LDS; a wavefront that waits longer than 2^22 polls gives up, raises errFlag and leaves (a hang would take the
// device with it; its partner gives up the same way) -- pqt_get_stats reports PQT_ERR_DEVICE.
__device__ __forceinline__ bool pqt_pair_meet(uint32_t* const cnt, uint32_t& target) {
  target += 2u;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if ((threadIdx.x & 63u) == 0u) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  for (uint32_t spins = 0;; ++spins) {
    const uint32_t c = __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if ((int32_t)(c - target) >= 0) break;
    if (spins > (1u << 22)) return false;
    __builtin_amdgcn_s_sleep(2);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  return true;
}

template <int LPV, int UREQ, bool SHARDED, int C1M>
__global__ __launch_bounds__(1024) void pqt_k_pair_scan(const PqtRsArgs A, uint32_t* const errFlag) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr uint32_t LP = LPV * 4, C1 = 1u << C1M, TB = LP * C1 * 4, NSLOT = 512, NPAIR = 8;
  const uint32_t wave = threadIdx.x >> 6, pair = wave >> 1, half = wave & 1u;
  float* const sTab = reinterpret_cast<float*>(smem_raw + (size_t)pair * TB);
  uint64_t* const sKeys = reinterpret_cast<uint64_t*>(smem_raw + (size_t)NPAIR * TB) + (size_t)wave * NSLOT;
  unsigned long long* const sRuns = reinterpret_cast<unsigned long long*>(smem_raw + (size_t)NPAIR * TB + (size_t)2 * NPAIR * NSLOT * 8) + (size_t)wave * (A.runCap + A.runCap / 2);
  uint32_t* const sMeet = reinterpret_cast<uint32_t*>(smem_raw + (size_t)NPAIR * TB + (size_t)2 * NPAIR * (NSLOT * 8 + (size_t)A.runCap * 12));
  // (what the evaluating kernel does for the next call: statistics block and the schedule's registration block zeroed)
  if (blockIdx.x == 0 && threadIdx.x < 8 && A.zero8) A.zero8[threadIdx.x] = 0;
  if (blockIdx.x == 0 && A.poolNext) for (uint32_t t = threadIdx.x; t < 16u + 8u * PQT_SCHED_CLASSES; t += 1024) A.poolNext[t] = 0;
  if (threadIdx.x < NPAIR) sMeet[threadIdx.x] = 0u;
  __syncthreads();
  uint32_t target = 0, tiesAcc = 0, nN = 0;
  bool first = true;
  // static shares: slot (workgroup, pair) owns the queries slot, slot + #slots, ... -- both wavefronts of a pair walk the same sequence
  for (uint32_t q = blockIdx.x * NPAIR + pair; q < A.qn; q += gridDim.x * NPAIR) {
    if (!first && !pqt_pair_meet(&sMeet[pair], target)) {
      if ((threadIdx.x & 63u) == 0u) atomicAdd(errFlag, 1u);
      return;
    }
    first = false;
    pqt_rs_query<LPV, UREQ, false, SHARDED, C1M, 2, true, false, NSLOT, 0, 1>(A, q, A.nLocal[q], sKeys, sTab, A.coarse, 0xffffffffu, nN, 0u, tiesAcc, sRuns, half);
    __builtin_amdgcn_wave_barrier();
  }
}

// 5. selection: one wavefront per query over the distances of step 4 (pqt_rs_query PRE): no row is read before the band re-evaluation, the
// query's table stays in global memory (the band reads ~k entries of it), so a wavefront needs its key slots and run list only -- 5 KB of LDS
// instead of 12.5 KB, and none of the row registers of the evaluating kernel.
// LIST: the queries of A.qlist (those the first selection handed back: not covered by the pass, or a near-tie band beyond its 256 slots),
// MODE 0 over the EXACT distances pqt_k_sr_exact_list wrote for them.
template <int NW, int LPV, int UREQ, bool SHARDED, int C1M, bool LIST, int PHASE = 1 /* 1 whole selection, 2 scan only, 3 band + results only (table copied to LDS) */>
__global__ __launch_bounds__(NW * 64) void pqt_k_sr_select(const PqtRsArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr uint32_t LP = LPV * 4, C1 = 1u << C1M;
  constexpr int SCANSLOT = 1024;  // PHASE 2: key slots of the lean scan (256 best + up to 768 pending: a flush every >= 512 survivors)
  constexpr int NSLOT = PHASE == 2 ? SCANSLOT : PQT_RS_BEST + PQT_RS_PEND;
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  (void)lane;
  uint64_t* const sKeys = reinterpret_cast<uint64_t*>(smem_raw) + (size_t)wave * NSLOT;
  unsigned long long* const sRuns = reinterpret_cast<unsigned long long*>(smem_raw + (size_t)NW * NSLOT * 8) + (size_t)wave * (A.runCap + A.runCap / 2);
  float* const sTab = reinterpret_cast<float*>(smem_raw + (size_t)NW * ((size_t)NSLOT * 8 + (size_t)A.runCap * 12)) + (size_t)wave * LP * C1;  // PHASE 3 only
  (void)sRuns; (void)sTab;
  if constexpr (!LIST && PHASE != 3) {
    // (what the evaluating kernel does for the next call: statistics block and the schedule's registration block zeroed)
    if (blockIdx.x == 0 && threadIdx.x < 8 && A.zero8) A.zero8[threadIdx.x] = 0;
    if (blockIdx.x == 0 && A.poolNext) for (uint32_t t = threadIdx.x; t < 16u + 8u * PQT_SCHED_CLASSES; t += NW * 64) A.poolNext[t] = 0;
  }
  uint32_t tiesAcc = 0;
  const uint32_t slot = blockIdx.x * NW + wave;
  const uint32_t cnt = LIST ? *A.qcount : A.qn;
  for (uint32_t e = slot; e < cnt; e += gridDim.x * NW) {
    const uint32_t q = LIST ? A.qlist[e] : e;
    uint32_t nN = 0;
    float* table = const_cast<float*>(A.qL1virt) + (size_t)q * LP * C1;
    if constexpr (PHASE == 3) {
      // the band re-evaluation reads ~k x 2 LP entries of the table: from an LDS copy (3 x faster than gathering them from global memory)
      constexpr uint32_t NV = LP * C1 / 4, IT = (NV + 63) / 64;
      const float4* src4 = reinterpret_cast<const float4*>(table);
      float4 tmp[IT];
#pragma unroll
      for (uint32_t x = 0; x < IT; ++x) { const uint32_t t = lane + 64 * x; tmp[x] = src4[t < NV ? t : 0]; }
#pragma unroll
      for (uint32_t x = 0; x < IT; ++x) { const uint32_t t = lane + 64 * x; if (t < NV) reinterpret_cast<float4*>(sTab)[t] = tmp[x]; }
      __builtin_amdgcn_wave_barrier();
      table = sTab;
    }
    if constexpr (PHASE == 2) pqt_sr_scan_query<SCANSLOT>(A, q, A.nLocal[q], sKeys);
    else
    pqt_rs_query<LPV, UREQ, false, SHARDED, C1M, LIST ? 0 : 2, true, false, NSLOT, PHASE>(A, q, A.nLocal[q], sKeys, table, A.coarse, 0xffffffffu, nN, slot, tiesAcc, sRuns);
    __builtin_amdgcn_wave_barrier();
  }
  if constexpr (PHASE != 2) pqt_count_ties(&A.counters[3], tiesAcc);
}

// 6. the queries the selection handed back: EXACT distances (the reference's association, term by term, p ascending: the sequence of the
// band re-evaluation in pqt_rs_query) of all their candidates by one workgroup per query -- a wavefront alone needs ~1 ms for the 23 k
// candidates of a configs[2] query (the plain exact list kernel: that latency, once per batch, was most of what a fresh batch lost).
// Positions come from the query's runs, or from its plain candidate list when the traversal wrote one.
template <int NW, int LPV, int C1M>
__global__ __launch_bounds__(NW * 64) void pqt_k_sr_exact_list(const PqtRsArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr uint32_t LP = LPV * 4, C1 = 1u << C1M;
  float* const sVirt = reinterpret_cast<float*>(smem_raw);
  unsigned long long* const sRun = reinterpret_cast<unsigned long long*>(smem_raw + (size_t)LP * C1 * 4);  // PQT_RUNCAP entries
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t cnt = *A.qcount;
  // an item = XT candidates of one listed query: the handful of queries spread over the whole device
  constexpr uint32_t XT = 2048;
  const uint32_t maxTiles = (uint32_t)((A.stride + XT - 1) / XT);
  for (uint64_t it = blockIdx.x; it < (uint64_t)cnt * maxTiles; it += gridDim.x) {
    const uint32_t e = (uint32_t)(it / maxTiles), tile = (uint32_t)(it % maxTiles);
    const uint32_t q = A.qlist[e];
    const uint32_t n = A.nLocal[q];
    if (tile * XT >= n) continue;  // (uniform)
    const uint32_t jEnd = n < (tile + 1) * XT ? n : (tile + 1) * XT;
    const uint32_t m = A.nRuns ? A.nRuns[q] : 0xffffffffu;
    const bool listed = m != 0xffffffffu;
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < LP * C1 / 4; t += NW * 64) reinterpret_cast<float4*>(sVirt)[t] = reinterpret_cast<const float4*>(A.qL1virt + (size_t)q * LP * C1)[t];
    if (listed) for (uint32_t t = threadIdx.x; t < m; t += NW * 64) sRun[t] = A.runs[(size_t)q * PQT_RUNCAP + t];
    __syncthreads();
    const uint32_t* cid = A.cand + (size_t)q * A.stride;
    float* const drow = const_cast<float*>(A.preDist) + (size_t)q * A.stride;
    for (uint32_t j0 = tile * XT + wave * 64; j0 < jEnd; j0 += NW * 64) {
      const uint32_t j = j0 + lane;
      const bool act = j < jEnd;
      uint32_t posj = 0;
      if (act) {
        if (listed) {
          uint32_t lo = 0;  // last run whose first visiting position is <= j
#pragma unroll
          for (uint32_t s2 = PQT_RUNCAP / 2; s2 >= 1; s2 >>= 1) { const uint32_t mid = lo + s2; if (mid < m && (uint32_t)sRun[mid] <= j) lo = mid; }
          const unsigned long long rr = sRun[lo];
          posj = (uint32_t)(rr >> 32) + (j - (uint32_t)rr);
        } else posj = cid[j];
      }
      const uint4* row4 = reinterpret_cast<const uint4*>(A.codes + (size_t)posj * LP);
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
          scv[p] = A.coarse[((((p << C1M) + Aa) << C1M) + Bb)];
        }
      }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);  // all LP gathers in flight before the first use (see the band re-evaluation of pqt_rs_query)
      float acc = 0.f;
#pragma unroll
      for (int v = 0; v < LPV; ++v) {
        const uint32_t w[4] = {rv[v].x, rv[v].y, rv[v].z, rv[v].w};
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          const uint32_t p = v * 4 + x;
          const uint32_t Aa = w[x] & 0xffu, Bb = (w[x] >> 8) & 0xffu;
          const float lam = __builtin_fmaf((float)(w[x] >> 16), 8.f / 65536.f, -4.f);
          const float sb = sVirt[(p << C1M) + Aa], sa = sVirt[(p << C1M) + Bb];
          acc = acc + pqt_extract_distance(sa, sb, scv[p], lam);
        }
      }
      if (act) drow[j] = acc;
    }
  }
}
