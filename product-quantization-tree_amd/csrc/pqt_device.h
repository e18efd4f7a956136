// pqt_device.h -- device-side primitives shared by the gfx950 kernels (wave64, LDS-based).
// Written for CDNA4 only: 64-lane wavefronts, 160 KiB LDS per CU.  No CUDA/portable paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PQT_MAXP 8

struct PqtDevParams {
  uint32_t D, P, C1, C2, W, LP;
  uint32_t S;    // D / P   : dims per part
  uint32_t SS;   // D / LP  : dims per line part
  uint32_t R;    // LP / P  : line parts per part
  uint32_t WC;   // W * C2  : second-level entries per part
  uint32_t hashMod;           // 0, or the CUDA library's HASH_SIZE modulo (PerturbationProTree.hh:12)
  uint32_t tableSeed;         // seed of the second hash of the bin table
  uint32_t powers[PQT_MAXP];  // (C1*C2)^p mod 2^32  (treequantizer.hpp:45-49)
};

// two-choice hash table of the non-empty bins; gcount == 0 marks a free slot
struct __attribute__((aligned(16))) PqtBinEntry {
  uint32_t key;     // bin id (uint32, wrapped exactly like the reference's globIdx)
  uint32_t gcount;  // population of the bin in the WHOLE database (drives the cut)
  uint32_t lstart;  // first local member in the ids array
  uint32_t lcount;  // members held by this device (== gcount when unsharded)
};

// Two-choice (cuckoo) table: a key lives in slot h1(key) or h2(key), so a look-up is ONE round trip of two
// independent 16-byte reads -- no probe chains, no divergence.  The host builds the table (pqt_hip.hip: uploadBins).
__host__ __device__ __forceinline__ uint32_t pqt_hash1(uint32_t key, uint32_t bits) {
  return (key * 0x9E3779B1u) >> (32u - bits);
}
__host__ __device__ __forceinline__ uint32_t pqt_hash2(uint32_t key, uint32_t bits, uint32_t seed) {
  uint32_t x = key ^ seed;
  x ^= x >> 15; x *= 0x85EBCA6Bu; x ^= x >> 13;
  return (x * 0xC2B2AE35u) >> (32u - bits);
}
// presence filter in front of the table (one bit per hashed key, 2^bits bits): a clear bit proves the bin empty
__host__ __device__ __forceinline__ uint32_t pqt_hash_filter(uint32_t key, uint32_t bits) {
  // one multiply (the traversal evaluates it for every enumerated row and 32-bit multiplies are quarter rate): the top bits
  // of the Fibonacci product depend on every key bit
  return (key * 0x9E3779B1u) >> (32u - bits);
}
// size class of a query for the rerank schedule (quarter octaves of its local candidate count, 64 classes): queries are
// drawn class by class, largest first -- a longest-first order to within 19 %
#define PQT_SCHED_CLASSES 64
__host__ __device__ __forceinline__ uint32_t pqt_sched_class(uint32_t n) {
  if (n < 4u) return n;  // 0..3
  const uint32_t e = 31u - (uint32_t)__builtin_clz(n);  // >= 2
  const uint32_t c = 4u * (e - 1u) + ((n >> (e - 2u)) & 3u);
  return c < PQT_SCHED_CLASSES - 1u ? c : PQT_SCHED_CLASSES - 1u;
}
// returns {key, gcount, lstart, lcount}; gcount == 0 when the bin does not exist.  *slotOut = slot of the hit.
__device__ __forceinline__ uint4 pqt_table_lookup(const uint4* __restrict__ table4, uint32_t key, uint32_t bits, uint32_t seed,
                                                  uint32_t* slotOut) {
  const uint32_t s1 = pqt_hash1(key, bits), s2 = pqt_hash2(key, bits, seed);
  const uint4 e1 = table4[s1];
  const uint4 e2 = table4[s2];
  const bool h1 = e1.y != 0 && e1.x == key;
  const bool h2 = e2.y != 0 && e2.x == key;
  uint4 r = h1 ? e1 : e2;
  if (!h1 && !h2) r = make_uint4(key, 0u, 0u, 0u);
  *slotOut = h1 ? s1 : s2;
  return r;
}

// total order on f32 matching operator< (with -0 == +0): key(a) < key(b)  <=>  a < b
__device__ __forceinline__ uint32_t pqt_f2key(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u << 1) == 0u) u = 0u;  // -0 -> +0
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float pqt_key2f(uint32_t k) {
  uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __uint_as_float(u);
}

// cpu_version/helper.hpp:60-62 code_t::lambda(): u16 * 2^-13 - 4 (both steps exact in f32)
__device__ __forceinline__ float pqt_lambda_decode(uint32_t u16) { return (float)u16 * (8.f / 65536.f) - 4.f; }
// cpu_version/helper.hpp:74-77 code_t::toUShort (== pqt/triangle.cuh:6-12)
__device__ __forceinline__ uint32_t pqt_lambda_encode(float f) {
  const float ftrans = (f + 4.f) * (65536.f / 8.f);
  // the reference converts float -> unsigned short directly; NaN (c == 0, duplicate centroids) is UB there
  // and yields 0 on x86; reproduce that
  if (!(ftrans == ftrans)) return 0u;
  return (f >= 4.f) ? 65535u : ((f < -4.f) ? 0u : ((uint32_t)(int)ftrans & 0xffffu));
}
// cpu_version/helper.hpp:132-136 extractDistance, association exactly as written there.
// The translation unit is compiled with -ffp-contract=off so no FMA is formed.
__device__ __forceinline__ float pqt_extract_distance(float a, float b, float c, float l) {
  return b + l * l * c + l * (a - b - c);
}
// cpu_version/helper.hpp:169-172 calcRatio
__device__ __forceinline__ float pqt_calc_ratio(float a, float b, float c) { return -0.5f * (a - b - c) / c; }

// ---- block-wide bitonic sort of n (power of two) u64 keys in LDS or global memory, ascending ------
template <int BLOCK>
__device__ __forceinline__ void pqt_bitonic_sort_u64(uint64_t* a, uint32_t n) {
  for (uint32_t k = 2; k <= n; k <<= 1) {
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      for (uint32_t i = threadIdx.x; i < (n >> 1); i += BLOCK) {
        const uint32_t lo = ((i & ~(j - 1)) << 1) | (i & (j - 1));
        const uint32_t hi = lo | j;
        const bool up = ((lo & k) == 0);
        const uint64_t x = a[lo], y = a[hi];
        if ((x > y) == up) { a[lo] = y; a[hi] = x; }
      }
      __syncthreads();
    }
  }
}

// ---- wave / block scans (wave64) -----------------------------------------------------------------
// inclusive scan over the 64 lanes: four row-local DPP steps (row_shr 1, 2, 4, 8; lanes whose source lies outside the row add 0) and two
// row broadcasts (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3) -- six VALU-latency moves instead of six
// ds_bpermute round trips (~24 issue cycles each plus the LDS latency, all dependent)
__device__ __forceinline__ uint32_t pqt_wave_incl_scan(uint32_t v) {
#ifndef PQT_NO_DPP_SCAN
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);  // row_shr:1
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);  // row_shr:2
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);  // row_shr:4
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);  // row_shr:8
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
  return v;
#else
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(v, d, 64);
    if (lane >= d) v += o;
  }
  return v;
#endif
}
// exclusive block scan; sPart must hold BLOCK/64 + 1 words; returns exclusive prefix, *total = block sum
template <int BLOCK>
__device__ __forceinline__ uint32_t pqt_block_excl_scan(uint32_t v, uint32_t* sPart, uint32_t* total) {
  constexpr int NW = BLOCK / 64;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t inc = pqt_wave_incl_scan(v);
  if (lane == 63) sPart[wv] = inc;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t run = 0;
    for (int w = 0; w < NW; ++w) { const uint32_t t = sPart[w]; sPart[w] = run; run += t; }
    sPart[NW] = run;
  }
  __syncthreads();
  const uint32_t base = sPart[wv];
  *total = sPart[NW];
  __syncthreads();
  return base + inc - v;
}
