// pqt_wave.h -- wave64 register-level primitives for gfx950: a bitonic sorting network over 64*R u64 keys held
// R per lane (blocked layout: element e = lane*R + r), cross-lane exchanges by lane-xor shuffles.
// No LDS traffic, no barriers: one wavefront sorts 512 keys in ~1.5k VALU/shuffle instructions.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// value of lane (lane ^ LM).  LM = 1, 2, 8 are single DPP moves (VALU latency), LM = 4 two DPP moves and a select;
// LM = 16, 32 use the gfx950 row / half swaps (v_permlane16_swap_b32: rows 1 and 3 of the first operand change places with rows 0
// and 2 of the second; v_permlane32_swap_b32: the upper half of the first with the lower half of the second -- with both operands =
// v, one of the two results holds the partner's value in every lane).  Measured issue cost (scripts/micro/valu_rate.hip): a
// ds_bpermute_b32 occupies a wavefront for ~24 cycles plus the LDS round trip, a VALU op for ~5.
template <int LM>
__device__ __forceinline__ uint32_t pqt_lane_xor_u32(uint32_t v) {
#ifndef PQT_NO_PERMLANE_SWAP
  if (LM == 16) {
    const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return (threadIdx.x & 16) ? r[0] : r[1];
  }
  if (LM == 32) {
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return (threadIdx.x & 32) ? r[0] : r[1];
  }
#endif
  if (LM == 1) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
  if (LM == 2) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
  if (LM == 8) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x128, 0xf, 0xf, true);  // row_ror:8
  if (LM == 4) {
    const uint32_t up = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x104, 0xf, 0xf, true);  // row_shl:4  lane i <- i+4
    const uint32_t dn = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x114, 0xf, 0xf, true);  // row_shr:4  lane i <- i-4
    return (threadIdx.x & 4) ? dn : up;
  }
  return __shfl_xor(v, LM, 64);
}
template <int LM>
__device__ __forceinline__ uint64_t pqt_lane_xor_u64(uint64_t v) {
  const uint32_t lo = pqt_lane_xor_u32<LM>((uint32_t)v);
  const uint32_t hi = pqt_lane_xor_u32<LM>((uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}

// value of lane + 1 (lane 63 gets its own): DPP row shift + the row's first lane through a row broadcast is not a single move on
// gfx9; wave_shr-style moves are -- v_mov_b32 wave_shl:1 reads lane + 1
__device__ __forceinline__ uint32_t pqt_lane_down1_u32(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x130, 0xf, 0xf, false);  // wave_shl:1
}
__device__ __forceinline__ uint64_t pqt_lane_down1_u64(uint64_t v) {
  return ((uint64_t)pqt_lane_down1_u32((uint32_t)(v >> 32)) << 32) | pqt_lane_down1_u32((uint32_t)v);
}

// one compare-exchange stage (K = bitonic block size, J = partner distance) of the network
template <int R, int K, int J>
__device__ __forceinline__ void pqt_sort_stage(uint64_t (&key)[R], const int lane) {
  if constexpr (J < R) {
    // partner inside the lane
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if ((r & J) == 0) {
        const int r2 = r | J;
        // direction of the pair: ascending iff bit K of the element index (lane*R + r) is clear
        const bool desc = (K < R) ? ((r & K) != 0) : (((lane * R) & K) != 0);
        const uint64_t a = key[r], b = key[r2];
        const bool sw = (a > b) != desc;  // branch-free: compare mask XOR direction mask, then selects
        key[r] = sw ? b : a;
        key[r2] = sw ? a : b;
      }
    }
  } else {
    constexpr int LM = J / R;  // partner lane = lane ^ LM, same register slot
    const bool isLow = (lane & LM) == 0;
    const bool asc = ((lane * R) & K) == 0;
    const bool keepMax = (isLow != asc);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const uint64_t o = pqt_lane_xor_u64<LM>(key[r]);
      const uint64_t a = key[r];
      // keys are unique (position in the low word) except for ~0 padding, where either choice is the same
      const bool takeO = (o < a) != keepMax;
      key[r] = takeO ? o : a;
    }
  }
}
template <int R, int K, int J>
__device__ __forceinline__ void pqt_sort_merge(uint64_t (&key)[R], const int lane) {
  pqt_sort_stage<R, K, J>(key, lane);
  if constexpr (J > 1) pqt_sort_merge<R, K, J / 2>(key, lane);
}
template <int R, int K>
__device__ __forceinline__ void pqt_sort_level(uint64_t (&key)[R], const int lane) {
  pqt_sort_merge<R, K, K / 2>(key, lane);
  if constexpr (K < 64 * R) pqt_sort_level<R, K * 2>(key, lane);
}
// ascending sort of the 64*R keys of one wavefront; on return lane L holds sorted elements [L*R, L*R+R)
template <int R>
__device__ __forceinline__ void pqt_wave_sort_u64(uint64_t (&key)[R]) {
  pqt_sort_level<R, 2>(key, (int)(threadIdx.x & 63));
}

// rank of this lane among the set lanes of a predicate, and the total
__device__ __forceinline__ uint32_t pqt_ballot_rank(bool pred, uint32_t* total) {
  const unsigned long long m = __ballot(pred);
  const int lane = threadIdx.x & 63;
  *total = (uint32_t)__popcll(m);
  return (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
}

// ---- exact k-th smallest of the unique u64 keys of one wavefront (R per lane, any layout; ~0 = empty slot) ---------
// Most-significant-digit radix select with 256 LDS counters: each pass histograms the keys still inside [lo, hi] on the
// top 8 bits of (key - lo) relative to the width of that range, scans the counters (4 per lane + one wave scan) and
// narrows [lo, hi] to the counter holding rank `kth`.  The first pass starts from the true min/max of the keys, so two
// passes resolve ordinary data; the range shrinks by 2^8 per pass, which bounds the loop at 8 passes for any input.
// ~130 VALU instructions per pass against ~4k for the 512-key sorting network.
// hist: 256 u32 + 4 u64 of LDS owned by this wavefront (1056 bytes, 16-byte aligned).  kth is 1-based, <= #keys.
template <int R>
__device__ __forceinline__ uint64_t pqt_wave_kth_u64(const uint64_t (&key)[R], uint32_t kth, uint32_t* hist) {
  const uint32_t lane = threadIdx.x & 63;
  unsigned long long* slot = reinterpret_cast<unsigned long long*>(hist + 256);
  uint64_t mn = ~0ull, mx = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const bool v = key[r] != ~0ull;
    mn = (v && key[r] < mn) ? key[r] : mn;
    mx = (v && key[r] > mx) ? key[r] : mx;
  }
  if (lane == 0) { slot[0] = ~0ull; slot[1] = 0; }
  __builtin_amdgcn_wave_barrier();
  __hip_atomic_fetch_min(&slot[0], (unsigned long long)mn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  __hip_atomic_fetch_max(&slot[1], (unsigned long long)mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  __builtin_amdgcn_wave_barrier();
  uint64_t lo = slot[0], hi = slot[1];
  uint64_t tau = lo;
  for (int pass = 0; pass < 9; ++pass) {
    const uint64_t range = hi - lo;
    if (range == 0) { tau = lo; break; }
    const int msb = 63 - __builtin_clzll(range);
    const uint32_t sh = msb > 7 ? (uint32_t)(msb - 7) : 0u;
    reinterpret_cast<uint4*>(hist)[lane] = make_uint4(0, 0, 0, 0);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (key[r] >= lo && key[r] <= hi) atomicAdd(&hist[(uint32_t)((key[r] - lo) >> sh)], 1u);
    }
    __builtin_amdgcn_wave_barrier();
    const uint4 h = reinterpret_cast<const uint4*>(hist)[lane];
    const uint32_t s = h.x + h.y + h.z + h.w;
    const uint32_t incl = pqt_wave_incl_scan(s);
    uint32_t before = incl - s;
    const bool mine = before < kth && kth <= incl;
    // inside the owning lane: which of its 4 counters
    uint32_t b = lane * 4, m = h.x;
    if (before + h.x < kth) { before += h.x; b += 1; m = h.y;
      if (before + h.y < kth) { before += h.y; b += 1; m = h.z;
        if (before + h.z < kth) { before += h.z; b += 1; m = h.w; } } }
    const int owner = __builtin_ctzll(__ballot(mine));
    b = (uint32_t)__builtin_amdgcn_readlane((int)b, owner);
    m = (uint32_t)__builtin_amdgcn_readlane((int)m, owner);
    before = (uint32_t)__builtin_amdgcn_readlane((int)before, owner);
    kth -= before;
    lo = lo + ((uint64_t)b << sh);
    const uint64_t top = lo + ((1ull << sh) - 1ull);
    hi = top < hi ? top : hi;
    if (m == 1) {
      // the one key left in [lo, hi]
#pragma unroll
      for (int r = 0; r < R; ++r) if (key[r] >= lo && key[r] <= hi) slot[2] = key[r];
      __builtin_amdgcn_wave_barrier();
      tau = slot[2];
      break;
    }
  }
  __builtin_amdgcn_wave_barrier();
  return tau;
}

// ---- wave-wide min / max of u32 (row-local DPP steps + the two row broadcasts; lanes without a source keep their own value) ----------
template <bool MAXV>
__device__ __forceinline__ uint32_t pqt_wave_minmax_u32(uint32_t v) {
#define PQT_MM_STEP(CTRL, RM) { const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, RM, 0xf, false); v = MAXV ? (o > v ? o : v) : (o < v ? o : v); }
  PQT_MM_STEP(0x111, 0xf) PQT_MM_STEP(0x112, 0xf) PQT_MM_STEP(0x114, 0xf) PQT_MM_STEP(0x118, 0xf) PQT_MM_STEP(0x142, 0xa) PQT_MM_STEP(0x143, 0xc)
#undef PQT_MM_STEP
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// ---- k-th smallest (1-based, WITH multiplicity) of the u32 keys of one wavefront: element e = r * 64 + lane is valid iff e < have ------
// Same most-significant-digit radix select as pqt_wave_kth_u64 on half-width keys: the in-range test is one subtraction and one
// unsigned compare, the digit one shift; equal keys are allowed (the loop ends when the range is a single value).  hist: 256 u32 of LDS
// owned by the wavefront.  Returns the k-th smallest VALUE; the caller resolves ties at that value.
template <int R>
__device__ __forceinline__ uint32_t pqt_wave_kth_u32(const uint32_t (&key)[R], const uint32_t have, uint32_t kth, uint32_t* hist) {
  const uint32_t lane = threadIdx.x & 63;
  uint32_t mn = 0xffffffffu, mx = 0u;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if ((uint32_t)r * 64u < have) {  // uniform
      const bool v = (uint32_t)r * 64u + lane < have;
      mn = (v && key[r] < mn) ? key[r] : mn;
      mx = (v && key[r] > mx) ? key[r] : mx;
    }
  }
  uint32_t lo = pqt_wave_minmax_u32<false>(mn), hi = pqt_wave_minmax_u32<true>(mx);
  for (int pass = 0; pass < 5; ++pass) {
    const uint32_t range = hi - lo;
    if (range == 0) break;
    const int msb = 31 - __builtin_clz(range);
    const uint32_t sh = msb > 7 ? (uint32_t)(msb - 7) : 0u;
    reinterpret_cast<uint4*>(hist)[lane] = make_uint4(0, 0, 0, 0);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if ((uint32_t)r * 64u < have) {
        const uint32_t d = key[r] - lo;
        if ((uint32_t)r * 64u + lane < have && d <= range) atomicAdd(&hist[d >> sh], 1u);
      }
    }
    __builtin_amdgcn_wave_barrier();
    const uint4 h = reinterpret_cast<const uint4*>(hist)[lane];
    const uint32_t s = h.x + h.y + h.z + h.w;
    const uint32_t incl = pqt_wave_incl_scan(s);
    uint32_t before = incl - s;
    const bool mine = before < kth && kth <= incl;
    uint32_t b = lane * 4;
    if (before + h.x < kth) { before += h.x; b += 1;
      if (before + h.y < kth) { before += h.y; b += 1;
        if (before + h.z < kth) { before += h.z; b += 1; } } }
    const int owner = __builtin_ctzll(__ballot(mine));
    b = (uint32_t)__builtin_amdgcn_readlane((int)b, owner);
    before = (uint32_t)__builtin_amdgcn_readlane((int)before, owner);
    kth -= before;
    lo = lo + (b << sh);
    const uint32_t top = lo + ((1u << sh) - 1u);
    hi = top < hi ? top : hi;
    __builtin_amdgcn_wave_barrier();
  }
  return lo;
}

// ---- the same bitonic network over UNIQUE u32 keys (R per lane, blocked layout): a compare-exchange is one lane move + min + max + select
// instead of two moves, a 64-bit compare and two selects -- a third of the instructions of the u64 network.  Callers that need a
// (value, position) order build keys whose low bits hold the position and check the result against the full order (pqt_k_traverse's
// part sorts: truncated distance key | position, exact unless two distances agree in all but their last bits -- then the u64 network runs).
template <int R, int K, int J>
__device__ __forceinline__ void pqt_sort_stage32(uint32_t (&key)[R], const int lane) {
  if constexpr (J < R) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if ((r & J) == 0) {
        const int r2 = r | J;
        const bool desc = (K < R) ? ((r & K) != 0) : (((lane * R) & K) != 0);
        const uint32_t a = key[r], b = key[r2];
        const uint32_t mn = a < b ? a : b, mx = a < b ? b : a;
        key[r] = desc ? mx : mn;
        key[r2] = desc ? mn : mx;
      }
    }
  } else {
    constexpr int LM = J / R;
    const bool isLow = (lane & LM) == 0;
    const bool asc = ((lane * R) & K) == 0;
    const bool keepMax = (isLow != asc);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const uint32_t o = pqt_lane_xor_u32<LM>(key[r]);
      const uint32_t a = key[r];
      const uint32_t mn = a < o ? a : o, mx = a < o ? o : a;
      key[r] = keepMax ? mx : mn;
    }
  }
}
template <int R, int K, int J>
__device__ __forceinline__ void pqt_sort_merge32(uint32_t (&key)[R], const int lane) {
  pqt_sort_stage32<R, K, J>(key, lane);
  if constexpr (J > 1) pqt_sort_merge32<R, K, J / 2>(key, lane);
}
template <int R, int K>
__device__ __forceinline__ void pqt_sort_level32(uint32_t (&key)[R], const int lane) {
  pqt_sort_merge32<R, K, K / 2>(key, lane);
  if constexpr (K < 64 * R) pqt_sort_level32<R, K * 2>(key, lane);
}
template <int R>
__device__ __forceinline__ void pqt_wave_sort_u32(uint32_t (&key)[R]) {
  pqt_sort_level32<R, 2>(key, (int)(threadIdx.x & 63));
}

// ---- FOUR independent 64-key sorts in one pass, one per 16-lane row: unique u32 keys, 4 per lane, blocked layout (element e of a row's
// list = (lane & 15) * 4 + r); on return lane l of a row holds its sorted elements [4 l, 4 l + 4).  The network is the bitonic sorter in
// its direction-free form -- the first stage of every merge level compares element i with its MIRROR in the block (i ^ (K - 1)), all
// later stages i with i ^ J, and the smaller key always goes to the smaller index -- so the 11 stages whose partner lives in the same lane
// are a plain v_min / v_max pair per compare-exchange, and the 10 cross-lane stages are one DPP-fused v_min, one DPP-fused v_max and one
// select per key: lane ^ 1, ^ 2, ^ 3 are quad permutes, the mirrors of 8 and 16 lanes are row_half_mirror / row_mirror, lane ^ 4 is the
// one exchange that takes two moves.  178 instructions for the four sorts together (four passes of the 64-key one-key-per-lane network
// above: 644), and the lists enter and leave as 16-byte LDS accesses.
template <int CTRL>
__device__ __forceinline__ uint32_t pqt_dpp_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true); }
__device__ __forceinline__ void pqt_cx_u32(uint32_t& a, uint32_t& b) {
  const uint32_t mn = a < b ? a : b, mx = a < b ? b : a;
  a = mn; b = mx;
}
// compare-exchange with the lane named by the DPP control CTRL; MIRROR: the partner's registers in reverse order (a mirror stage);
// low = this lane holds the smaller index of every pair
template <int CTRL, bool MIRROR>
__device__ __forceinline__ void pqt_row_stage_u32(uint32_t (&k)[4], const bool low) {
  uint32_t o[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) o[r] = pqt_dpp_u32<CTRL>(k[MIRROR ? 3 - r : r]);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const uint32_t mn = k[r] < o[r] ? k[r] : o[r], mx = k[r] < o[r] ? o[r] : k[r];
    k[r] = low ? mn : mx;
  }
}
__device__ __forceinline__ void pqt_row_sort64_u32(uint32_t (&k)[4]) {
  const uint32_t l = threadIdx.x & 15u;
  const bool low1 = !(l & 1u), low2 = !(l & 2u), low4 = !(l & 4u), low8 = !(l & 8u);
  constexpr int X1 = 0xB1 /* quad_perm [1,0,3,2] */, X2 = 0x4E /* [2,3,0,1] */, X3 = 0x1B /* [3,2,1,0] */, X7 = 0x141 /* row_half_mirror */,
                X15 = 0x140 /* row_mirror */;
#define PQT_ROW_TAIL() do { pqt_cx_u32(k[0], k[2]); pqt_cx_u32(k[1], k[3]); pqt_cx_u32(k[0], k[1]); pqt_cx_u32(k[2], k[3]); } while (0)
  pqt_cx_u32(k[0], k[1]); pqt_cx_u32(k[2], k[3]);                                                                  // K = 2
  pqt_cx_u32(k[0], k[3]); pqt_cx_u32(k[1], k[2]); pqt_cx_u32(k[0], k[1]); pqt_cx_u32(k[2], k[3]);                  // K = 4
  pqt_row_stage_u32<X1, true>(k, low1); PQT_ROW_TAIL();                                                            // K = 8
  pqt_row_stage_u32<X3, true>(k, low2); pqt_row_stage_u32<X1, false>(k, low1); PQT_ROW_TAIL();                     // K = 16
  pqt_row_stage_u32<X7, true>(k, low4); pqt_row_stage_u32<X2, false>(k, low2); pqt_row_stage_u32<X1, false>(k, low1); PQT_ROW_TAIL();  // K = 32
  pqt_row_stage_u32<X15, true>(k, low8);                                                                           // K = 64
  {
    uint32_t o[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint32_t dn = pqt_dpp_u32<0x114>(k[r]);                                                    // row_shr:4  lane i <- i - 4
      o[r] = (uint32_t)__builtin_amdgcn_update_dpp((int)dn, (int)k[r], 0x104, 0xf, 0x5, false);        // row_shl:4 into lanes 0-3, 8-11: i <- i + 4
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint32_t mn = k[r] < o[r] ? k[r] : o[r], mx = k[r] < o[r] ? o[r] : k[r];
      k[r] = low4 ? mn : mx;
    }
  }
  pqt_row_stage_u32<X2, false>(k, low2); pqt_row_stage_u32<X1, false>(k, low1); PQT_ROW_TAIL();
#undef PQT_ROW_TAIL
}
