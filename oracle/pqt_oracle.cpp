// =====================================================================================
// oracle/pqt_oracle.cpp  --  TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT.
//
// A CPU restatement of the reference's `cpu_version` Product-Quantization-Tree
// (treequantizer<T,D,C1,C2,P,W,LP>), with run-time instead of compile-time parameters,
// used ONLY as the checker for the HIP path:
//   * tests/            (parity checks)
//   * __graft_entry__.smoke()
//   * bench.py's `cpu_baseline` leg
// Nothing in product-quantization-tree_amd/ may import, link or call this file.
//
// Every function cites the reference file:line it restates (paths relative to the
// reference root, i.e. cpu_version/...).  No reference source is copied; the code below is
// a re-expression of the algorithm with flat arrays instead of Eigen objects.
//
// PARITY PINNING STATUS
//   * pinned:   the line-quantisation arithmetic (lambda codec, extractDistance,
//               calcRatio, 4-byte code layout, pow<uint> wrap) is checked against
//               (a) the reference's own known answers in run.cu:33-104 and
//               (b) the genuine reference functions compiled from cpu_version/helper.hpp
//                   and pqt/triangle.cuh where they lie (oracle/_ref, see oracle/Makefile).
//   * UNPINNED: the tree traversal (`id`, `segmentInfo`, `orderBins`, `rerankVectors`,
//               `prepareHeuristic`, `prepareReranking`, k-means) cannot be checked against a
//               run of the reference: cpu_version/quantizer/*.hpp need the Eigen library,
//               which is neither vendored in the reference nor installed in this image, and
//               the reference holds no golden query results.  For those functions this file
//               *is* the definition of "reference behaviour" used by the tests: "parity
//               unpinned".  Two consequences are documented in DESIGN.md:
//                 - float summation order: Eigen's vectorised squaredNorm()/sum() order is
//                   not pinned by anything in the reference; this restatement uses plain
//                   left-to-right accumulation with separate multiply and add (compile with
//                   -ffp-contract=off), which is the literal reading of the source.
//                 - tie order of the reference's std::sort calls depends on libstdc++'s
//                   introsort; this restatement calls std::sort with the same comparators on
//                   the same initial sequences (sort_mode 0) and can alternatively use
//                   std::stable_sort (sort_mode 1, the canonical tie order the HIP path
//                   implements).  Without exact float ties both modes coincide.
// =====================================================================================
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <string>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned int uint;

namespace {

// ---- cpu_version/helper.hpp:19-22  pow<T>(x,n): plain repeated multiplication in T (uint wraps mod 2^32)
static inline uint upow(uint x, uint n) {
  uint r = 1;
  for (uint i = 0; i < n; ++i) r = x * r;
  return r;
}

// ---- cpu_version/helper.hpp:74-77  code_t::toUShort (identical to pqt/triangle.cuh:6-12)
static inline unsigned short lambda_encode(float f) {
  float ftrans = (f + 4.f) * (65536.f / 8.f);
  return (unsigned short)((f >= 4.f) ? 65535 : ((f < -4.f) ? 0 : ftrans));
}
// ---- cpu_version/helper.hpp:60-62  code_t::lambda()  (pqt/triangle.cuh:14-18 toFloat)
static inline float lambda_decode(unsigned short u) { return (float(u) * (8.f / 65536.f) - 4.f); }

// ---- cpu_version/helper.hpp:39-90  code_t: 4 bytes {u8 p1; u8 p2; u16 l} (little endian)
static inline uint32_t code_pack(uint a, uint b, float l) {
  return (uint32_t)(a & 0xff) | ((uint32_t)(b & 0xff) << 8) | ((uint32_t)lambda_encode(l) << 16);
}
static inline uint code_a(uint32_t c) { return c & 0xff; }
static inline uint code_b(uint32_t c) { return (c >> 8) & 0xff; }
static inline float code_lambda(uint32_t c) { return lambda_decode((unsigned short)(c >> 16)); }

// ---- cpu_version/helper.hpp:132-136  extractDistance: b + l*l*c + l*(a-b-c), f32, this association
static inline float extract_distance(float a, float b, float c, float l) {
  return b + l * l * c + l * (a - b - c);
}
// ---- cpu_version/helper.hpp:169-172  calcRatio
static inline float calc_ratio(float a, float b, float c) { return -0.5f * (a - b - c) / c; }

// squared norm of (x - y) over n dims.  Stands for `(vec - cec).segment(..).squaredNorm()` (treequantizer.hpp:197,
// 645-655, vectorquantizer.hpp:88,106-109, productquantizer.hpp:55).  Eigen's reduction order is not pinned by anything in
// the reference (un-vendored library, version unknown), so the order is a run-time mode:
//   0  sequential left to right, separate multiply and add: the literal reading of the source.  DEFAULT, and the order
//      the HIP kernels implement (bit-identical tables).
//   1..5  SENSITIVITY PROBES (tests/test_cpu_sum_order.py): the orders Eigen's vectorised redux would produce, restated
//      from Eigen's published algorithm (Core/Redux.h: packet partial sums, then a horizontal add `predux`):
//      1  SSE2 packets of 4 -- what the reference's own build flags select (cpu_version/CMakeLists.txt:20 has no -march)
//         -- linear traversal with two packet accumulators (Eigen 3.3 NoUnrolling), predux (a0+a2)+(a1+a3)
//      2  SSE2 packets of 4, balanced tree over the packets (CompleteUnrolling of a fixed-size vector), same predux
//      3  SSE2 packets of 4, one packet accumulator (Eigen 3.2), same predux
//      4  AVX packets of 8 (a -march=native build): two accumulators, predux = fold halves then the SSE predux; fewer
//         than 8 elements are summed sequentially (too small to vectorise)
//      5  sequential with fused multiply-add (a -Ofast -march=native scalar loop)
//      For 4 and 8 elements (the L1virt segments of every BASELINE shape) modes 1-3 coincide.
static inline float predux4(const float* a) { return (a[0] + a[2]) + (a[1] + a[3]); }
static inline float sqdist(int mode, const float* x, const float* y, uint n) {
  if (mode == 0 || n < 4 || (mode == 4 && n < 8)) {
    float s = 0;
    for (uint i = 0; i < n; ++i) {
      const float d = x[i] - y[i];
      s += d * d;
    }
    return s;
  }
  if (mode == 5) {
    float s = 0;
    for (uint i = 0; i < n; ++i) { const float d = x[i] - y[i]; s = __builtin_fmaf(d, d, s); }
    return s;
  }
  const uint PS = mode == 4 ? 8 : 4;
  const uint np = n / PS;  // whole packets; the tail is added sequentially after the horizontal add (Redux.h)
  float pk[64][8];
  if (np > 64) return -1.f;
  for (uint k = 0; k < np; ++k)
    for (uint l = 0; l < PS; ++l) { const float d = x[k * PS + l] - y[k * PS + l]; pk[k][l] = d * d; }
  auto addp = [&](float* a, const float* b2) { for (uint l = 0; l < PS; ++l) a[l] = a[l] + b2[l]; };
  float acc[8];
  if (mode == 2) {
    // balanced tree: redux_vec_unroller splits [start, start+len) into halves
    struct T { static void run(float (*pk)[8], uint PS, uint s0, uint len, float* out) {
      if (len == 1) { for (uint l = 0; l < PS; ++l) out[l] = pk[s0][l]; return; }
      float a[8], b[8];
      run(pk, PS, s0, len / 2, a); run(pk, PS, s0 + len / 2, len - len / 2, b);
      for (uint l = 0; l < PS; ++l) out[l] = a[l] + b[l];
    } };
    T::run(pk, PS, 0, np, acc);
  } else if (mode == 3 || np == 1) {
    for (uint l = 0; l < PS; ++l) acc[l] = pk[0][l];
    for (uint k = 1; k < np; ++k) addp(acc, pk[k]);
  } else {  // modes 1, 4: two accumulators over pairs of packets, then res0 + res1, then a possible odd last packet
    float r1[8];
    for (uint l = 0; l < PS; ++l) { acc[l] = pk[0][l]; r1[l] = pk[1][l]; }
    const uint np2 = np / 2 * 2;
    for (uint k = 2; k < np2; k += 2) { addp(acc, pk[k]); addp(r1, pk[k + 1]); }
    addp(acc, r1);
    if (np > np2) addp(acc, pk[np2]);
  }
  float s;
  if (PS == 8) { float h[4]; for (uint l = 0; l < 4; ++l) h[l] = acc[l] + acc[l + 4]; s = predux4(h); }
  else s = predux4(acc);
  for (uint i = np * PS; i < n; ++i) { const float d = x[i] - y[i]; s += d * d; }
  return s;
}
static inline float sqdist_seq(const float* x, const float* y, uint n) { return sqdist(0, x, y, n); }

template <class It, class Cmp>
static inline void do_sort(int mode, It b, It e, Cmp c) {
  if (mode == 0) std::sort(b, e, c); else std::stable_sort(b, e, c);
}

struct Ctx {  // per-query mutable state (the reference keeps it in the object: treequantizer.hpp:913-915)
  std::vector<float> L1, L1virt;
  std::vector<uint> L1order;
  std::vector<uint> segOrder, segL1, segL2;
  std::vector<float> segD1, segD2;
  std::vector<float> vqd;  // vectorquantizer::_L1distances
  std::vector<uint> vqo;   // vectorquantizer::_L1order
};

struct Oracle {
  uint D, P, C1, C2, W, LP, S, SS, base;
  size_t maxMultiIndex;
  std::vector<uint> powers;
  std::vector<float> cb1, cb2, coarse;
  std::vector<uint> heur;  // first heurRows tuples, heurRows*P
  size_t heurRows;
  std::map<uint, std::vector<uint>> bins;
  std::vector<uint32_t> codes;
  size_t curId;
  int sortMode;
  int sumMode = 0;  // order of the squared-norm sums, see sqdist(); 0 = sequential (the parity definition)
  Ctx ctx;  // context used by the single-query entry points

  void initCtx(Ctx& c) const {
    c.L1.assign(P * C1, 0); c.L1virt.assign(LP * C1, 0); c.L1order.assign(P * C1, 0);
    const size_t n = (size_t)P * W * C2;
    c.segOrder.assign(n, 0); c.segL1.assign(n, 0); c.segL2.assign(n, 0);
    c.segD1.assign(n, 0); c.segD2.assign(n, 0);
    c.vqd.assign(C2, 0); c.vqo.assign(C2, 0);
  }

  const float* cb2ptr(uint p, uint c1) const { return &cb2[((size_t)p * C1 + c1) * C2 * S]; }

  // ---- treequantizer.hpp:75-127 prepareHeuristic: all tuples in {0..base-1}^P (digit p of idx in
  // base `base`), sorted by squared norm with the unstable std::sort; only the first `keep` rows are
  // retained here (the reference keeps all of them but orderBins reads only the first boundBins).
  void prepareHeuristic(size_t keep) {
    mode2d = false;
    const size_t M = maxMultiIndex;
    std::vector<float> norm(M);
    std::vector<uint> order(M);
    for (size_t idx = 0; idx < M; ++idx) {
      uint dec = (uint)idx;
      float s = 0;  // squaredNorm of the P-vector of digits, sequential
      // digits beyond the most significant non-zero one are 0 and add 0
      std::vector<float> dig(P, 0.f);
      uint p = 0;
      while (dec > 0) { dig[p] = (float)(dec % base); dec /= base; ++p; }
      for (uint q = 0; q < P; ++q) s += dig[q] * dig[q];
      norm[idx] = s;
      order[idx] = (uint)idx;
    }
    const float* nptr = norm.data();
    do_sort(sortMode, order.begin(), order.end(), [nptr](const uint& l, const uint& r) { return nptr[l] < nptr[r]; });
    heurRows = std::min(keep, M);
    heur.assign(heurRows * P, 0);
    for (size_t h = 0; h < heurRows; ++h) {
      uint dec = order[h];
      uint p = 0;
      while (dec > 0) { heur[h * P + p] = dec % base; dec /= base; ++p; }
    }
  }

  // ==== optional mode: the CUDA library's 2-D anisotropic sequences.  NOT cpu_version behaviour: it replaces the
  // shared tuple table by per-query rows chosen the way the CUDA 1B path chooses its bins; everything after the choice of the rows
  // (bin ids, distances, exact sort, cut, rerank) stays the cpu_version text above.  Restated from the CUDA sources, which cannot
  // be run here (no CUDA): "parity unpinned" like the rest of the traversal.
  bool mode2d = false;
  uint seqDc = 0;
  std::vector<uint> seq2d;  // [10][65536] cell numbers
  float thr2d[9];
  // ---- pqt/ProTree.cu:50-126 prepare2DDistSequence(_maxCluster); ProTree.hh:9-13 NUM_DISTSEQ 65536, NUM_ANISO_DIR 10, ANISO_BASE 1.2f
  void prepare2DDistSequence(uint maxCluster) {
    const uint NUM_DISTSEQ = 65536, NUM_ANISO_DIR = 10;
    const float ANISO_BASE = 1.2f;
    const uint nVec = maxCluster * maxCluster;                    // :55 pow(_maxCluster, 2)
    const uint copyVec = nVec < NUM_DISTSEQ ? nVec : NUM_DISTSEQ;  // :57
    seq2d.assign((size_t)NUM_DISTSEQ * NUM_ANISO_DIR, 0u);        // :113-114 zero fill
    for (uint slope = 0; slope < NUM_ANISO_DIR; ++slope) {
      const float s = (float)pow(0.9 * ANISO_BASE, (int)slope - (int)(NUM_ANISO_DIR / 2));  // :68 (double arithmetic, stored in a float)
      std::vector<std::pair<float, uint> > dists;
      for (uint i = 0; i < nVec; ++i) {
        const float x = (float)(i % maxCluster);  // :76
        const float y = (float)(i / maxCluster);  // :77
        const float n = 0.8f;                     // :89
        const float dist = std::pow(x, n) + s * std::pow(y, n);  // :91 (float overloads: the file is compiled as C++ with <cmath>)
        dists.push_back(std::pair<float, uint>(dist, i));
      }
      std::sort(dists.begin(), dists.end());  // :105 (pairs: ties in the key fall back to the cell number, so the order is unique)
      for (uint i = 0; i < copyVec; ++i) seq2d[(size_t)slope * NUM_DISTSEQ + i] = dists[i].second;  // :116-117
    }
    seqDc = maxCluster;
    // slope classes: the CUDA text computes si = roundf(logf(slope) / logf(ANISO_BASE)) + NUM_ANISO_DIR / 2 clamped to [0, 9]
    // (PerturbationProTree.cu:2851-2853).  roundf(t) + 5 >= j + 1  <=>  t >= j - 4.5  <=>  slope >= 1.2^(j - 4.5): the class is the number
    // of these 9 boundaries at or below the slope.  Stated through the boundaries (runtime powf, like the engine) so that the
    // class does not depend on which logf is linked; a NaN or negative slope falls into class 0 as the clamped expression does.
    volatile float b = ANISO_BASE;
    for (int j = 0; j < 9; ++j) thr2d[j] = powf(b, (float)j - 4.5f);
    heurRows = copyVec;
    heur.assign(P, 0);
    mode2d = true;
  }
  // ---- pqt/PerturbationProTree.cu:2839-2858 computeSlopeIdx (sample positions passed in: sqrtf(2 N) and one below, see rows2D)
  uint slopeIdx(const float* val0, const float* val1, uint s, uint sm) const {
    const float slope = (val1[s] + val1[sm] - 2 * val1[0]) / (val0[s] + val0[sm] - 2 * val0[0]);  // :2847-2848
    uint si = 0;
    for (int j = 0; j < 9; ++j) if (slope >= thr2d[j]) ++si;
    return si;
  }
  // ---- the per-query rows: selectBinKernel2D2Parts (:2914-3006, with generate2DBins :2888-2910) for the pairs of parts, then the
  // row loop of selectBinKernel2DFinal (:3047-3075) over the two merged lists.  rows[h*P .. ] = the four part ranks of row h,
  // or 0xffffffff in digit 0 when the cell names no bin (the CUDA kernels give such cells distance 99999999999 and bin 0).
  void rows2D(const Ctx& c, size_t h_e, std::vector<uint>& rows) const {
    const uint WC = W * C2;
    const uint kMax = WC < 64 ? WC : 64;     // getBIGBins2D :3729 kMax = 64 (clamped to the list length here)
    const uint nInter = 256;                 // :3735 nIntermediateBin
    // computeSlopeIdx samples at sqrtf(2.f * N) and one below: N = nInter for the pairs (:2970), N = 1024 for the final merge (:3071)
    uint s1 = (uint)sqrtf(2.f * nInter); if (s1 > kMax - 1) s1 = kMax - 1;
    const uint s1m = s1 ? s1 - 1 : 0;
    const uint s2 = (uint)sqrtf(2.f * 1024), s2m = s2 - 1;
    struct Ent { float d; uint t, x, y; bool pad; };
    std::vector<Ent> pair[2];
    std::vector<float> ld[2];
    for (uint j = 0; j < 2; ++j) {
      std::vector<float> v0(kMax), v1(kMax);  // the sorted assignment values of parts 2j and 2j+1 (:2947-2962)
      for (uint r = 0; r < kMax; ++r) {
        v0[r] = c.segD2[(2 * j) * WC + c.segOrder[(2 * j) * WC + r]];
        v1[r] = c.segD2[(2 * j + 1) * WC + c.segOrder[(2 * j + 1) * WC + r]];
      }
      const uint si = slopeIdx(v0.data(), v1.data(), s1, s1m);
      pair[j].resize(nInter);
      for (uint t = 0; t < nInter; ++t) {   // generate2DBins :2894-2907
        const uint cell = seq2d[(size_t)si * 65536 + t];
        const uint x = cell % seqDc, y = cell / seqDc;
        Ent e; e.t = t; e.x = x; e.y = y;
        if (x < kMax && y < kMax) { e.d = v0[x] + v1[y]; e.pad = false; }
        else { e.d = 99999999999.f; e.pad = true; }
        pair[j][t] = e;
      }
      // bitonic3 (:2987) orders by the distance; equal distances are taken in cell order here
      std::sort(pair[j].begin(), pair[j].end(), [](const Ent& l, const Ent& r) { return l.d < r.d || (l.d == r.d && l.t < r.t); });
      ld[j].resize(nInter);
      for (uint t = 0; t < nInter; ++t) ld[j][t] = pair[j][t].d;
    }
    const uint si2 = slopeIdx(ld[0].data(), ld[1].data(), s2, s2m);
    rows.assign(h_e * P, 0);
    for (size_t h = 0; h < h_e; ++h) {
      const uint cell = seq2d[(size_t)si2 * 65536 + h];  // :3079-3085 (chunk nIter, thread t: entry nIter * 1024 + t)
      const uint x = cell % seqDc, y = cell / seqDc;
      if (x < nInter && y < nInter && !pair[0][x].pad && !pair[1][y].pad) {
        rows[h * P + 0] = pair[0][x].x; rows[h * P + 1] = pair[0][x].y;
        rows[h * P + 2] = pair[1][y].x; rows[h * P + 3] = pair[1][y].y;
      } else rows[h * P] = 0xffffffffu;
    }
  }

  // ---- treequantizer.hpp:183-203 computeLookupTable: coarse[(lp*C1 + j)*C1 + i] = ||cb1[i]-cb1[j]||^2 on sub-segment lp
  void computeLookupTable() {
    coarse.assign((size_t)LP * C1 * C1, 0);
    for (uint i = 0; i < C1; ++i)
      for (uint j = i; j < C1; ++j)
        for (uint p = 0; p < LP; ++p) {
          const float d = sqdist(sumMode, &cb1[(size_t)i * D + p * SS], &cb1[(size_t)j * D + p * SS], SS);
          coarse[((size_t)p * C1 + j) * C1 + i] = d;
          coarse[((size_t)p * C1 + i) * C1 + j] = d;
        }
  }

  // ---- vectorquantizer.hpp:104-115 dist(): distances of a segment to the C2 centroids of one cell
  void vqDist(Ctx& c, uint p, uint c1, const float* seg) const {
    const float* cen = cb2ptr(p, c1);
    for (uint k = 0; k < C2; ++k) { c.vqd[k] = sqdist(sumMode, seg, cen + (size_t)k * S, S); c.vqo[k] = k; }
  }
  // ---- vectorquantizer.hpp:83-102 id(): dist() + sort of the order array, returns the nearest
  // (the comparator's uint8_t parameters truncate the indices; harmless for C2 <= 256)
  uint vqId(Ctx& c, uint p, uint c1, const float* seg) const {
    vqDist(c, p, c1, seg);
    const float* d = c.vqd.data();
    do_sort(sortMode, c.vqo.begin(), c.vqo.end(), [d](const uint8_t& l, const uint8_t& r) { return d[l] < d[r]; });
    return c.vqo[0];
  }

  // ---- treequantizer.hpp:640-688 id(): L1 tables + per-part order + bin id of the nearest cell
  uint id(Ctx& c, const float* vec) const {
    const uint R = LP / P;
    for (uint cc = 0; cc < C1; ++cc) {
      const float* cen = &cb1[(size_t)cc * D];
      for (uint p = 0; p < P; ++p) {
        c.L1order[p * C1 + cc] = cc;
        float d = 0;
        for (uint pp = 0; pp < R; ++pp) {
          const uint lp = pp + p * R;
          const float dd = sqdist(sumMode, vec + lp * SS, cen + lp * SS, SS);
          c.L1virt[(size_t)lp * C1 + cc] = dd;
          d += dd;
        }
        c.L1[p * C1 + cc] = d;
      }
    }
    for (uint p = 0; p < P; ++p) {
      const float* dist = &c.L1[p * C1];
      do_sort(sortMode, c.L1order.begin() + p * C1, c.L1order.begin() + (p + 1) * C1,
              [dist](const uint& l, const uint& r) { return dist[l] < dist[r]; });
    }
    uint i = 0;
    for (uint p = 0; p < P; ++p) {
      const uint cbest = c.L1order[p * C1 + 0];
      const uint second = vqId(c, p, cbest, vec + p * S);
      i += (cbest * C2 + second) * powers[p];
    }
    return i;
  }

  // L1 tables only (what a query needs from id(); the trailing bin-id part of id() has no effect on queries)
  void l1tables(Ctx& c, const float* vec) const {
    const uint R = LP / P;
    for (uint cc = 0; cc < C1; ++cc) {
      const float* cen = &cb1[(size_t)cc * D];
      for (uint p = 0; p < P; ++p) {
        c.L1order[p * C1 + cc] = cc;
        float d = 0;
        for (uint pp = 0; pp < R; ++pp) {
          const uint lp = pp + p * R;
          const float dd = sqdist(sumMode, vec + lp * SS, cen + lp * SS, SS);
          c.L1virt[(size_t)lp * C1 + cc] = dd;
          d += dd;
        }
        c.L1[p * C1 + cc] = d;
      }
    }
    for (uint p = 0; p < P; ++p) {
      const float* dist = &c.L1[p * C1];
      do_sort(sortMode, c.L1order.begin() + p * C1, c.L1order.begin() + (p + 1) * C1,
              [dist](const uint& l, const uint& r) { return dist[l] < dist[r]; });
    }
  }

  // ---- treequantizer.hpp:597-630 segmentInfo
  void segmentInfo(Ctx& c, const float* vec) const {
    const uint WC = W * C2;
    for (uint p = 0; p < P; ++p) {
      uint pos = 0;
      for (uint h1 = 0; h1 < W; ++h1) {
        const uint c1 = c.L1order[p * C1 + h1];
        const float l1 = c.L1[p * C1 + c1];
        vqDist(c, p, c1, vec + p * S);
        for (uint h2 = 0; h2 < C2; ++h2) {
          c.segL1[p * WC + pos] = c1;
          c.segL2[p * WC + pos] = h2;
          c.segD1[p * WC + pos] = l1;
          c.segD2[p * WC + pos] = c.vqd[h2];
          c.segOrder[p * WC + pos] = pos;
          ++pos;
        }
      }
      const float* d2 = &c.segD2[p * WC];
      do_sort(sortMode, c.segOrder.begin() + p * WC, c.segOrder.begin() + (p + 1) * WC,
              [d2](const uint& l, const uint& r) { return d2[l] < d2[r]; });
    }
  }

  // ---- treequantizer.hpp:548-588 orderBins
  void orderBins(const Ctx& c, uint maxBins, std::vector<std::pair<uint, float>>& binCand,
                 std::vector<uint>& seqOrder, bool sortBins) const {
    const uint WC = W * C2;
    size_t h_e = std::min((size_t)maxBins, maxMultiIndex);
    if (h_e > heurRows) h_e = heurRows;  // callers keep heurRows >= boundBins
    binCand.clear(); binCand.reserve(h_e);
    seqOrder.clear(); seqOrder.reserve(h_e);
    std::vector<uint> rowsQ;
    if (mode2d) rows2D(c, h_e, rowsQ);  // optional mode: this query's own rows
    const uint* tab = mode2d ? rowsQ.data() : heur.data();
    for (size_t h = 0; h < h_e; ++h) {
      uint globIdx = 0;
      float fine = 0;
      if (mode2d && tab[h * P] == 0xffffffffu) continue;  // a cell without a bin
      for (uint p = 0; p < P; ++p) {
        const uint idx = tab[h * P + p];
        const uint kkk = c.segOrder[p * WC + idx];
        fine += c.segD2[p * WC + kkk];
        globIdx += (c.segL1[p * WC + kkk] * C2 + c.segL2[p * WC + kkk]) * powers[p];
      }
      binCand.push_back({globIdx, fine});
      seqOrder.push_back((uint)(binCand.size() - 1));
    }
    if (sortBins) {
      const std::pair<uint, float>* bc = binCand.data();
      do_sort(sortMode, seqOrder.begin(), seqOrder.end(),
              [bc](const uint& l, const uint& r) { return bc[l].second < bc[r].second; });
    }
  }

  // ---- treequantizer.hpp:423-439 distance(dbIdx)
  float distance(const Ctx& c, size_t dbIdx) const {
    float approx = 0;
    for (uint p = 0; p < LP; ++p) {
      const uint32_t code = codes[dbIdx * LP + p];
      const uint A = code_a(code), B = code_b(code);
      const float lambda = code_lambda(code);
      const float side_b = c.L1virt[(size_t)p * C1 + A];
      const float side_a = c.L1virt[(size_t)p * C1 + B];
      const float side_c = coarse[((size_t)p * C1 + A) * C1 + B];
      approx += extract_distance(side_a, side_b, side_c, lambda);
    }
    return approx;
  }

  // ---- treequantizer.hpp:450-484 rerankVectors (the reference's _bins[...] inserts empty bins; find() does not)
  void rerankVectors(const Ctx& c, uint maxVecs, const std::vector<std::pair<uint, float>>& binDesc,
                     const std::vector<uint>& seqOrder, std::vector<std::pair<uint, float>>& out) const {
    uint used = 0;
    bool stop = false;
    for (size_t b = 0; b < binDesc.size(); ++b) {
      const uint g = binDesc[seqOrder[b]].first;
      auto it = bins.find(g);
      if (it != bins.end()) {
        for (uint v : it->second) {
          out.push_back({v, distance(c, v)});
          ++used;
          if (used > maxVecs) stop = true;
        }
      }
      if (stop) break;
    }
    do_sort(sortMode, out.begin(), out.end(),
            [](const std::pair<uint, float>& l, const std::pair<uint, float>& r) { return l.second < r.second; });
  }

  // ---- treequantizer.hpp:323-350 query
  void query(Ctx& c, uint Bv, uint Bb, const float* vec, std::vector<std::pair<uint, float>>& out, bool sortBins) const {
    std::vector<std::pair<uint, float>> binCand;
    std::vector<uint> seqOrder;
    l1tables(c, vec);
    segmentInfo(c, vec);
    orderBins(c, Bb, binCand, seqOrder, sortBins);
    out.clear();
    rerankVectors(c, Bv, binCand, seqOrder, out);
  }

  // ---- treequantizer.hpp:356-412 prepareReranking: line code of the vector whose L1virt is in ctx
  void prepareReranking(const Ctx& c, uint32_t* out) const {
    for (uint p = 0; p < LP; ++p) {
      float best_lambda = 0, best_err = HUGE_VALF;
      uint bestA = 0, bestB = 0;
      for (uint A = 0; A < C1; ++A) {
        const float side_b = c.L1virt[(size_t)p * C1 + A];
        for (uint B = A + 1; B < C1; ++B) {
          const float side_a = c.L1virt[(size_t)p * C1 + B];
          const float side_c = coarse[((size_t)p * C1 + A) * C1 + B];
          const float lambda = calc_ratio(side_a, side_b, side_c);
          const float err = extract_distance(side_a, side_b, side_c, lambda);
          if (err < best_err) { best_err = err; best_lambda = lambda; bestA = A; bestB = B; }
        }
      }
      out[p] = code_pack(bestA, bestB, best_lambda);
    }
  }
};

// ------------------------------------------------------------------------------------------
// k-means by centroid splitting (offline; "next" row of SURVEY §8f).  Restates
// productquantizer.hpp:40-158 and vectorquantizer.hpp:33-146.
// ------------------------------------------------------------------------------------------
static void pq_generate(uint D, uint P, uint C, const float* const* vecs, size_t n, float* cen /*C*D*/, std::vector<uint8_t>& mapping) {
  const uint S = D / P;
  mapping.assign(n * P, 0);
  std::vector<float> dist(n * P, 0.f);
  std::fill(cen, cen + (size_t)C * D, 0.f);
  // iterator::center() (iterator/iterator.hpp:41-53)
  {
    std::vector<float> avg(D, 0.f);
    for (size_t i = 0; i < n; ++i) for (uint d = 0; d < D; ++d) avg[d] += vecs[i][d];
    for (uint d = 0; d < D; ++d) cen[d] = avg[d] / (float)n;
  }
  uint step = 1;
  float cur = 0, last = 0;
  do {
    uint run = 1000;
    for (uint i = 0; i < step; ++i)  // augmentCentroids (productquantizer.hpp:97-103)
      for (uint d = 0; d < D; ++d) {
        cen[(size_t)(i + step) * D + d] = cen[(size_t)i * D + d] + 0.001f;
        cen[(size_t)i * D + d] = cen[(size_t)i * D + d] - 0.001f;
      }
    step *= 2;
    do {
      last = cur;
#pragma omp parallel for schedule(static)
      for (long long i = 0; i < (long long)n; ++i)  // getAssignment (:40-66)
        for (uint p = 0; p < P; ++p) {
          uint best = 0; float bd = HUGE_VALF;
          for (uint c = 0; c < step; ++c) {
            const float d = sqdist_seq(vecs[i] + p * S, cen + (size_t)c * D + p * S, S);
            if (d < bd) { bd = d; best = c; }
          }
          mapping[i * P + p] = (uint8_t)best; dist[i * P + p] = bd;
        }
      std::fill(cen, cen + (size_t)C * D, 0.f);  // updateCentroids (:74-91)
      std::vector<float> cnt((size_t)C * P, 0.f);
      for (size_t i = 0; i < n; ++i)
        for (uint p = 0; p < P; ++p) {
          const uint c = mapping[i * P + p];
          for (uint s = 0; s < S; ++s) cen[(size_t)c * D + p * S + s] += vecs[i][p * S + s];
          cnt[p * C + c] += 1.f;
        }
      for (uint c = 0; c < C; ++c) for (uint p = 0; p < P; ++p)
        if (cnt[p * C + c] != 0) for (uint s = 0; s < S; ++s) cen[(size_t)c * D + p * S + s] /= cnt[p * C + c];
      float sum = 0;  // loss (:113-124); the reference's omp reduction order is unpinned
      for (size_t i = 0; i < n * P; ++i) sum += dist[i];
      cur = sum;
      run--;
    } while ((std::fabs(last - cur) > 0.005f) && (run > 0));
  } while (step < C);
}

static void vq_generate(uint S, uint C, const std::vector<const float*>& segs, float* cen /*C*S*/) {
  const size_t n = segs.size();
  std::vector<uint8_t> mapping(n, 0);
  std::vector<float> dist(n, 0.f);
  std::fill(cen, cen + (size_t)C * S, 0.f);
  {
    std::vector<float> avg(S, 0.f);
    for (size_t i = 0; i < n; ++i) for (uint d = 0; d < S; ++d) avg[d] += segs[i][d];
    for (uint d = 0; d < S; ++d) cen[d] = avg[d] / (float)n;  // n==0 -> NaN, zeroed by the first M step below
  }
  uint step = 1;
  float cur = 0, last = 0;
  auto estep = [&]() {
    for (size_t i = 0; i < n; ++i) {
      uint best = 0; float bd = HUGE_VALF;
      for (uint c = 0; c < step; ++c) {
        const float d = sqdist_seq(segs[i], cen + (size_t)c * S, S);
        if (d < bd) { bd = d; best = c; }
      }
      mapping[i] = (uint8_t)best; dist[i] = bd;
    }
  };
  auto mstep = [&]() {
    std::fill(cen, cen + (size_t)C * S, 0.f);
    std::vector<float> cnt(C, 0.f);
    for (size_t i = 0; i < n; ++i) { const uint c = mapping[i]; for (uint s = 0; s < S; ++s) cen[(size_t)c * S + s] += segs[i][s]; cnt[c] += 1.f; }
    for (uint c = 0; c < C; ++c) if (cnt[c] != 0) for (uint s = 0; s < S; ++s) cen[(size_t)c * S + s] /= cnt[c];
  };
  if (C == 1) { estep(); mstep(); return; }
  do {
    for (uint i = 0; i < step; ++i)
      for (uint d = 0; d < S; ++d) {
        cen[(size_t)(i + step) * S + d] = cen[(size_t)i * S + d] + 0.001f;
        cen[(size_t)i * S + d] = cen[(size_t)i * S + d] - 0.001f;
      }
    step *= 2;
    uint guard = 100000;  // the reference has no iteration cap here (vectorquantizer.hpp:137-143)
    do {
      last = cur; estep(); mstep();
      float sum = 0; for (size_t i = 0; i < n; ++i) sum += dist[i];
      cur = sum;
    } while ((std::fabs(last - cur) > 0.005f) && --guard);
  } while (step < C);
}

template <class T> static void wr(std::fstream& f, T v) { f.write(reinterpret_cast<const char*>(&v), sizeof(T)); }
template <class T> static void rd(std::fstream& f, T& v) { f.read(reinterpret_cast<char*>(&v), sizeof(T)); }

}  // namespace

// ==========================================================================================
// C ABI (ctypes)
// ==========================================================================================
extern "C" {

void* pqo_create(uint D, uint P, uint C1, uint C2, uint W, uint LP, unsigned long long heur_keep, int sort_mode) {
  if (!D || !P || !C1 || !C2 || !W || !LP || D % P || D % LP || LP % P || C1 < W || C1 > 256 || C2 > 256) return nullptr;
  Oracle* o = new Oracle();
  o->D = D; o->P = P; o->C1 = C1; o->C2 = C2; o->W = W; o->LP = LP; o->S = D / P; o->SS = D / LP;
  o->base = W * C2;
  o->maxMultiIndex = upow(o->base, P);  // treequantizer.hpp:40-41 (uint wrap)
  o->powers.resize(P);
  for (uint p = 0; p < P; ++p) o->powers[p] = upow(C1 * C2, p);  // :45-49
  o->curId = 0; o->sortMode = sort_mode; o->heurRows = 0;
  o->prepareHeuristic((size_t)heur_keep);
  o->initCtx(o->ctx);
  return o;
}
void pqo_destroy(void* h) { delete (Oracle*)h; }
void pqo_set_sort_mode(void* h, int m) { ((Oracle*)h)->sortMode = m; }
// sensitivity probe: order of the squared-norm sums (see sqdist); the coarse table is recomputed in the new order
void pqo_set_sum_mode(void* h, int m) { Oracle* o = (Oracle*)h; o->sumMode = m; if (!o->cb1.empty()) o->computeLookupTable(); }
unsigned long long pqo_max_multi_index(void* h) { return ((Oracle*)h)->maxMultiIndex; }
// NOT reference behaviour: lifts the uint wrap of (W*C2)^P (treequantizer.hpp:40-41) so that the checker can follow the engine's
// throughput-only "enumerate_beyond_wrap" mode at BASELINE configs[4] (the reference itself enumerates 0 rows there)
void pqo_set_max_multi_index(void* h, unsigned long long v) { ((Oracle*)h)->maxMultiIndex = (size_t)v; }
unsigned long long pqo_heuristic_rows(void* h) { return ((Oracle*)h)->heurRows; }
void pqo_get_heuristic(void* h, uint* out, unsigned long long rows) {
  Oracle* o = (Oracle*)h; memcpy(out, o->heur.data(), std::min((size_t)rows, o->heurRows) * o->P * sizeof(uint));
}
// replace the heuristic prefix (used to feed the oracle the very table the HIP index holds)
void pqo_set_heuristic(void* h, const uint* rows, unsigned long long n) {
  Oracle* o = (Oracle*)h; o->heur.assign(rows, rows + n * o->P); o->heurRows = n; o->mode2d = false;
}
// optional mode: the CUDA library's 2-D anisotropic sequences (p = 4); returns 0 on success
int pqo_build_heuristic_2d(void* h, uint max_cluster) {
  Oracle* o = (Oracle*)h;
  if (o->P != 4 || max_cluster < 2 || max_cluster > 4096) return 1;
  o->prepare2DDistSequence(max_cluster);
  return 0;
}
// the 10 cell orders, [10][65536]
void pqo_get_heuristic_2d(void* h, uint* out) { Oracle* o = (Oracle*)h; memcpy(out, o->seq2d.data(), o->seq2d.size() * sizeof(uint)); }
// the rows one query enumerates in that mode (h_e rows of P digits, digit 0 = 0xffffffff: no bin)
void pqo_rows_2d(void* h, const float* vec, uint h_e, uint* out) {
  Oracle* o = (Oracle*)h; Ctx& c = o->ctx;
  o->l1tables(c, vec); o->segmentInfo(c, vec);
  std::vector<uint> rows; o->rows2D(c, h_e, rows);
  memcpy(out, rows.data(), rows.size() * sizeof(uint));
}
void pqo_set_codebooks(void* h, const float* cb1, const float* cb2) {
  Oracle* o = (Oracle*)h;
  o->cb1.assign(cb1, cb1 + (size_t)o->C1 * o->D);
  o->cb2.assign(cb2, cb2 + (size_t)o->P * o->C1 * o->C2 * o->S);
  o->computeLookupTable();
}
void pqo_get_codebooks(void* h, float* cb1, float* cb2) {
  Oracle* o = (Oracle*)h;
  memcpy(cb1, o->cb1.data(), o->cb1.size() * 4); memcpy(cb2, o->cb2.data(), o->cb2.size() * 4);
}
void pqo_get_coarse(void* h, float* out) { Oracle* o = (Oracle*)h; memcpy(out, o->coarse.data(), o->coarse.size() * 4); }

// ---- treequantizer.hpp:155-177 generate (k-means tree); data row-major n x D
void pqo_train(void* h, const float* data, unsigned long long n) {
  Oracle* o = (Oracle*)h;
  const uint D = o->D, P = o->P, C1 = o->C1, C2 = o->C2, S = o->S;
  std::vector<const float*> vecs(n);
  for (size_t i = 0; i < n; ++i) vecs[i] = data + i * D;
  o->cb1.assign((size_t)C1 * D, 0.f);
  // grouping uses _PQ->_mapping, i.e. the assignment left by the LAST E step (treequantizer.hpp:140-147)
  std::vector<uint8_t> mapping;
  pq_generate(D, P, C1, vecs.data(), n, o->cb1.data(), mapping);
  o->cb2.assign((size_t)P * C1 * C2 * S, 0.f);
#pragma omp parallel for schedule(dynamic) collapse(2)
  for (uint p = 0; p < P; ++p)
    for (uint c = 0; c < C1; ++c) {
      std::vector<const float*> segs;
      for (size_t i = 0; i < n; ++i) if (mapping[i * P + p] == c) segs.push_back(vecs[i] + p * S);
      vq_generate(S, C2, segs, &o->cb2[((size_t)p * C1 + c) * C2 * S]);
    }
  o->computeLookupTable();
}

// ---- treequantizer.hpp:64-66 notify + :212-217 insert, n vectors (ids continue from the current count)
void pqo_insert(void* h, const float* vecs, unsigned long long n) {
  Oracle* o = (Oracle*)h;
  const size_t base = o->curId;
  o->codes.resize((base + n) * o->LP);
  std::vector<uint> binOf(n);
#pragma omp parallel
  {
    Ctx c; o->initCtx(c);
#pragma omp for schedule(static)
    for (long long i = 0; i < (long long)n; ++i) {
      binOf[i] = o->id(c, vecs + (size_t)i * o->D);
      o->prepareReranking(c, &o->codes[(base + i) * o->LP]);
    }
  }
  for (size_t i = 0; i < n; ++i) o->bins[binOf[i]].push_back((uint)(base + i));
  o->curId = base + n;
}
// bin id only (id())
uint pqo_bin_id(void* h, const float* vec) { Oracle* o = (Oracle*)h; return o->id(o->ctx, vec); }

unsigned long long pqo_num_vectors(void* h) { return ((Oracle*)h)->curId; }
unsigned long long pqo_num_bins(void* h) { return ((Oracle*)h)->bins.size(); }
// bins in ascending key order (std::map order): ids[nb], sizes[nb], members concatenated
void pqo_export_bins(void* h, uint* binIds, uint* binSizes, uint* members) {
  Oracle* o = (Oracle*)h; size_t b = 0, m = 0;
  for (auto& kv : o->bins) { binIds[b] = kv.first; binSizes[b] = (uint)kv.second.size(); ++b; for (uint v : kv.second) members[m++] = v; }
}
void pqo_import_bins(void* h, unsigned long long nb, const uint* binIds, const uint* binSizes, const uint* members) {
  Oracle* o = (Oracle*)h; o->bins.clear(); size_t m = 0;
  for (size_t b = 0; b < nb; ++b) { auto& v = o->bins[binIds[b]]; for (uint j = 0; j < binSizes[b]; ++j) v.push_back(members[m++]); }
}
void pqo_export_codes(void* h, uint32_t* out) { Oracle* o = (Oracle*)h; memcpy(out, o->codes.data(), o->codes.size() * 4); }
void pqo_import_codes(void* h, const uint32_t* codes, unsigned long long n) {
  Oracle* o = (Oracle*)h; o->codes.assign(codes, codes + n * o->LP); o->curId = n;
}

// ---- file formats: treequantizer.hpp:699-737 saveTree / :782-837 loadTree
int pqo_save_tree(void* h, const char* path) {
  Oracle* o = (Oracle*)h;
  std::fstream f(path, std::ios_base::out | std::ios_base::binary);
  if (!f.good()) return -1;
  wr<uint>(f, o->D); wr<uint>(f, o->C1); wr<uint>(f, o->C2); wr<uint>(f, o->P); wr<uint>(f, o->W);
  for (float v : o->cb1) wr<float>(f, v);
  for (float v : o->cb2) wr<float>(f, v);  // [p][c1][c2][s] is exactly the reference's loop order
  return 0;
}
int pqo_load_tree(void* h, const char* path) {
  Oracle* o = (Oracle*)h;
  std::fstream f(path, std::ios_base::in | std::ios_base::binary);
  if (!f.good()) return -1;
  uint d, c1, c2, p, w; rd(f, d); rd(f, c1); rd(f, c2); rd(f, p); rd(f, w);
  if (d != o->D || c1 != o->C1 || c2 != o->C2 || p != o->P) return -2;  // W is read but not checked (:802-805)
  o->cb1.resize((size_t)o->C1 * o->D); o->cb2.resize((size_t)o->P * o->C1 * o->C2 * o->S);
  for (float& v : o->cb1) rd(f, v);
  for (float& v : o->cb2) rd(f, v);
  if (!f.good()) return -3;
  o->computeLookupTable();
  return 0;
}
// ---- treequantizer.hpp:745-774 saveBins / :845-893 loadBins
int pqo_save_bins(void* h, const char* path) {
  Oracle* o = (Oracle*)h;
  std::fstream f(path, std::ios_base::out | std::ios_base::binary);
  if (!f.good()) return -1;
  wr<uint>(f, (uint)o->bins.size());
  for (auto& kv : o->bins) { wr<uint>(f, kv.first); wr<uint>(f, (uint)kv.second.size()); for (uint v : kv.second) wr<uint>(f, v); }
  wr<uint>(f, (uint)o->curId); wr<uint>(f, o->LP);
  f.write(reinterpret_cast<const char*>(o->codes.data()), o->curId * o->LP * 4);
  return 0;
}
int pqo_load_bins(void* h, const char* path) {
  Oracle* o = (Oracle*)h;
  std::fstream f(path, std::ios_base::in | std::ios_base::binary);
  if (!f.good()) return -1;
  uint nb = 0; rd(f, nb);
  o->bins.clear(); size_t cnt = 0;
  for (uint i = 0; i < nb; ++i) {
    uint id = 0, sz = 0; rd(f, id); rd(f, sz);
    auto& v = o->bins[id];
    for (uint j = 0; j < sz; ++j) { uint x = 0; rd(f, x); v.push_back(x); ++cnt; }
  }
  uint len = 0, lp = 0; rd(f, len); rd(f, lp);
  if (lp != o->LP) return -2;
  if (len != cnt) return -3;
  o->codes.resize(cnt * o->LP);
  f.read(reinterpret_cast<char*>(o->codes.data()), cnt * o->LP * 4);
  if (!f.good()) return -4;
  o->curId = cnt;
  return 0;
}

// ---- stage-level outputs for one query vector
void pqo_stage_l1(void* h, const float* vec, float* L1virt, float* L1, uint* L1order) {
  Oracle* o = (Oracle*)h; o->l1tables(o->ctx, vec);
  memcpy(L1virt, o->ctx.L1virt.data(), o->ctx.L1virt.size() * 4);
  memcpy(L1, o->ctx.L1.data(), o->ctx.L1.size() * 4);
  memcpy(L1order, o->ctx.L1order.data(), o->ctx.L1order.size() * 4);
}
// runs l1tables + segmentInfo; outputs arrays of P*W*C2
void pqo_stage_segments(void* h, const float* vec, uint* segL1, uint* segL2, float* segD1, float* segD2, uint* segOrder) {
  Oracle* o = (Oracle*)h; Ctx& c = o->ctx; o->l1tables(c, vec); o->segmentInfo(c, vec);
  const size_t n = c.segOrder.size();
  memcpy(segL1, c.segL1.data(), n * 4); memcpy(segL2, c.segL2.data(), n * 4);
  memcpy(segD1, c.segD1.data(), n * 4); memcpy(segD2, c.segD2.data(), n * 4); memcpy(segOrder, c.segOrder.data(), n * 4);
}
// runs through orderBins; returns number of bins; binIds/binDists in heuristic order, seqOrder = visiting order
unsigned long long pqo_stage_bins(void* h, const float* vec, uint Bb, int sortBins, uint* binIds, float* binDists, uint* seqOrder) {
  Oracle* o = (Oracle*)h; Ctx& c = o->ctx; o->l1tables(c, vec); o->segmentInfo(c, vec);
  std::vector<std::pair<uint, float>> bc; std::vector<uint> so;
  o->orderBins(c, Bb, bc, so, sortBins != 0);
  for (size_t i = 0; i < bc.size(); ++i) { binIds[i] = bc[i].first; binDists[i] = bc[i].second; seqOrder[i] = so[i]; }
  return bc.size();
}
// full query; returns candidate count n; writes the first min(n,cap) (id, dist) pairs of the sorted list
unsigned long long pqo_query(void* h, const float* vec, uint Bv, uint Bb, int sortBins, uint* outIds, float* outDist, unsigned long long cap) {
  Oracle* o = (Oracle*)h;
  std::vector<std::pair<uint, float>> out;
  o->query(o->ctx, Bv, Bb, vec, out, sortBins != 0);
  const size_t n = std::min((size_t)cap, out.size());
  for (size_t i = 0; i < n; ++i) { outIds[i] = out[i].first; outDist[i] = out[i].second; }
  return out.size();
}
// unsorted candidate list in visiting order (what rerankVectors pushes before its final sort)
unsigned long long pqo_query_unsorted(void* h, const float* vec, uint Bv, uint Bb, uint* outIds, float* outDist, unsigned long long cap) {
  Oracle* o = (Oracle*)h; Ctx& c = o->ctx;
  std::vector<std::pair<uint, float>> bc; std::vector<uint> so;
  o->l1tables(c, vec); o->segmentInfo(c, vec); o->orderBins(c, Bb, bc, so, true);
  size_t n = 0; uint used = 0; bool stop = false;
  for (size_t b = 0; b < bc.size(); ++b) {
    auto it = o->bins.find(bc[so[b]].first);
    if (it != o->bins.end())
      for (uint v : it->second) { if (n < cap) { outIds[n] = v; outDist[n] = o->distance(c, v); } ++n; ++used; if (used > Bv) stop = true; }
    if (stop) break;
  }
  return n;
}
// batch: QN queries, top-k of the sorted list (padded with id 0xffffffff / dist +inf), nthreads contexts.
// This is the `cpu_baseline` leg: the whole reference query (full candidate sort included) per vector.
void pqo_query_batch(void* h, const float* Q, unsigned long long QN, uint Bv, uint Bb, uint k,
                     uint* outIds, float* outDist, uint* outCount, int nthreads) {
  Oracle* o = (Oracle*)h;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel
  {
    Ctx c; o->initCtx(c);
    std::vector<std::pair<uint, float>> out;
#pragma omp for schedule(dynamic, 16)
    for (long long q = 0; q < (long long)QN; ++q) {
      o->query(c, Bv, Bb, Q + (size_t)q * o->D, out, true);
      const size_t n = std::min((size_t)k, out.size());
      for (size_t i = 0; i < n; ++i) { outIds[q * k + i] = out[i].first; outDist[q * k + i] = out[i].second; }
      for (size_t i = n; i < k; ++i) { outIds[q * k + i] = 0xffffffffu; outDist[q * k + i] = HUGE_VALF; }
      if (outCount) outCount[q] = (uint)out.size();
    }
  }
}

// ---- scalar helpers exposed for the known-answer tests
float pqo_extract_distance(float a, float b, float c, float l) { return extract_distance(a, b, c, l); }
float pqo_calc_ratio(float a, float b, float c) { return calc_ratio(a, b, c); }
unsigned short pqo_lambda_encode(float f) { return lambda_encode(f); }
float pqo_lambda_decode(unsigned short u) { return lambda_decode(u); }
uint32_t pqo_code_pack(uint a, uint b, float l) { return code_pack(a, b, l); }
uint pqo_upow(uint x, uint n) { return upow(x, n); }
int pqo_max_threads() {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

}  // extern "C"
