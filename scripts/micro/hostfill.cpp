// hostfill.cpp -- how fast can the host write the padded result arrays of queryKNN (2 x 67 MB at QN = 4096, nVec = 4096)?
// std::fill vs non-temporal stores, 1..16 threads.   g++ -O2 -pthread -mavx2 -o hostfill hostfill.cpp
#include <immintrin.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <thread>
#include <vector>

static void fill_plain(uint32_t* p, size_t n, uint32_t v) { std::fill(p, p + n, v); }
static void fill_nt(uint32_t* p, size_t n, uint32_t v) {
  size_t i = 0;
  while (i < n && ((uintptr_t)(p + i) & 31)) p[i++] = v;
  const __m256i x = _mm256_set1_epi32((int)v);
  for (; i + 8 <= n; i += 8) _mm256_stream_si256((__m256i*)(p + i), x);
  for (; i < n; ++i) p[i] = v;
  _mm_sfence();
}
template <class F>
static double run(F f, uint32_t* p, size_t rows, size_t cols, size_t keep, int nt) {
  auto t0 = std::chrono::steady_clock::now();
  auto work = [&](size_t r0, size_t r1) { for (size_t r = r0; r < r1; ++r) f(p + r * cols + keep, cols - keep, 0xffffffffu); };
  std::vector<std::thread> th;
  for (int t = 1; t < nt; ++t) th.emplace_back(work, rows * t / nt, rows * (t + 1) / nt);
  work(0, rows / nt);
  for (auto& x : th) x.join();
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
int main() {
  const size_t rows = 4096, cols = 4096, keep = 721;
  std::vector<uint32_t> v(rows * cols, 1u);
  printf("hardware threads %u\n", std::thread::hardware_concurrency());
  for (int nt : {1, 2, 4, 8, 16, 32}) {
    double a = 1e9, b = 1e9;
    for (int rep = 0; rep < 5; ++rep) { a = std::min(a, run(fill_plain, v.data(), rows, cols, keep, nt)); b = std::min(b, run(fill_nt, v.data(), rows, cols, keep, nt)); }
    const double gb = rows * (cols - keep) * 4 / 1e9;
    printf("%2d threads: std::fill %.3f ms (%.1f GB/s)   non-temporal %.3f ms (%.1f GB/s)\n", nt, a, gb / a * 1e3, b, gb / b * 1e3);
  }
  return 0;
}
