// Throughput-bound random 64-byte row gathers over several live buffers: time per wave by XCC.  Is the slow half of the chip a
// property of the buffer (physical placement)?  Build: hipcc --offload-arch=gfx950 -O2 -o xcd_gather xcd_gather.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void gather(const uint4* __restrict__ buf, uint32_t rowMask, uint32_t iters, uint32_t* xcc, unsigned long long* ticks, uint32_t* sink) {
  const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
  uint32_t h = gid * 2654435761u + 12345u;
  uint32_t acc = 0;
  const unsigned long long t0 = wall_clock64();
  for (uint32_t i = 0; i < iters; ++i) {
    uint4 v[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      h = h * 1664525u + 1013904223u;
      const uint4* row = buf + (size_t)((h >> 4) & rowMask) * 4;
#pragma unroll
      for (int p = 0; p < 4; ++p) v[u][p] = row[p];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int p = 0; p < 4; ++p) acc ^= v[u][p].x ^ v[u][p].y ^ v[u][p].z ^ v[u][p].w;
  }
  const unsigned long long t1 = wall_clock64();
  if ((threadIdx.x & 63) == 0) { const uint32_t w = gid >> 6; xcc[w] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf; ticks[w] = t1 - t0; }
  if (acc == 0x12345678u) *sink = acc;
}

int main(int argc, char** argv) {
  const size_t mb = argc > 1 ? atoi(argv[1]) : 64;
  const int nbuf = argc > 2 ? atoi(argv[2]) : 8;
  const size_t padmb = argc > 3 ? atoi(argv[3]) : 0;
  const size_t bytes = mb << 20, rows = bytes / 64;
  uint32_t *xcc, *sink; unsigned long long* ticks;
  const uint32_t G = 256 * 4, waves = G * 4, iters = 200;
  CHK(hipMalloc(&xcc, waves * 4)); CHK(hipMalloc(&ticks, waves * 8)); CHK(hipMalloc(&sink, 4));
  std::vector<void*> bufs(nbuf), pads;
  for (int b = 0; b < nbuf; ++b) { CHK(hipMalloc(&bufs[b], bytes)); CHK(hipMemset(bufs[b], b + 1, bytes)); if (padmb) { void* p; CHK(hipMalloc(&p, padmb << 20)); pads.push_back(p); } }
  for (int b = 0; b < nbuf; ++b) {
    for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL(gather, dim3(G), dim3(256), 0, 0, (const uint4*)bufs[b], (uint32_t)(rows - 1), iters, xcc, ticks, sink); CHK(hipDeviceSynchronize()); }
    std::vector<uint32_t> hx(waves); std::vector<unsigned long long> ht(waves);
    CHK(hipMemcpy(hx.data(), xcc, waves * 4, hipMemcpyDeviceToHost)); CHK(hipMemcpy(ht.data(), ticks, waves * 8, hipMemcpyDeviceToHost));
    double sum[16] = {0}; int cnt[16] = {0};
    for (uint32_t i = 0; i < waves; ++i) { sum[hx[i] & 15] += (double)ht[i]; cnt[hx[i] & 15]++; }
    printf("buffer %2d (%zu MB at %p): us per wave by XCC:", b, mb, bufs[b]);
    for (int x = 0; x < 8; ++x) printf(" %d:%.1f", x, cnt[x] ? sum[x] / cnt[x] / 100.0 : 0.0);
    printf("\n");
  }
  return 0;
}
