// Per-XCD latency of dependent random reads over SEVERAL live buffers of one process: is the slow half of the chip a property
// of the buffer (physical placement) or of the process?  Build: hipcc --offload-arch=gfx950 -O2 -o xcd_latency2 xcd_latency2.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void chase(const uint32_t* __restrict__ buf, uint32_t mask, uint32_t steps, uint32_t* xcc, unsigned long long* ticks, uint32_t* sink) {
  const uint32_t lane = threadIdx.x;
  uint32_t p = (blockIdx.x * 64u + lane) * 2654435761u & mask;
  for (uint32_t i = 0; i < 8; ++i) p = buf[p];
  const unsigned long long t0 = wall_clock64();
  for (uint32_t i = 0; i < steps; ++i) p = buf[p];
  const unsigned long long t1 = wall_clock64();
  if (lane == 0) { xcc[blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf; ticks[blockIdx.x] = t1 - t0; }
  if (p == 0xffffffffu) *sink = p;
}

int main(int argc, char** argv) {
  const size_t mb = argc > 1 ? atoi(argv[1]) : 64;
  const int nbuf = argc > 2 ? atoi(argv[2]) : 8;
  const size_t n = mb * 1024 * 1024 / 4, lines = n / 16;
  std::vector<uint32_t> h(n), perm(lines);
  for (size_t i = 0; i < lines; ++i) perm[i] = (uint32_t)i;
  uint64_t s = 88172645463325252ull;
  for (size_t i = lines - 1; i > 0; --i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; size_t j = s % (i + 1); std::swap(perm[i], perm[j]); }
  for (size_t i = 0; i < lines; ++i) { const uint32_t a = perm[i], b = perm[(i + 1) % lines]; for (int w = 0; w < 16; ++w) h[(size_t)a * 16 + w] = b * 16 + (uint32_t)((w * 7 + 3) & 15); }
  uint32_t *xcc, *sink; unsigned long long* ticks;
  CHK(hipMalloc(&xcc, 4096 * 4)); CHK(hipMalloc(&ticks, 4096 * 8)); CHK(hipMalloc(&sink, 4));
  std::vector<uint32_t*> bufs(nbuf);
  for (int b = 0; b < nbuf; ++b) { CHK(hipMalloc(&bufs[b], n * 4)); CHK(hipMemcpy(bufs[b], h.data(), n * 4, hipMemcpyHostToDevice)); }
  const uint32_t G = 1024, steps = 2000;
  for (int b = 0; b < nbuf; ++b) {
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(chase, dim3(G), dim3(64), 0, 0, bufs[b], (uint32_t)(n - 1), steps, xcc, ticks, sink); CHK(hipDeviceSynchronize()); }
    std::vector<uint32_t> hx(G); std::vector<unsigned long long> ht(G);
    CHK(hipMemcpy(hx.data(), xcc, G * 4, hipMemcpyDeviceToHost)); CHK(hipMemcpy(ht.data(), ticks, G * 8, hipMemcpyDeviceToHost));
    double sum[16] = {0}; int cnt[16] = {0};
    for (uint32_t i = 0; i < G; ++i) { sum[hx[i] & 15] += (double)ht[i]; cnt[hx[i] & 15]++; }
    printf("buffer %d (%zu MB at %p): ns per dependent read by XCC:", b, mb, (void*)bufs[b]);
    for (int x = 0; x < 8; ++x) printf(" %d:%.0f", x, cnt[x] ? sum[x] / cnt[x] / steps * 10.0 : 0.0);
    printf("\n");
  }
  return 0;
}
