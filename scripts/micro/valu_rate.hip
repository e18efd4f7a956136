// valu_rate.hip -- issue cost (shader cycles per wave64 instruction) of the VALU / LDS instructions the rerank's ADC loop is made of,
// measured with s_memtime around long unrolled runs of 8 independent chains; 1, 2 and 3 wavefronts per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/micro/valu_rate scripts/micro/valu_rate.hip && scripts/micro/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define LOOPS 64

template <int OP>
__global__ void k_rate(unsigned long long* out, float seed, uint32_t useed) {
  __shared__ float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = (float)i;
  __syncthreads();
  f2 a[8]; float s[8]; uint32_t u[8];
  unsigned long long w[8];
  for (int i = 0; i < 8; ++i) { a[i] = f2{seed + i, seed * 2 + i}; s[i] = seed + 3 * i; u[i] = useed * 2654435761u + i * 97u + threadIdx.x * 4; w[i] = ((unsigned long long)u[i] << 32) | (useed + i); }
  f2 b = {seed * 0.5f, seed * 0.25f}; float sb = seed * 0.75f; uint32_t ub = useed | 0x3cu;
  uint32_t addr = (threadIdx.x * 4u) & 0x3ffcu;
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int it = 0; it < LOOPS; ++it) {
#define X(i) \
    if (OP == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[i]) : "v"(sb)); \
    if (OP == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); \
    if (OP == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); \
    if (OP == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b)); \
    if (OP == 4) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(s[i]) : "v"(sb)); \
    if (OP == 5) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[i]) : "v"(ub)); \
    if (OP == 6) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(u[i]) : "v"(ub)); \
    if (OP == 7) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(u[i]) : "v"(ub)); \
    if (OP == 8) asm volatile("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(s[i]) : "v"(u[i])); \
    if (OP == 9) asm volatile("v_add_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(u[i]) : "v"(ub)); \
    if (OP == 10) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(ub)); \
    if (OP == 11) asm volatile("v_cmp_lt_u64 vcc, %0, %1" :: "v"(w[i]), "v"(w[(i + 1) & 7]) : "vcc"); \
    if (OP == 12) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(u[i]) : "v"(u[(i + 1) & 7])); \
    if (OP == 13) asm volatile("ds_read_b32 %0, %1" : "=v"(s[i]) : "v"(addr)); \
    if (OP == 14) asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(u[i]) : "v"(addr)); \
    if (OP == 15) asm volatile("v_bfe_u32 %0, %0, 7, 5" : "+v"(u[i])); \
    if (OP == 16) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(u[i]) : "v"(ub)); \
    if (OP == 17) asm volatile("v_min_u32 %0, %0, %1" : "+v"(u[i]) : "v"(ub)); \
    if (OP == 18) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(s[i]) : "v"(sb)); \
    if (OP == 19) asm volatile("ds_add_u32 %0, %1" :: "v"(addr), "v"(u[i])); \
    if (OP == 20) asm volatile("v_readlane_b32 s20, %0, 3" :: "v"(u[i]) : "s20"); \
    if (OP == 21) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(u[i]) : "v"(ub));
    REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
  }
  asm volatile("s_waitcnt lgkmcnt(0) vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  float acc = 0; uint32_t ua = 0;
  for (int i = 0; i < 8; ++i) { acc += a[i].x + a[i].y + s[i]; ua ^= u[i] ^ (uint32_t)w[i]; }
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
  if (acc == 123.456f && ua == 77u) out[1] = 1;  // keep the chains alive
}

template <int OP>
__global__ void k_rate_all(unsigned long long* out, float seed, uint32_t useed) {
  f2 a[8]; float s[8]; uint32_t u[8];
  for (int i = 0; i < 8; ++i) { a[i] = f2{seed + i, seed * 2 + i}; s[i] = seed + 3 * i; u[i] = useed * 2654435761u + i * 97u + threadIdx.x * 4; }
  f2 b = {seed * 0.5f, seed * 0.25f}; float sb = seed * 0.75f; uint32_t ub = useed | 0x3cu;
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int it = 0; it < LOOPS; ++it) {
#define X(i) \
    if (OP == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[i]) : "v"(sb)); \
    if (OP == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); \
    if (OP == 5) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[i]) : "v"(ub));
    REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float acc = 0; uint32_t ua = 0;
  for (int i = 0; i < 8; ++i) { acc += a[i].x + a[i].y + s[i]; ua ^= u[i]; }
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (acc == 123.456f && ua == 77u) out[0] = 1;
}

const char* names[] = {"v_add_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_pk_fma_f32", "v_fma_f32", "v_and_b32", "v_lshl_add_u32", "v_and_or_b32", "v_cvt_f32_u32_sdwa",
                       "v_add_u32_sdwa", "v_cndmask_b32", "v_cmp_lt_u64", "v_mov_b32_dpp", "ds_read_b32", "ds_bpermute_b32", "v_bfe_u32", "v_perm_b32", "v_min_u32", "v_mul_f32",
                       "ds_add_u32", "v_readlane_b32", "v_add3_u32"};

template <int OP>
void run(unsigned long long* d) {
  for (int wps : {1, 2, 3, 4}) {  // wavefronts per SIMD (one workgroup on one CU)
    unsigned long long h[2] = {0, 0};
    hipMemset(d, 0, 16);
    hipLaunchKernelGGL(k_rate<OP>, dim3(1), dim3(256 * wps), 0, 0, d, 1.5f, 12345u);
    hipLaunchKernelGGL(k_rate<OP>, dim3(1), dim3(256 * wps), 0, 0, d, 1.5f, 12345u);
    hipDeviceSynchronize();
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    const double per = (double)h[0] / (LOOPS * 32.0);
    printf("%-22s %d waves/SIMD: %6.2f cycles per instruction of one wave  (%.2f per SIMD-instruction)\n", names[OP], wps, per, per / wps);
  }
}

// all CUs filled with `wgs` workgroups of 256 threads each (one wavefront per SIMD per workgroup): wgs wavefronts per SIMD, beyond the 4 a
// single workgroup can bring; the slowest wavefront's cycles are reported
template <int OP>
void runMany(unsigned long long* d) {
  for (int wgs : {4, 5, 6, 8}) {
    unsigned long long* dd;
    const int nwg = 256 * wgs;
    hipMalloc((void**)&dd, (size_t)nwg * 8);
    hipMemset(dd, 0, (size_t)nwg * 8);
    hipLaunchKernelGGL(k_rate_all<OP>, dim3(nwg), dim3(256), 0, 0, dd, 1.5f, 12345u);
    hipLaunchKernelGGL(k_rate_all<OP>, dim3(nwg), dim3(256), 0, 0, dd, 1.5f, 12345u);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(nwg);
    hipMemcpy(h.data(), dd, (size_t)nwg * 8, hipMemcpyDeviceToHost);
    unsigned long long mx = 0, sum = 0;
    for (auto v : h) { mx = v > mx ? v : mx; sum += v; }
    const double per = (double)sum / nwg / (LOOPS * 32.0);
    printf("%-22s %d workgroups/CU (= waves/SIMD if all resident): mean %6.2f cycles per instruction of one wave (%.2f per SIMD-instruction), slowest wave %.2f\n", names[OP], wgs, per, per / wgs,
           (double)mx / (LOOPS * 32.0));
    hipFree(dd);
  }
}

int main() {
  unsigned long long* d;
  hipMalloc((void**)&d, 16);
  run<0>(d); run<4>(d); run<18>(d); run<1>(d); run<2>(d); run<3>(d); run<5>(d); run<6>(d); run<7>(d); run<21>(d); run<8>(d); run<9>(d); run<10>(d); run<11>(d); run<12>(d);
  run<15>(d); run<16>(d); run<17>(d); run<20>(d); run<13>(d); run<14>(d); run<19>(d);
  runMany<0>(d); runMany<1>(d); runMany<5>(d);
  return 0;
}
