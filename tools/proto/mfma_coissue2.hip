// Micro-benchmark: cost, in MFMA-slot time, of ONE extra instruction of a given kind after every v_mfma_f32_32x32x2_f32
// (1 wave per SIMD, 4 rotating accumulators).  Kinds: none, v_fma, s_add, s_nop, ds_read_b128 (result never awaited
// inside the loop), global_load_dwordx4 (same), global_store_dword, v_accvgpr_read, s_waitcnt (already satisfied).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
template <int KIND>
__global__ __launch_bounds__(256) void k_co(float *out, const float *g, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[4096];
  for (int e = threadIdx.x; e < 4096; e += 256) lds[e] = e;
  __syncthreads();
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = 1.f + threadIdx.x, b = 0.5f;
  float v = threadIdx.x * 0.001f;
  int sreg = 0;
  v4f d = {0, 0, 0, 0};
  const unsigned laddr = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float *)lds + (threadIdx.x & 63) * 16;
  const float *gp = g + threadIdx.x * 4;
  float *op = out + (size_t)blockIdx.x * 256 + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u & 3], 0, 0, 0);
      if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v) : "v"(b));
      if (KIND == 2) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sreg));
      if (KIND == 3) asm volatile("s_nop 0");
      if (KIND == 4) asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(laddr));
      if (KIND == 5) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(gp));
      if (KIND == 6) asm volatile("global_store_dword %0, %1, off" ::"v"(op), "v"(v) : "memory");
      if (KIND == 7) asm volatile("v_accvgpr_read_b32 %0, a0" : "=v"(v));
      if (KIND == 8) asm volatile("s_waitcnt lgkmcnt(15)");
      if (KIND == 9) asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(v));
    }
    if (KIND == 4 || KIND == 5) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  }
  float s = v + sreg + d[0];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  op[0] = s;
}
template <int KIND>
void run(const char *name, float *out, const float *g, int iters) {
  float best = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_co<KIND>, dim3(256), dim3(256), 0, 0, out, g, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double nm = (double)iters * 32;
  static double base = 0;
  const double ns = best * 1e6 / nm;
  if (KIND == 0) base = ns;
  printf("%-22s %.1f ns per MFMA slot (+%.1f ns = +%.1f cycles at 2.35 GHz), %.1f TF\n", name, ns, ns - base, (ns - base) * 2.35,
         1024.0 * nm * 4096 / (best * 1e-3) / 1e12);
}
int main(int argc, char **argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  float *out, *g; hipMalloc(&out, 256 * 256 * 4); hipMalloc(&g, 1 << 20); hipMemset(g, 0, 1 << 20);
  run<0>("none", out, g, iters); run<1>("v_fma_f32", out, g, iters); run<2>("s_add_u32", out, g, iters); run<3>("s_nop 0", out, g, iters);
  run<4>("ds_read_b128", out, g, iters); run<5>("global_load_dwordx4", out, g, iters); run<6>("global_store_dword", out, g, iters);
  run<7>("v_accvgpr_read", out, g, iters); run<8>("s_waitcnt (satisfied)", out, g, iters); run<9>("v_add_f32 dpp", out, g, iters);
  return 0;
}
