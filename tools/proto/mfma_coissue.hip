// Micro-benchmark: does non-MFMA work of a wave hide under ITS OWN MFMAs, or only under another wave's?
// Each wave runs `iters` x 32 v_mfma_f32_32x32x2_f32 (4 rotating accumulators) with NV independent VALU ops after every
// MFMA; blocks of 256 threads (1 wave per SIMD) and of 512 threads (2 waves per SIMD, same total MFMA count per SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NV>
__global__ void k_co(float *out, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = 1.f + threadIdx.x, b = 0.5f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u & 3], 0, 0, 0);
#pragma unroll
      for (int n = 0; n < NV; ++n) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[(u * NV + n) & 7]) : "v"(b));
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NV>
void run(float *out, int iters) {
  for (int threads = 256; threads <= 512; threads += 256) {
    const int it = threads == 256 ? iters : iters / 2;   // same MFMA count per SIMD
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_co<NV>, dim3(256), dim3(threads), 0, 0, out, it);
      hipEventRecord(e1); hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    const double nm = (double)iters * 32;   // MFMAs per SIMD
    printf("NV=%2d VALU per MFMA, %d waves/SIMD: %.3f ms, %.1f ns per MFMA slot, %.1f TF\n", NV, threads / 256, best, best * 1e6 / nm,
           1024.0 * nm * 4096 / (best * 1e-3) / 1e12);
  }
}
int main(int argc, char **argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  float *out; hipMalloc(&out, 256 * 512 * 4);
  run<0>(out, iters); run<2>(out, iters); run<4>(out, iters); run<8>(out, iters); run<12>(out, iters);
  return 0;
}
