// Micro-benchmark: cycles per v_mfma_f32_32x32x2_f32 for one wave per SIMD, 4 rotating accumulators, operands in
// registers; variants: (0) bare, (1) + one ds_read_b128 per 8 MFMAs, (2) + one global_load_dwordx4 per 8 MFMAs.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256, 1) void k_rate(const float *g, float *out, unsigned long long *cyc, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  const int tid = threadIdx.x;
  for (int e = tid; e < 8192; e += 256) lds[e] = 0.001f * e;
  __syncthreads();
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  v4f a = {1.f, 2.f, 3.f, 4.f}, b = {0.5f, 0.25f, 0.125f, 1.f};
  const v4f *gp = reinterpret_cast<const v4f *>(g) + tid;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE == 1) { v4f x = *reinterpret_cast<const v4f *>(&lds[((tid * 4 + u * 64 + it) & 8188)]); a[0] += x[0] * 1e-9f; }
      if (MODE == 2) { v4f x = gp[(u * 256 + it * 2048) & 65535]; b[0] += x[0] * 1e-9f; }
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c], b[c], acc[i], 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + tid] = s;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}
int main(int argc, char **argv) {
  float *g, *out; unsigned long long *cyc;
  hipMalloc(&g, 65536 * 16 * 2); hipMemset(g, 0, 65536 * 16 * 2);
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
  const int iters = argc > 1 ? atoi(argv[1]) : 200;
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k_rate<0>, dim3(256), dim3(256), 0, 0, g, out, cyc, iters);
      if (mode == 1) hipLaunchKernelGGL(k_rate<1>, dim3(256), dim3(256), 0, 0, g, out, cyc, iters);
      if (mode == 2) hipLaunchKernelGGL(k_rate<2>, dim3(256), dim3(256), 0, 0, g, out, cyc, iters);
      hipEventRecord(e1); hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
      double mean = 0; for (int i = 0; i < 256; ++i) mean += h[i]; mean /= 256;
      const double n = (double)iters * 64;
      printf("mode %d: %.1f memtime ticks per MFMA, kernel %.3f ms -> %.1f ns per MFMA, %.1f TF chip-wide\n", mode, mean / n, ms,
             ms * 1e6 / n, 1024.0 * n * 4096 / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
