// Prototype (not product code): "row-block stationary" fp32 MFMA layer  C[M][256] = lrelu(A[M][K] * W[256][K]^T + b).
// One 256-thread workgroup owns 64 rows x ALL 256 output columns: the 64 x K activation block is put in LDS once
// (LDS-DMA, one 1-KiB row per wave-instruction), the weights go global -> registers directly as MFMA B fragments
// (lane (li, lh) of column tile tn holds W[n0 + 32 tn + li][32 g + 16 lh + 0..15]: 64 contiguous bytes per lane, the two
// lane halves complete the 128-byte line), no LDS staging of B, no barrier inside the K loop, one wave per SIMD.
// K-pairing of an MFMA k-step: lanes lh=0 supply k = 32g + 4j + c, lanes lh=1 k = 32g + 16 + 4j + c (A and B agree).
// Purpose: decide whether the critic / encoder chains should be built on this structure.  M % 64 == 0, K % 32 == 0.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void *lds_vp;
typedef const __attribute__((address_space(1))) void *glb_vp;
typedef const __attribute__((address_space(1))) f32x4v *gcf4;
typedef __attribute__((address_space(1))) float *gf;

#ifndef KMAX
#define KMAX 256
#endif
constexpr int BM = 64, P = KMAX + 4;   // pitch: P/4 odd -> conflict-free ds_read_b128 of 32 rows at one k

__device__ __forceinline__ unsigned lds_off(const float *p) { return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float *)p; }
template <int OFF>
__device__ __forceinline__ void rd128(f32x4v &d, unsigned addr) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(d) : "v"(addr), "n"(OFF)); }
template <int N> __device__ __forceinline__ void lgkm_wait() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }

__global__ __launch_bounds__(256, 1) void k_rowblock(const float *__restrict__ A, const float *__restrict__ W, const float *__restrict__ bias,
                                                      float *__restrict__ C, int M, int K) {
  __shared__ __attribute__((aligned(16))) float lds[BM * P];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int r0 = blockIdx.x * BM;
  // ---- activation block -> LDS: row r of the block = one 1-KiB piece (K = 256) per wave-instruction
#ifndef NOLOAD
  for (int r = wave; r < BM; r += 4) {
    for (int kq = 0; kq < K; kq += 256)
      __builtin_amdgcn_global_load_lds((glb_vp)(A + (size_t)(r0 + r) * K + kq + lane * 4), (lds_vp)(lds + r * P + kq), 16, 0, 0);
  }
#endif
  // ---- B fragments: direct global loads, double-buffered over 32-k groups
  const int n0 = wave * 64;
  gcf4 wp[2];
#pragma unroll
  for (int tn = 0; tn < 2; ++tn) wp[tn] = (gcf4)(W + (size_t)(n0 + 32 * tn + li) * K + 16 * lh);
  f32x4v b[2][2][4];   // [buffer][tn][j]
  auto load_b = [&](int buf, int g) __attribute__((always_inline)) {
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int j = 0; j < 4; ++j) b[buf][tn][j] = wp[tn][8 * g + j];
  };
  load_b(0, 0);
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // the activation rows have landed (the 8 B loads may still fly)
#ifdef NOLOAD
  for (int e = tid; e < BM * P; e += 256) lds[e] = 1.0f;
#endif
  __syncthreads();
  const unsigned abase = lds_off(lds) + (unsigned)(li * P + 16 * lh) * 4;
  const int G = K / 32;
  auto group = [&](int buf, int g) __attribute__((always_inline)) {
    const unsigned ag = abase + (unsigned)g * 128, ag1 = ag + 32 * P * 4;
    f32x4v a[2][2];   // [parity of j][tm]
    rd128<0>(a[0][0], ag);
    rd128<0>(a[0][1], ag1);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j < 3) {
        if (j == 0) { rd128<16>(a[1][0], ag); rd128<16>(a[1][1], ag1); }
        if (j == 1) { rd128<32>(a[0][0], ag); rd128<32>(a[0][1], ag1); }
        if (j == 2) { rd128<48>(a[1][0], ag); rd128<48>(a[1][1], ag1); }
        lgkm_wait<2>();
      } else {
        lgkm_wait<0>();
      }
      asm volatile("" : "+v"(a[j & 1][0]), "+v"(a[j & 1][1]));
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
          for (int tn = 0; tn < 2; ++tn)
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j & 1][tm][c], b[buf][tn][j][c], acc[tm][tn], 0, 0, 0);
    }
    asm volatile("" ::"v"(ag), "v"(ag1));
  };
  for (int g = 0; g < G; g += 2) {   // G even; the last iteration re-requests group G-2 (harmless) to stay branch-free
    load_b(1, g + 1);
    group(0, g);
    load_b(0, g + 2 < G ? g + 2 : g);
    group(1, g + 1);
  }
  // ---- epilogue
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      const int col = n0 + 32 * tn + li;
      const float bv = bias ? bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = r0 + 32 * tm + (r & 3) + 8 * (r >> 2) + 4 * lh;
        float x = acc[tm][tn][r] + bv;
        x = x > 0.f ? x : 0.01f * x;
#ifdef NOSTORE
        if (x == 12345.678f)
#endif
        ((gf)C)[(size_t)row * 256 + col] = x;
      }
    }
}

extern "C" int proto_rowblock(const float *A, const float *W, const float *bias, float *C, int M, int K, void *stream) {
  if (M % BM || K % 64 || K > KMAX || K % 256) return -1;
  hipLaunchKernelGGL(k_rowblock, dim3(M / BM), dim3(256), 0, (hipStream_t)stream, A, W, bias, C, M, K);
  return (int)hipGetLastError();
}
