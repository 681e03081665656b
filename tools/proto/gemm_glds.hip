// Prototype (not product code): C[M][N] = A[M][K] * B[N][K]^T, fp32 MFMA 32x32x2, 64x64 tile, operands staged by LDS-DMA
// (global_load_lds_dwordx4) into a 3-deep LDS ring, fragments by swizzled ds_read_b128.  Interior-only: M,N % 64 == 0,
// K % 16 == 0.  Purpose: decide whether the product GEMM's loader should move to this structure.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#ifndef NBUF
#define NBUF 3
#endif
#ifndef MINB
#define MINB 6
#endif
#ifndef TM
#define TM 1
#endif
#ifndef TN
#define TN 1
#endif
constexpr int BM = 64 * TM, BN = 64 * TN, BK = 16, OPA = BM * BK, OPB = BN * BK;
typedef __attribute__((address_space(3))) void *lds_vp;
typedef const __attribute__((address_space(1))) void *glb_vp;

__device__ __forceinline__ unsigned lds_off(const float *p) { return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float *)p; }
__device__ __forceinline__ void rd128(f32x4 &d, unsigned addr) { asm volatile("ds_read_b128 %0, %1" : "=&v"(d) : "v"(addr)); }
template <int N> __device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__global__ __launch_bounds__(256, MINB) void k_proto(const float *__restrict__ A, const float *__restrict__ B, float *__restrict__ C,
                                                     int M, int N, int K) {
  __shared__ __attribute__((aligned(16))) float lds[NBUF][OPA + OPB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
  const int tiles_n = N / BN, tiles_m = M / BM;
  int tile_m, tile_n;
  {
    const int rem = blockIdx.x, full = (tiles_m >> 3) << 3;
    if (rem < full * tiles_n) {
      const int grp = rem / (8 * tiles_n), r = rem - grp * 8 * tiles_n;
      tile_n = r >> 3; tile_m = grp * 8 + (r & 7);
    } else {
      const int r = rem - full * tiles_n;
      tile_m = full + r / tiles_n; tile_n = r - (r / tiles_n) * tiles_n;
    }
  }
  const int r0 = tile_m * BM, c0 = tile_n * BN;
  // staging: 1-KiB LDS-DMA pieces of 16 rows x 16 floats; wave w takes pieces w, w+4, ...; float4 slot s of row r holds
  // k-group s ^ ((r>>2)&3)
  const int srow = 16 * wave + (lane >> 2);                  // row of piece 0 (pieces are 64 rows apart: same swizzle key)
  const int sg = (lane & 3) ^ ((srow >> 2) & 3);
  const float *ga = A + (size_t)(r0 + srow) * K + 4 * sg;
  const float *gb = B + (size_t)(c0 + srow) * K + 4 * sg;
  const size_t a64 = (size_t)64 * K;
  const int nch = K / BK;
  auto issue = [&](int ch) {
    float *base = lds[ch % NBUF];
#pragma unroll
    for (int q = 0; q < TM; ++q)
      __builtin_amdgcn_global_load_lds((glb_vp)(ga + q * a64 + ch * BK), (lds_vp)(base + (wave + 4 * q) * 256), 16, 0, 0);
#pragma unroll
    for (int q = 0; q < TN; ++q)
      __builtin_amdgcn_global_load_lds((glb_vp)(gb + q * a64 + ch * BK), (lds_vp)(base + OPA + (wave + 4 * q) * 256), 16, 0, 0);
  };
  constexpr int PER = TM + TN;   // LDS-DMA instructions per wave per chunk
  // fragment reads: read j of a chunk takes k-group 2j + lh; component c of it feeds k-step 4j + c (A and B agree on k)
  unsigned fa[TM][2], fb[TN][2];
#pragma unroll
  for (int t = 0; t < TM; ++t) {
    const int r = wm * 32 * TM + t * 32 + li, key = (r >> 2) & 3;
    fa[t][0] = (unsigned)(r * 4 + ((0 + lh) ^ key)) * 16; fa[t][1] = (unsigned)(r * 4 + ((2 + lh) ^ key)) * 16;
  }
#pragma unroll
  for (int t = 0; t < TN; ++t) {
    const int r = wn * 32 * TN + t * 32 + li, key = (r >> 2) & 3;
    fb[t][0] = (unsigned)(r * 4 + ((0 + lh) ^ key)) * 16 + OPA * 4; fb[t][1] = (unsigned)(r * 4 + ((2 + lh) ^ key)) * 16 + OPA * 4;
  }
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
  for (int p = 0; p < NBUF - 1; ++p)
    if (p < nch) issue(p);
  for (int ch = 0; ch < nch; ++ch) {
    const int ahead = min(NBUF - 2, nch - 1 - ch);     // chunks requested beyond this one
    switch (ahead) {
      case 0: vm_wait<0>(); break;
      case 1: vm_wait<PER>(); break;
      case 2: vm_wait<2 * PER>(); break;
      case 3: vm_wait<3 * PER>(); break;
      default: vm_wait<4 * PER>(); break;
    }
    __builtin_amdgcn_s_barrier();
    if (ch + NBUF - 1 < nch) issue(ch + NBUF - 1);
    const unsigned lb = lds_off(lds[ch % NBUF]);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      f32x4 a[TM], b[TN];
#pragma unroll
      for (int t = 0; t < TM; ++t) rd128(a[t], lb + fa[t][j]);
#pragma unroll
      for (int t = 0; t < TN; ++t) rd128(b[t], lb + fb[t][j]);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int t = 0; t < TM; ++t) asm volatile("" : "+v"(a[t]));
#pragma unroll
      for (int t = 0; t < TN; ++t) asm volatile("" : "+v"(b[t]));
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][c], b[tn][c], acc[tm][tn], 0, 0, 0);
    }
  }
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int col = c0 + wn * 32 * TN + tn * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = r0 + wm * 32 * TM + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        C[(size_t)row * N + col] = acc[tm][tn][r];
      }
    }
}

extern "C" int proto_gemm_nt(const float *A, const float *B, float *C, int M, int N, int K, void *stream) {
  if (M % BM || N % BN || K % BK) return -1;
  hipLaunchKernelGGL(k_proto, dim3((M / BM) * (N / BN)), dim3(256), 0, (hipStream_t)stream, A, B, C, M, N, K);
  return (int)hipGetLastError();
}
