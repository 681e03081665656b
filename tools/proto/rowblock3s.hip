// Prototype (not product code): PERSISTENT, software-pipelined row-block layer.  C[M][256] = lrelu(A[M][256] W[256][256]^T + b).
// One 256-thread workgroup per CU walks 64-row blocks.  Per block, all under the MFMA stream of the K loop:
//   * the NEXT block's activation rows stream global -> LDS (second image, LDS-DMA),
//   * the PREVIOUS block's result (second accumulator set) gets bias + LeakyReLU and is stored, a few stores per k-group.
// Registers: 2 x 64 accumulators + 64 (B double buffer) + A fragments; LDS: 2 x 66.5 KB.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <stdio.h>
#include <type_traits>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void *lds_vp;
typedef const __attribute__((address_space(1))) void *glb_vp;
typedef const __attribute__((address_space(1))) f32x4v *gcf4;
typedef __attribute__((address_space(1))) float *gf;
constexpr int K = 256, BM = 64, P = K + 4, G = K / 32;

__device__ __forceinline__ unsigned lds_off(const float *p) { return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float *)p; }
template <int OFF>
__device__ __forceinline__ void rd128(f32x4v &d, unsigned addr) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(d) : "v"(addr), "n"(OFF)); }
template <int N> __device__ __forceinline__ void lgkm_wait() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }

__global__ __launch_bounds__(256, 1) void k_rowblock3(const float *__restrict__ A, const float *__restrict__ W, const float *__restrict__ bias,
                                                       float *__restrict__ C, int M, int *__restrict__ next_tile) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // two images [64][P]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int n0 = wave * 64;
  gcf4 wp[2];
#pragma unroll
  for (int tn = 0; tn < 2; ++tn) wp[tn] = (gcf4)(W + (size_t)(n0 + 32 * tn + li) * K + 16 * lh);
  const unsigned abase = lds_off(lds) + (unsigned)(li * P + 16 * lh) * 4;
  const int nblk = M / BM;

  f32x4v b[2][2][4];   // [buffer][tn][j]
  auto load_b = [&](int buf, int g) __attribute__((always_inline)) {
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int j = 0; j < 4; ++j) b[buf][tn][j] = wp[tn][8 * g + j];
  };
  auto load_a = [&](int img, int blk) __attribute__((always_inline)) {
    const float *src = A + (size_t)blk * BM * K + lane * 4;
    float *dst = lds + img * (BM * P);
#pragma unroll
    for (int i = 0; i < BM / 4; ++i) {   // constant trip count: the compiler's vmcnt bookkeeping stays exact
      const int r = wave + 4 * i;
      __builtin_amdgcn_global_load_lds((glb_vp)(src + (size_t)r * K), (lds_vp)(dst + r * P), 16, 0, 0);
    }
  };
  f32x16 acc[2][2][2];   // [set][tm][tn]

  // store the 8 values (tm, tn, r = 4 q .. 4 q + 3 for q = 2 h, 2 h + 1) of a finished accumulator set
  // swapped operands: a lane owns ONE batch row (li) of tile tm and 16 features of tile tn (4 runs of 4 consecutive ones)
  v4f bq[2][4];   // bias of this lane's features, [tn][run]
#pragma unroll
  for (int tn = 0; tn < 2; ++tn)
#pragma unroll
    for (int q = 0; q < 4; ++q) bq[tn][q] = bias ? *(const v4f *)(bias + n0 + 32 * tn + 8 * q + 4 * lh) : v4f{0.f, 0.f, 0.f, 0.f};
  auto store_part = [&](f32x16 (&ac)[2][2], int r0, int part) __attribute__((always_inline)) {
    // part 0..7 -> (tm, tn) = (part >> 2, (part >> 1) & 1), half h = part & 1 covers runs 2 h, 2 h + 1
    const int tm = part >> 2, tn = (part >> 1) & 1, h = part & 1;
    const int row = r0 + 32 * tm + li;
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) {
      const int q = 2 * h + qq;
      v4f x;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float t = ac[tm][tn][4 * q + c] + bq[tn][q][c];
        x[c] = fmaxf(t, 0.01f * t);
      }
      *(__attribute__((address_space(1))) v4f *)((gf)C + (size_t)row * 256 + n0 + 32 * tn + 8 * q + 4 * lh) = x;
    }
  };

  auto group = [&](f32x16 (&ac)[2][2], int buf, int g, int img) __attribute__((always_inline)) {
    const unsigned ag = abase + (unsigned)(img * (BM * P) * 4) + (unsigned)g * 128, ag1 = ag + 32 * P * 4;
    f32x4v a[2][2];
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
            ac[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[buf][tn][j][c], a[j & 1][tm][c], ac[tm][tn], 0, 0, 0);
    }
    asm volatile("" ::"v"(ag), "v"(ag1));
  };

  // one block: K loop into set `ac` reading image `img`; the stores of set `pv` (the previous block) trickle along.
  // The image has landed: its LDS-DMA loads are older than B loads this wave has already consumed (in-order vmcnt),
  // for every wave that reaches the barrier.  No fence: nothing here needs the outstanding stores to drain.
  // next block's rows: global -> registers during k-groups 0..3 (4 rows per wave each), registers -> LDS three groups
  // later.  (The LDS-DMA form would be cheaper, but with LDS-DMA in flight the compiler waits vmcnt(0) before every
  // consumer of a B fragment - it cannot tell the DMA from the loads it tracks - which serialises the whole loop.)
  f32x4v stg[4][4];
  auto block = [&](auto has_prev, f32x16 (&ac)[2][2], f32x16 (&pv)[2][2], int img, int next_blk, int prev_r0) __attribute__((always_inline)) {
    asm volatile("s_barrier" ::: "memory");
    const f32x4v *nsrc = reinterpret_cast<const f32x4v *>(A + (size_t)next_blk * BM * K) + lane;
    float *ndst = lds + (img ^ 1) * (BM * P) + lane * 4;
    auto issue = [&](int q) __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < 4; ++u) stg[q][u] = ((gcf4)nsrc)[(size_t)(wave + 4 * (4 * q + u)) * (K / 4)];
    };
    auto commit = [&](int q) __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < 4; ++u) *reinterpret_cast<f32x4v *>(ndst + (wave + 4 * (4 * q + u)) * P) = stg[q][u];
    };
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) ac[i][j][r] = 0.f;
#pragma unroll
    for (int g = 0; g < G; g += 2) {
      load_b(1, g + 1);
      if (g < 4) issue(g);
      if (g >= 3 && g < 7) commit(g - 3);
      group(ac, 0, g, img);
      if (decltype(has_prev)::value) store_part(pv, prev_r0, g);
      load_b(0, g + 2 < G ? g + 2 : 0);     // the last one fetches group 0 for the next block
      if (g + 1 < 4) issue(g + 1);
      if (g + 1 >= 3 && g + 1 < 7) commit(g + 1 - 3);
      group(ac, 1, g + 1, img);
      if (decltype(has_prev)::value) store_part(pv, prev_r0, g + 1);
    }
  };
  using T = std::true_type;
  using F = std::false_type;

  int blk = blockIdx.x;
  if (blk >= nblk) return;
  load_a(0, blk);
  load_b(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // first image landed (once per workgroup)
  const int stride = gridDim.x;
  // the image prefetch of a workgroup's last block re-reads that block (nobody consumes it): keeps the loop branch-free
  block(F(), acc[0], acc[1], 0, blk + stride < nblk ? blk + stride : blk, 0);
  int prev_r0 = blk * BM;
  blk += stride;
  int set = 1;
#pragma unroll 1
  while (blk < nblk) {
    const int nxt = blk + stride < nblk ? blk + stride : blk;
    block(T(), acc[1], acc[0], 1, nxt, prev_r0);
    prev_r0 = blk * BM;
    blk += stride;
    set = 0;
    if (blk >= nblk) break;
    const int nxt2 = blk + stride < nblk ? blk + stride : blk;
    block(T(), acc[0], acc[1], 0, nxt2, prev_r0);
    prev_r0 = blk * BM;
    blk += stride;
    set = 1;
  }
  if (set == 1) {   // the last block's result sits in set 0
#pragma unroll
    for (int p = 0; p < 8; ++p) store_part(acc[0], prev_r0, p);
  } else {
#pragma unroll
    for (int p = 0; p < 8; ++p) store_part(acc[1], prev_r0, p);
  }
}

extern "C" int proto_rowblock(const float *A, const float *W, const float *bias, float *C, int M, int Kk, void *stream) {
  if (M % BM || Kk != K) return -1;
  static int ncu = 0;
  const int lds_bytes = 2 * BM * P * 4;
  if (!ncu) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, 0) != hipSuccess) return -2;
    ncu = p.multiProcessorCount;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rowblock3), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) return -3;
  }
  int grid = ncu;
  if (const char *e = getenv("RB3_GRID")) grid = atoi(e);
  if (grid > M / BM) grid = M / BM;
  hipLaunchKernelGGL(k_rowblock3, dim3(grid), dim3(256), lds_bytes, (hipStream_t)stream, A, W, bias, C, M, nullptr);
  return (int)hipGetLastError();
}
