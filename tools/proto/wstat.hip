// Prototype (not product code): WEIGHT-STATIONARY persistent row-block layer.  C[M][256] = lrelu(A[M][256] W[256][256]^T + b).
// One 256-thread workgroup per CU; each of its 4 waves keeps its 64 output columns of W (64 x 256 floats = 256 VGPRs per
// lane) in REGISTERS for the whole launch, so the K loop of a tile issues no weight loads at all: per 8 (BM = 32) or 16
// (BM = 64) MFMAs one ds_read_b128 of activations.  The next tile's rows go global -> LDS by LDS-DMA (no staging
// registers), the previous tile's result (second accumulator set) is finished and stored between the MFMA steps.
// WS_TR = 1: the MFMA computes the TRANSPOSED tile (W as the A operand, activations as the B operand): a lane then holds
// 4 consecutive output columns of one row per register quad -> global_store_dwordx4, bias per register, and the finished
// registers are directly the B operand of v_mfma_f32_4x4x1_16B_f32 riders (skip-head sums without VALU work).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <stdio.h>
#include <type_traits>
#include <utility>
#ifndef WS_BM
#define WS_BM 32
#endif
#ifndef WS_TR
#define WS_TR 1
#endif
#ifndef WS_HEAD
#define WS_HEAD 0   // 1: ride the skip-head sums of the input (A fragments) and of the output (finished registers) as 4x4x1 MFMAs
#endif
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void *lds_vp;
typedef const __attribute__((address_space(1))) void *glb_vp;
typedef const __attribute__((address_space(1))) v4f *gcf4;
typedef const __attribute__((address_space(1))) float *gcf;
typedef __attribute__((address_space(1))) float *gf;
typedef __attribute__((address_space(1))) v4f *gf4;
constexpr int K = 256, BM = WS_BM, TM = BM / 32, P = K + 4, IMG = BM * P, NSTEP = 32;   // 32 steps of 8 k each
constexpr bool TR = WS_TR != 0;

template <int I, int N, typename F>
__device__ __forceinline__ void sfor(F &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    sfor<I + 1, N>(f);
  }
}
__device__ __forceinline__ unsigned lds_off(const float *p) { return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float *)p; }
template <int OFF>
__device__ __forceinline__ void rd128(v4f &d, unsigned addr) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(d) : "v"(addr), "n"(OFF)); }
template <int N> __device__ __forceinline__ void lgkm_wait() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
// MFMA by inline asm: the stationary weights are forced into AccVGPRs ("a"), accumulators and activation fragments into
// architectural VGPRs ("v") - hipcc on its own keeps MFMA sources in VGPRs and parks the overflow in AccVGPRs behind a
// v_accvgpr_read per use.  (The compiler inserts no hazard nops around these: see the s_nop pads at the tile seams.)
// One step (8 k) of one row tile: 4 k-pairs x 2 column tiles in ONE asm statement (between separate asm statements the
// hazard recogniser pads a def -> use pair with s_nop).  WA: weights are the A operand (transposed result).
template <bool WA>
__device__ __forceinline__ void mfma_step(f32x16 &a0, f32x16 &a1, const v4f &w0, const v4f &w1, const v4f &x) {
  if constexpr (WA)
    asm volatile(
        "v_mfma_f32_32x32x2_f32 %0, %2, %10, %0\n\tv_mfma_f32_32x32x2_f32 %1, %6, %10, %1\n\t"
        "v_mfma_f32_32x32x2_f32 %0, %3, %11, %0\n\tv_mfma_f32_32x32x2_f32 %1, %7, %11, %1\n\t"
        "v_mfma_f32_32x32x2_f32 %0, %4, %12, %0\n\tv_mfma_f32_32x32x2_f32 %1, %8, %12, %1\n\t"
        "v_mfma_f32_32x32x2_f32 %0, %5, %13, %0\n\tv_mfma_f32_32x32x2_f32 %1, %9, %13, %1"
        : "+v"(a0), "+v"(a1)
        : "a"(w0.x), "a"(w0.y), "a"(w0.z), "a"(w0.w), "a"(w1.x), "a"(w1.y), "a"(w1.z), "a"(w1.w), "v"(x.x), "v"(x.y), "v"(x.z), "v"(x.w));
  else
    asm volatile(
        "v_mfma_f32_32x32x2_f32 %0, %10, %2, %0\n\tv_mfma_f32_32x32x2_f32 %1, %10, %6, %1\n\t"
        "v_mfma_f32_32x32x2_f32 %0, %11, %3, %0\n\tv_mfma_f32_32x32x2_f32 %1, %11, %7, %1\n\t"
        "v_mfma_f32_32x32x2_f32 %0, %12, %4, %0\n\tv_mfma_f32_32x32x2_f32 %1, %12, %8, %1\n\t"
        "v_mfma_f32_32x32x2_f32 %0, %13, %5, %0\n\tv_mfma_f32_32x32x2_f32 %1, %13, %9, %1"
        : "+v"(a0), "+v"(a1)
        : "a"(w0.x), "a"(w0.y), "a"(w0.z), "a"(w0.w), "a"(w1.x), "a"(w1.y), "a"(w1.z), "a"(w1.w), "v"(x.x), "v"(x.y), "v"(x.z), "v"(x.w));
}
__device__ __forceinline__ void mfma4(v4f &acc, float a, float b) {
  asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ float lrelu(float v) {   // max(v, 0.01 v) without the canonicalising v_max of fmaxf
  float t = 0.01f * v, o;
  asm("v_max_f32 %0, %1, %2" : "=v"(o) : "v"(v), "v"(t));
  return o;
}
template <typename T>
__device__ __forceinline__ T *uni(T *p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (T *)(((unsigned long long)hi << 32) | lo);
}

__global__ __launch_bounds__(256, 1) void k_wstat(const float *__restrict__ A, const float *__restrict__ W, const float *__restrict__ bias,
                                                  float *__restrict__ C, int M, const float *__restrict__ Wh, float *__restrict__ hout) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // two images [BM][P]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int n0 = wave * 64;
  const int nblk = M / BM;
  int blk = blockIdx.x;
  if (blk >= nblk) return;
  const int stride = gridDim.x;

  // ---- stationary weights: wb[tn][s][c] = W[n0 + 32 tn + li][8 s' ...]: step s = 4 g + j covers k = 32 g + 16 lh + 4 j + c
  v4f wb[2][NSTEP];
#pragma unroll
  for (int tn = 0; tn < 2; ++tn)
#pragma unroll
    for (int s = 0; s < NSTEP; ++s)
      wb[tn][s] = *(gcf4)(W + (size_t)(n0 + 32 * tn + li) * K + 32 * (s >> 2) + 16 * lh + 4 * (s & 3));
  // bias: TR -> per register (column (r&3) + 8 (r>>2) + 4 lh of the tile), else per lane (column li)
  v4f bq[2][4];
  float bl[2];
#pragma unroll
  for (int tn = 0; tn < 2; ++tn) {
    bl[tn] = bias[n0 + 32 * tn + li];
#pragma unroll
    for (int q = 0; q < 4; ++q) bq[tn][q] = *(gcf4)(bias + n0 + 32 * tn + 8 * q + 4 * lh);
  }
#if WS_HEAD
  // head riders (Q = 2): input part: this wave sums over ITS quarter of k (steps 8 w .. 8 w + 7): A operand of the 4x4x1
  // MFMA = the activation fragment (lane 4 b + i: row 4 (b % 8) + i, k half b / 8), B operand lane 4 b + j = Wh[j][k] (j < 2)
  float whin[8][4];     // [step in the quarter][c]
  float whout[2][16];   // output part: [tn][r] = Wh_out[j][n(r, lh)], lane 4 b + j
  {
    const int j = lane & 3;
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int k = 32 * ((8 * wave + s) >> 2) + 16 * lh + 4 * ((8 * wave + s) & 3) + c;
        whin[s][c] = j < 2 ? Wh[j * 512 + k] : 0.f;
      }
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + 32 * tn + (r & 3) + 8 * (r >> 2) + 4 * lh;
        whout[tn][r] = j < 2 ? Wh[j * 512 + 256 + n] : 0.f;
      }
  }
  v4f hacc_in[TM], hacc_out[2][TM];   // [set][tm] for the output part (finished one tile later)
#endif

  const unsigned abase = lds_off(lds) + (unsigned)(li * P + 16 * lh) * 4u;
  const unsigned abase1 = abase + (unsigned)IMG * 4u;   // image 1 (BM = 64: beyond the 16-bit offset field)

  // ---- LDS-DMA of one row of a tile: global row (1 KiB) -> image row; lane term added by the hardware
  auto dma_row = [&](const float *src_tile, int img, int r) __attribute__((always_inline)) {
    __builtin_amdgcn_global_load_lds((glb_vp)(src_tile + (size_t)r * K + lane * 4), (lds_vp)(lds + img * IMG + r * P), 16, 0, 0);
  };

  f32x16 acc[2][TM][2];   // [set][tm][tn]

  // ---- finish + store one register quad (tm, tn, q) of a finished set
  auto store_quad = [&](f32x16 (&pv)[TM][2], int pblk, int tm, int tn, int q) __attribute__((always_inline)) {
    if constexpr (TR) {
      v4f x;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float v = pv[tm][tn][4 * q + c];
        v = lrelu(v);
        pv[tm][tn][4 * q + c] = v;   // kept: B operand of the output head rider
        x[c] = v;
      }
      gf base = uni((gf)C + ((size_t)pblk * BM + 32 * tm) * 256 + n0 + 32 * tn + 8 * q);
      *(gf4)(&base[(unsigned)(li * 256 + 4 * lh)]) = x;
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int r = 4 * q + c;
        float v = pv[tm][tn][r];
        v = lrelu(v);
        gf base = uni((gf)C + ((size_t)pblk * BM + 32 * tm + (r & 3) + 8 * (r >> 2)) * 256 + n0 + 32 * tn);
        base[(unsigned)(4 * lh * 256 + li)] = v;
      }
    }
  };

  // one tile: K loop into set `ac` from image IMGI; previous tile `pv` is finished and stored, next tile's rows are fetched
  auto block = [&](auto has_prev, auto imgc, f32x16 (&ac)[TM][2], f32x16 (&pv)[TM][2], int nxt, int pblk) __attribute__((always_inline)) {
    constexpr bool HP = decltype(has_prev)::value;
    constexpr int IM = decltype(imgc)::value;
    asm volatile("s_barrier" ::: "memory");
    const float *nsrc = uni(A + (size_t)nxt * BM * K);
    // accumulator init = bias
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r) ac[tm][tn][r] = TR ? bq[tn][r >> 2][r & 3] : bl[tn];
#if WS_HEAD
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) hacc_in[tm] = v4f{0.f, 0.f, 0.f, 0.f};
#endif
    const unsigned ab = (IM == 1 && BM == 64) ? abase1 : abase;
    constexpr int IOFF = (IM == 1 && BM == 32) ? IMG * 4 : 0;
    v4f af[2][TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) rd128<IOFF>(af[0][tm], ab + (unsigned)(tm * 32 * P * 4));
    sfor<0, NSTEP>([&](auto sc) __attribute__((always_inline)) {
      constexpr int s = decltype(sc)::value;
      if constexpr (s + 1 < NSTEP) {
        constexpr int s1 = s + 1;
        constexpr int off = IOFF + (s1 >> 2) * 128 + (s1 & 3) * 16;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) rd128<off>(af[s1 & 1][tm], ab + (unsigned)(tm * 32 * P * 4));
        lgkm_wait<TM>();
      } else {
        lgkm_wait<0>();
      }
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) asm volatile("" : "+v"(af[s & 1][tm]));
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) mfma_step<TR>(ac[tm][0], ac[tm][1], wb[0][s], wb[1][s], af[s & 1][tm]);
#if WS_HEAD
      // input-part head rider on the first 8 steps (product: each wave walks K rotated by its quarter, so these ARE its
      // quarter of k and the four waves split the head's K range without any wave-dependent control flow)
      if constexpr (s < 8) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int tm = 0; tm < TM; ++tm) mfma4(hacc_in[tm], af[s & 1][tm][c], whin[s & 7][c]);
      }
#endif
      // ---- side work of step s
      if constexpr (HP) {
        // TM * 8 quads per tile: one per step from step 0
        if constexpr (s < TM * 8) {
          constexpr int tm = s / 8, tn = (s >> 2) & 1, q = s & 3;
          store_quad(pv, pblk, tm, tn, q);
#if WS_HEAD
          if constexpr (TR) {
            if constexpr (q == 0 && tn == 0) hacc_out[IM ^ 1][tm] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 4; ++c) mfma4(hacc_out[IM ^ 1][tm], whout[tn][4 * q + c], pv[tm][tn][4 * q + c]);
          }
#endif
        }
      }
      // next tile's rows: BM / 4 rows per wave, one per step from step 8 TM
      if constexpr (s >= 8 * TM && s < 8 * TM + BM / 4) {
        constexpr int u = s - 8 * TM;
        dma_row(nsrc, IM ^ 1, wave + 4 * u);
      }
    });
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");   // MFMA results -> VALU / VMEM readers (no compiler hazard handling around asm)
#if WS_HEAD
    // head sums of this tile's input part and of the previous tile's output part: lanes j < 2 hold (row 4 (b % 8) + i, q = j)
    if (hout) {
      const int j = lane & 3, b = lane >> 2;
      if (j < 2) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            // planes: [wave][k half] ; (not the product layout: enough to price the rider)
            hout[(((size_t)(wave * 2 + (b >> 3)) * M) + (size_t)blk * BM + 32 * tm + 4 * (b & 7) + i) * 2 + j] = hacc_in[tm][i];
            if (HP) hout[(((size_t)(8 + wave * 2 + (b >> 3)) * M) + (size_t)pblk * BM + 32 * tm + 4 * (b & 7) + i) * 2 + j] = hacc_out[IM ^ 1][tm][i];
          }
      }
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's rows of the next image have landed
  };
  using T = std::true_type;
  using F = std::false_type;
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  // first image (once per workgroup)
#pragma unroll
  for (int u = 0; u < BM / 4; ++u) dma_row(A + (size_t)blk * BM * K, 0, wave + 4 * u);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  int nxt = blk + stride < nblk ? blk + stride : blk;
  block(F(), I0(), acc[0], acc[1], nxt, 0);
  int prv = blk, set = 1;
  blk += stride;
#pragma unroll 1
  while (blk < nblk) {
    nxt = blk + stride < nblk ? blk + stride : blk;
    block(T(), I1(), acc[1], acc[0], nxt, prv);
    prv = blk; blk += stride; set = 0;
    if (blk >= nblk) break;
    nxt = blk + stride < nblk ? blk + stride : blk;
    block(T(), I0(), acc[0], acc[1], nxt, prv);
    prv = blk; blk += stride; set = 1;
  }
  if (set == 1) {
#pragma unroll
    for (int e = 0; e < TM * 8; ++e) store_quad(acc[0], prv, e / 8, (e >> 2) & 1, e & 3);
  } else {
#pragma unroll
    for (int e = 0; e < TM * 8; ++e) store_quad(acc[1], prv, e / 8, (e >> 2) & 1, e & 3);
  }
}

extern "C" int proto_rowblock(const float *A, const float *W, const float *bias, float *C, int M, int Kk, void *stream) {
  if (M % BM || Kk != K) return -1;
  static int ncu = 0;
  const int lds_bytes = 2 * IMG * 4;
  if (!ncu) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, 0) != hipSuccess) return -2;
    ncu = p.multiProcessorCount;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_wstat), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) return -3;
  }
  int grid = ncu;
  if (const char *e = getenv("WS_GRID")) grid = atoi(e);
  if (grid > M / BM) grid = M / BM;
  static float *wh = nullptr, *hout = nullptr;
#if WS_HEAD
  if (!wh) {
    if (hipMalloc(&wh, 2 * 512 * 4) != hipSuccess || hipMemset(wh, 0, 2 * 512 * 4) != hipSuccess) return -4;
    if (hipMalloc(&hout, (size_t)16 * M * 2 * 4) != hipSuccess) return -5;
  }
#endif
  hipLaunchKernelGGL(k_wstat, dim3(grid), dim3(256), lds_bytes, (hipStream_t)stream, A, W, bias, C, M, wh, hout);
  return (int)hipGetLastError();
}
