// Output-stationary fp32 MFMA kernel for the dense 256 x 256 blocks of the weight gradients (wgrad.h).
//
// dW[n][k] = sum_m G[m][n] X[m][k]: the contraction runs over the ROWS, so a 32-row tile of G and of X (LDS, by LDS-DMA) is
// 16 MFMA k-steps of 2 rows.  Per k-step a wave reads its A operand (G: one ds_read_b64 = 2 of its 64 columns n per lane)
// and its B operand (X: two ds_read_b128 = 8 of the 256 columns k per lane) and issues 16 v_mfma_f32_32x32x2_f32 into 16
// accumulator tiles that live in the AccVGPRs from the first tile of the workgroup to its last: 3 LDS instructions and 2
// LDS-DMA pieces per 1024 MFMA cycles, no epilogue per tile, no VALU work in the loop at all (fp32 MFMA and the vector ALU
// share one issue stream on gfx950: profiles/r02_rowgemm_notes.txt).  The k-split this needs anyway (12 544 rows / 256
// CUs) is the workgroups' own: workgroup j of a problem writes slab j.
// Column assignment: lane li of the A operand holds columns n0 + 2 li + tn (tn = 0, 1), of the B operand columns
// 128 (t / 4) + 4 li + t % 4 (t = 0..7): consecutive lanes read consecutive 8 / 16 bytes (no bank conflicts) and the four
// results a lane holds for t % 4 = 0..3 are consecutive in memory (one 16-byte store).
//
// Riders (WgInst::X2 / G2): the operand registers are, read as 16 blocks of 4 lanes, also operands of
// v_mfma_f32_4x4x1_16b_f32 (block b = lane / 4 multiplies A[lane 4b + i] by B[lane 4b + j] into D register i of lane
// 4b + j): the G registers give dW2[n][a] += G[m][n] X2[m][a] (B = the row's few X2 columns, 4 MFMAs per k-step), the X
// columns give dW3[q][k] += G2[m][q] X[m][k] (A = the row's few G2 columns; wave w takes column groups t = 2w, 2w + 1:
// 2 MFMAs per k-step).  Lane half lh holds the sums over the rows 2 s + lh; the halves are added at the end.  The X2 / G2
// tiles (32 x 8, 32 x 4 floats, zero-padded) are staged through registers one tile ahead, like the images.
#include "wgrad.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <type_traits>
#include <utility>

namespace fdql {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void *lds_vp;
typedef const __attribute__((address_space(1))) void *glb_vp;
typedef __attribute__((address_space(1))) float *gf;
typedef __attribute__((address_space(1))) v4f *gf4;

constexpr int P = WG_N + 4, IMG = WG_BM * P;   // image [32][P]; G images at 0 / IMG, X images at 2 IMG / 3 IMG
constexpr int NKS = WG_BM / 2;                 // k-steps per tile

template <int I, int N, typename F>
__device__ __forceinline__ void sfor(F &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    sfor<I + 1, N>(f);
  }
}
__device__ __forceinline__ unsigned wg_lds_addr(const float *p) { return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float *)p; }
template <int OFF>
__device__ __forceinline__ void wg_rd128(v4f &d, unsigned addr) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(d) : "v"(addr), "n"(OFF)); }
template <int OFF>
__device__ __forceinline__ void wg_rd64(v2f &d, unsigned addr) { asm volatile("ds_read_b64 %0, %1 offset:%2" : "=&v"(d) : "v"(addr), "n"(OFF)); }
template <int N>
__device__ __forceinline__ void wg_lgkm_wait() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ int wg_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <typename T>
__device__ __forceinline__ T *wg_uni(T *p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (T *)(((unsigned long long)hi << 32) | lo);
}

// One k-step (2 rows): 2 x 8 accumulator tiles, accumulators in AccVGPRs.  ZERO: first k-step of a workgroup (C = 0).
#define WG_M(acc, a, b, c) "v_mfma_f32_32x32x2_f32 %" #acc ", %" #a ", %" #b ", " c "\n\t"
#define WG_ROW(a0, a1, a2, a3, a4, a5, a6, a7, ga, Z)                                                                      \
  WG_M(a0, ga, 18, Z(a0)) WG_M(a1, ga, 19, Z(a1)) WG_M(a2, ga, 20, Z(a2)) WG_M(a3, ga, 21, Z(a3))                       \
  WG_M(a4, ga, 22, Z(a4)) WG_M(a5, ga, 23, Z(a5)) WG_M(a6, ga, 24, Z(a6)) WG_M(a7, ga, 25, Z(a7))
#define WG_ACC(x) "%" #x
#define WG_ZERO(x) "0"
template <bool ZERO>
__device__ __forceinline__ void wg_step(f32x16 (&c)[2][8], const v2f &g, const v4f &x0, const v4f &x1) {
  if constexpr (ZERO)
    asm volatile(WG_ROW(0, 1, 2, 3, 4, 5, 6, 7, 16, WG_ZERO) WG_ROW(8, 9, 10, 11, 12, 13, 14, 15, 17, WG_ZERO)
                 : "=&a"(c[0][0]), "=&a"(c[0][1]), "=&a"(c[0][2]), "=&a"(c[0][3]), "=&a"(c[0][4]), "=&a"(c[0][5]), "=&a"(c[0][6]), "=&a"(c[0][7]),
                   "=&a"(c[1][0]), "=&a"(c[1][1]), "=&a"(c[1][2]), "=&a"(c[1][3]), "=&a"(c[1][4]), "=&a"(c[1][5]), "=&a"(c[1][6]), "=&a"(c[1][7])
                 : "v"(g.x), "v"(g.y), "v"(x0.x), "v"(x0.y), "v"(x0.z), "v"(x0.w), "v"(x1.x), "v"(x1.y), "v"(x1.z), "v"(x1.w));
  else
    asm volatile(WG_ROW(0, 1, 2, 3, 4, 5, 6, 7, 16, WG_ACC) WG_ROW(8, 9, 10, 11, 12, 13, 14, 15, 17, WG_ACC)
                 : "+a"(c[0][0]), "+a"(c[0][1]), "+a"(c[0][2]), "+a"(c[0][3]), "+a"(c[0][4]), "+a"(c[0][5]), "+a"(c[0][6]), "+a"(c[0][7]),
                   "+a"(c[1][0]), "+a"(c[1][1]), "+a"(c[1][2]), "+a"(c[1][3]), "+a"(c[1][4]), "+a"(c[1][5]), "+a"(c[1][6]), "+a"(c[1][7])
                 : "v"(g.x), "v"(g.y), "v"(x0.x), "v"(x0.y), "v"(x0.z), "v"(x0.w), "v"(x1.x), "v"(x1.y), "v"(x1.z), "v"(x1.w));
}

// riders: independent accumulators inside a statement; a rider's accumulator is next touched 16 MFMAs later
__device__ __forceinline__ void wg_rider_x(v4f (&r)[2][2], const v2f &g, const v2f &b) {
  asm volatile(
      "v_mfma_f32_4x4x1_16b_f32 %0, %4, %6, %0\n\t"
      "v_mfma_f32_4x4x1_16b_f32 %1, %4, %7, %1\n\t"
      "v_mfma_f32_4x4x1_16b_f32 %2, %5, %6, %2\n\t"
      "v_mfma_f32_4x4x1_16b_f32 %3, %5, %7, %3"
      : "+v"(r[0][0]), "+v"(r[0][1]), "+v"(r[1][0]), "+v"(r[1][1])
      : "v"(g.x), "v"(g.y), "v"(b.x), "v"(b.y));
}
__device__ __forceinline__ void wg_rider_g(v4f (&r)[2], float dz, const v2f &x) {
  asm volatile(
      "v_mfma_f32_4x4x1_16b_f32 %0, %2, %3, %0\n\t"
      "v_mfma_f32_4x4x1_16b_f32 %1, %2, %4, %1"
      : "+v"(r[0]), "+v"(r[1])
      : "v"(dz), "v"(x.x), "v"(x.y));
}
template <int OFF>
__device__ __forceinline__ void wg_rd32(float &d, unsigned addr) { asm volatile("ds_read_b32 %0, %1 offset:%2" : "=&v"(d) : "v"(addr), "n"(OFF)); }

constexpr int X2S = 4 * IMG, G2S = X2S + 2 * WG_BM * 8;   // rider stages behind the images: X2 [2][32][8], G2 [2][32][4] floats
constexpr int LDS_FLOATS = 4 * IMG, LDS_FLOATS_RIDERS = G2S + 2 * WG_BM * 4;

template <bool RIDERS>
__global__ __launch_bounds__(256, 1) void k_wgrad_stat(const WgArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // G images [2][32][P], X images [2][32][P]
  const int tid = threadIdx.x, lane = tid & 63, wave = wg_uni(tid >> 6);
  const int li = lane & 31, lh = lane >> 5, n0 = wave * 64;

  int inst = 0;
  for (int i = 1; i < a.ninst; ++i)
    if ((int)blockIdx.x >= a.wg_first[i]) inst = i;
  inst = wg_uni(inst);
  const int j0 = (int)blockIdx.x - a.wg_first[inst], stride = a.wg_first[inst + 1] - a.wg_first[inst];
  const int nblk = a.blocks_per_inst;
  if (j0 >= nblk) return;
  const WgInst &I = a.inst[inst];   // kernel-argument segment: scalar loads, each field read once
  const float *G = wg_uni(I.G), *X = wg_uni(I.X);

  const unsigned gbase = wg_lds_addr(lds) + (unsigned)(lh * P + n0 + 2 * li) * 4u;             // G[row 2 s + lh][n0 + 2 li ..]
  const unsigned xbase = wg_lds_addr(lds) + (unsigned)(2 * IMG + lh * P + 4 * li) * 4u;        // X[row 2 s + lh][4 li ..], [128 + 4 li ..]

  // one row of a tile: 1 KiB global -> LDS (the hardware adds the lane's 16 bytes on the LDS side).  The row offsets of the
  // wave's eight pieces live in VGPRs (computed once): one scalar base per tile instead of one per row.
  unsigned voff[WG_BM / 4];
#pragma unroll
  for (int u = 0; u < WG_BM / 4; ++u) voff[u] = (unsigned)((wave + 4 * u) * WG_N + lane * 4) * 4u;
  auto dma_row = [&](const float *src_tile, int img_floats, int u) __attribute__((always_inline)) {
    __builtin_amdgcn_global_load_lds((glb_vp)(reinterpret_cast<const char *>(src_tile) + voff[u]), (lds_vp)(lds + img_floats + (wave + 4 * u) * P), 16, 0, 0);
  };

  f32x16 acc[2][8];   // [tn][t]: AccVGPRs for the whole life of the workgroup
  // riders
  const float *X2 = nullptr, *G2 = nullptr;
  int ldx2 = 0, ldg2 = 0;
  if constexpr (RIDERS) { X2 = wg_uni(I.X2); G2 = wg_uni(I.G2); ldx2 = I.ldx2; ldg2 = I.ldg2; }
  v4f rxa[2][2], rga[2];   // [tn][column group a = 4 ag + j], [t = 2 wave + u]
#pragma unroll
  for (int u = 0; u < 2; ++u) { rxa[u][0] = rxa[u][1] = rga[u] = v4f{0.f, 0.f, 0.f, 0.f}; }
  const bool x2_lane = RIDERS && (tid & 7) < I.nx2, g2_lane = RIDERS && tid < 4 * WG_BM && (tid & 3) < I.ng2;
  // staging: thread -> element (row tid / 8, column a = tid % 8) of the X2 tile, stored at [row][2 (a % 4) + a / 4] (a lane's
  // two column groups side by side); (row tid / 4, q = tid % 4) of the G2 tile
  const int x2_at = X2S + (tid >> 3) * 8 + 2 * (tid & 3) + ((tid >> 2) & 1), g2_at = G2S + tid;
  const unsigned x2base = wg_lds_addr(lds) + (unsigned)(X2S + lh * 8 + 2 * (lane & 3)) * 4u;    // X2[row 2 s + lh][j, 4 + j]
  const unsigned g2base = wg_lds_addr(lds) + (unsigned)(G2S + lh * 4 + (lane & 3)) * 4u;        // G2[row 2 s + lh][i]
  const unsigned xpbase = xbase + (unsigned)(128 * (wave >> 1) + 2 * (wave & 1)) * 4u;          // X[row 2 s + lh][t = 2 wave, 2 wave + 1 of this lane]
  auto stage_load = [&](int blk_, float &sx, float &sg, auto rxc, auto rgc) __attribute__((always_inline)) {
    sx = 0.f; sg = 0.f;
    if constexpr (decltype(rxc)::value) { if (x2_lane) sx = X2[(long long)(blk_ * WG_BM + (tid >> 3)) * ldx2 + (tid & 7)]; }
    if constexpr (decltype(rgc)::value) { if (g2_lane) sg = G2[(long long)(blk_ * WG_BM + (tid >> 2)) * ldg2 + (tid & 3)]; }
  };
  auto stage_store = [&](int im, float sx, float sg, auto rxc, auto rgc) __attribute__((always_inline)) {
    if constexpr (decltype(rxc)::value) lds[x2_at + im * WG_BM * 8] = sx;
    if constexpr (decltype(rgc)::value) { if (tid < 4 * WG_BM) lds[g2_at + im * WG_BM * 4] = sg; }
  };

  // One tile from image pair IM; the rows of tile `nxt` are fetched into the other pair during its first 8 k-steps.
  auto tile = [&](auto firstc, auto imgc, auto rxc, auto rgc, int nxt) __attribute__((always_inline)) {
    constexpr bool FIRST = decltype(firstc)::value;
    constexpr int IM = decltype(imgc)::value;
    constexpr bool RX = decltype(rxc)::value, RG = decltype(rgc)::value;
    constexpr int NR = 3 + (RX ? 1 : 0) + (RG ? 2 : 0);   // LDS reads per k-step
    constexpr int IOFF = IM * IMG * 4, XOFF = IM * WG_BM * 8 * 4, GOFF = IM * WG_BM * 4 * 4;
    // every wave is done with the other image pair and this pair has landed (each wave waited for its own pieces)
    asm volatile("s_barrier" ::: "memory");
    const float *ng = wg_uni(G + (long long)nxt * WG_BM * WG_N), *nx = wg_uni(X + (long long)nxt * WG_BM * WG_N);
    float sx, sg;   // the next tile's rider elements: requested ahead of the DMA pieces (vector memory returns in order)
    stage_load(nxt, sx, sg, rxc, rgc);
    v2f ga[2];
    v4f xa[2], xb[2];
    v2f bx[2], xp[2];
    float dz[2];
    wg_rd64<IOFF>(ga[0], gbase);
    wg_rd128<IOFF>(xa[0], xbase);
    wg_rd128<IOFF + 512>(xb[0], xbase);
    if constexpr (RX) wg_rd64<XOFF>(bx[0], x2base);
    if constexpr (RG) { wg_rd32<GOFF>(dz[0], g2base); wg_rd64<IOFF>(xp[0], xpbase); }
    sfor<0, NKS>([&](auto sc) __attribute__((always_inline)) {
      constexpr int s = decltype(sc)::value;
      if constexpr (s + 1 < NKS) {
        constexpr int off = IOFF + (s + 1) * 2 * P * 4;
        wg_rd64<off>(ga[(s + 1) & 1], gbase);
        wg_rd128<off>(xa[(s + 1) & 1], xbase);
        wg_rd128<off + 512>(xb[(s + 1) & 1], xbase);
        if constexpr (RX) wg_rd64<XOFF + (s + 1) * 64>(bx[(s + 1) & 1], x2base);
        if constexpr (RG) { wg_rd32<GOFF + (s + 1) * 32>(dz[(s + 1) & 1], g2base); wg_rd64<off>(xp[(s + 1) & 1], xpbase); }
        wg_lgkm_wait<NR>();
      } else {
        wg_lgkm_wait<0>();
      }
      asm volatile("" : "+v"(ga[s & 1]), "+v"(xa[s & 1]), "+v"(xb[s & 1]));
      wg_step<FIRST && s == 0>(acc, ga[s & 1], xa[s & 1], xb[s & 1]);
      if constexpr (RX) {
        asm volatile("" : "+v"(bx[s & 1]));
        wg_rider_x(rxa, ga[s & 1], bx[s & 1]);
      }
      if constexpr (RG) {
        asm volatile("" : "+v"(dz[s & 1]), "+v"(xp[s & 1]));
        wg_rider_g(rga, dz[s & 1], xp[s & 1]);
      }
      if constexpr (s < 8) {   // two pieces of the next tile per k-step: rows wave + 4 u of G (u < 8) and of X
        dma_row(ng, (IM ^ 1) * IMG, s);
        dma_row(nx, (2 + (IM ^ 1)) * IMG, s);
      }
      if constexpr (s == 11) stage_store(IM ^ 1, sx, sg, rxc, rgc);
      asm volatile("" ::: "memory");
    });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the next pair have landed
  };
  using T = std::true_type;
  using F = std::false_type;
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  auto finish = [&](auto rxc, auto rgc) __attribute__((always_inline)) {
    // ---- this workgroup's partial -> slab j0; the slabs beyond the workgroups of the block are cleared (j0, j0 + per, ...)
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");   // MFMA results -> v_accvgpr_read (no compiler hazard handling around asm)
  #pragma unroll
    for (int tn = 0; tn < 2; ++tn)
  #pragma unroll
      for (int t = 0; t < 8; ++t) asm volatile("" : "+a"(acc[tn][t]));
    float *const dW = wg_uni(I.dW);
    const int ldw = I.ldw;
    // element (n, k): n = n0 + 2 ((r & 3) + 8 (r >> 2) + 4 lh) + tn, k = 128 tq + 4 li + (0..3)
    const unsigned vo = (unsigned)((n0 + 8 * lh) * ldw + 4 * li);
    {
      gf dst = (gf)(dW + (long long)j0 * a.slab_stride);
  #pragma unroll
      for (int tn = 0; tn < 2; ++tn)
  #pragma unroll
        for (int tq = 0; tq < 2; ++tq)
  #pragma unroll
          for (int r = 0; r < 16; ++r) {
            const v4f v = {acc[tn][4 * tq][r], acc[tn][4 * tq + 1][r], acc[tn][4 * tq + 2][r], acc[tn][4 * tq + 3][r]};
            *(gf4)(&dst[vo + (unsigned)((2 * ((r & 3) + 8 * (r >> 2)) + tn) * ldw + 128 * tq)]) = v;
          }
    }
    for (int e = j0 + stride; e < a.nslab; e += stride) {
      gf dst = (gf)wg_uni(dW + (long long)e * a.slab_stride);
      const v4f z = {0.f, 0.f, 0.f, 0.f};
  #pragma unroll
      for (int tn = 0; tn < 2; ++tn)
  #pragma unroll
        for (int tq = 0; tq < 2; ++tq)
  #pragma unroll
          for (int r = 0; r < 16; ++r) *(gf4)(&dst[vo + (unsigned)((2 * ((r & 3) + 8 * (r >> 2)) + tn) * ldw + 128 * tq)]) = z;
    }
    if constexpr (decltype(rxc)::value || decltype(rgc)::value) {
      // rider results: register i of lane 4 b + j; the lane halves (rows 2 s, rows 2 s + 1) are added, half 0 stores
      asm volatile("" : "+v"(rxa[0][0]), "+v"(rxa[0][1]), "+v"(rxa[1][0]), "+v"(rxa[1][1]), "+v"(rga[0]), "+v"(rga[1]));
      const int b8 = (lane >> 2) & 7, j = lane & 3;
      if constexpr (decltype(rxc)::value) {   // dW2[n = n0 + 2 (4 b8 + i) + tn][a = 4 ag + j]
        float *const dW2 = wg_uni(I.dW2);
        const int ldw2 = I.ldw2, nx2 = I.nx2;
  #pragma unroll
        for (int tn = 0; tn < 2; ++tn)
  #pragma unroll
          for (int ag = 0; ag < 2; ++ag)
  #pragma unroll
            for (int i = 0; i < 4; ++i) {
              float v = rxa[tn][ag][i];
              v += __shfl_xor(v, 32);
              const int at = (n0 + 2 * (4 * b8 + i) + tn) * ldw2 + 4 * ag + j;
              if (lh == 0 && 4 * ag + j < nx2) {
                ((gf)(dW2 + (long long)j0 * a.slab_stride))[at] = v;
                for (int e = j0 + stride; e < a.nslab; e += stride) ((gf)(dW2 + (long long)e * a.slab_stride))[at] = 0.f;
              }
            }
      }
      if constexpr (decltype(rgc)::value) {   // dW3[q = i][k = 128 (wave / 2) + 16 b8 + 4 j + 2 (wave % 2) + u]
        float *const dW3 = wg_uni(I.dW3);
        const int ldw3 = I.ldw3, ng2 = I.ng2;
  #pragma unroll
        for (int u = 0; u < 2; ++u)
  #pragma unroll
          for (int i = 0; i < 4; ++i) {
            float v = rga[u][i];
            v += __shfl_xor(v, 32);
            const int at = i * ldw3 + 128 * (wave >> 1) + 16 * b8 + 4 * j + 2 * (wave & 1) + u;
            if (lh == 0 && i < ng2) {
              ((gf)(dW3 + (long long)j0 * a.slab_stride))[at] = v;
              for (int e = j0 + stride; e < a.nslab; e += stride) ((gf)(dW3 + (long long)e * a.slab_stride))[at] = 0.f;
            }
          }
      }
    }
  };

  auto run = [&](auto rxc, auto rgc) __attribute__((always_inline)) {
    // ---- first tile's images (once per workgroup)
    int blk = j0;
    {
      const float *g0 = G + (long long)blk * WG_BM * WG_N, *x0 = X + (long long)blk * WG_BM * WG_N;
      float sx, sg;
      stage_load(blk, sx, sg, rxc, rgc);
#pragma unroll
      for (int u = 0; u < WG_BM / 4; ++u) {
        dma_row(g0, 0, u);
        dma_row(x0, 2 * IMG, u);
      }
      stage_store(0, sx, sg, rxc, rgc);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    int nxt = blk + stride < nblk ? blk + stride : blk;   // the last tile prefetches itself again (nobody reads it)
    tile(T(), I0(), rxc, rgc, nxt);
    blk += stride;
#pragma unroll 1
    while (blk < nblk) {
      nxt = blk + stride < nblk ? blk + stride : blk;
      tile(F(), I1(), rxc, rgc, nxt);
      blk += stride;
      if (blk >= nblk) break;
      nxt = blk + stride < nblk ? blk + stride : blk;
      tile(F(), I0(), rxc, rgc, nxt);
      blk += stride;
    }
    finish(rxc, rgc);
  };
  // workgroup-uniform: one of the four loop bodies (each with its own end: the accumulators never cross a join)
  if constexpr (RIDERS) {
    if (X2 && G2) run(T(), T());
    else if (G2) run(F(), T());
    else if (X2) run(T(), F());
    else run(F(), F());
  } else {
    run(F(), F());
  }
}

}  // namespace

bool wgrad_stat_takes(const GemmProblem &p) {
  if (!plan_switches().wgrad_stat) return false;   // FDQL_WGRAD_STAT=0
  if (p.M != WG_N || p.N != WG_N || p.nseg != 1 || p.ksplit < 1 || p.bias || p.epi != EPI_NONE || p.colsum || p.C2 || p.hf_w || p.fz_h) return false;
  const GemmSeg &s = p.seg[0];
  if (s.a_kc || s.b_kc || s.lda != WG_N || s.ldb != WG_N || s.K % WG_BM || s.K < WG_BM) return false;
  auto aligned = [](const void *q, uintptr_t n) { return (reinterpret_cast<uintptr_t>(q) & (n - 1)) == 0; };
  return aligned(s.A, 16) && aligned(s.B, 16) && aligned(p.C, 4);
}

bool wgrad_stat_from_problems(const GemmProblem *probs, int nprob, int nslab, long long slab_stride, WgArgs &args) {
  if (nprob < 1 || nprob > WG_MAX_INST || nslab < 1) return false;
  memset(&args, 0, sizeof(args));
  const int R = probs[0].seg[0].K;
  for (int i = 0; i < nprob; ++i) {
    const GemmProblem &p = probs[i];
    if (!wgrad_stat_takes(p) || p.seg[0].K != R || p.ksplit != nslab || p.split_stride != slab_stride) return false;
    WgInst I;
    memset(&I, 0, sizeof(I));
    I.G = p.seg[0].A; I.X = p.seg[0].B; I.dW = p.C; I.ldw = p.ldc;
    args.inst[i] = I;
  }
  args.M = R; args.ninst = nprob; args.blocks_per_inst = R / WG_BM;
  args.nslab = nslab; args.slab_stride = slab_stride;
  int dev = 0;
  static int ncu_of[64];
  static std::mutex mu;
  {
    std::lock_guard<std::mutex> lk(mu);
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    if (!ncu_of[dev]) {
      hipDeviceProp_t pr;
      if (hipGetDeviceProperties(&pr, dev) != hipSuccess) return false;
      ncu_of[dev] = pr.multiProcessorCount;
    }
  }
  args.ncu = cu_budget(ncu_of[dev]);
  return wgrad_stat_balance(args);
}

// Workgroups per block: one per CU in all, dealt so that the slowest block finishes as early as possible - a block with
// riders costs more per tile (measured: + 7 % with the narrow-input rider, + 4 % with the narrow-output one), and tiles come
// in whole numbers.  At most one workgroup per slab and per tile.
bool wgrad_stat_balance(WgArgs &args) {
  const int n = args.ninst, tiles = args.blocks_per_inst;
  int cap = tiles < args.nslab ? tiles : args.nslab;
  if (n < 1 || cap < 1 || args.ncu < n) return false;
  int w[WG_MAX_INST];
  double cost[WG_MAX_INST];
  for (int i = 0; i < n; ++i) {
    w[i] = 1;
    cost[i] = 1.0 + (args.inst[i].X2 ? 0.07 : 0.0) + (args.inst[i].G2 ? 0.04 : 0.0);
  }
  auto time_of = [&](int i, int wi) { return cost[i] * ((tiles + wi - 1) / wi); };
  for (int left = args.ncu - n; left > 0; --left) {
    int worst = -1;
    for (int i = 0; i < n; ++i)
      if (w[i] < cap && (worst < 0 || time_of(i, w[i]) > time_of(worst, w[worst]))) worst = i;
    if (worst < 0) break;
    ++w[worst];
  }
  // trim: a workgroup that does not lower its block's tile count only adds a slab to write
  for (int i = 0; i < n; ++i)
    while (w[i] > 1 && (tiles + w[i] - 2) / (w[i] - 1) == (tiles + w[i] - 1) / w[i]) --w[i];
  args.wg_first[0] = 0;
  for (int i = 0; i < n; ++i) args.wg_first[i + 1] = args.wg_first[i] + w[i];
  return true;
}

bool wgrad_stat_add_rider(WgArgs &args, int inst, const GemmProblem &p) {
  if (!plan_switches().wgrad_riders) return false;   // FDQL_WGRAD_RIDERS=0
  if (inst < 0 || inst >= args.ninst) return false;
  WgInst &I = args.inst[inst];
  if (p.nseg != 1 || p.ksplit != args.nslab || p.split_stride != args.slab_stride || p.bias || p.epi != EPI_NONE || p.colsum || p.C2 ||
      p.hf_w || p.fz_h)
    return false;
  const GemmSeg &s = p.seg[0];
  if (s.a_kc || s.b_kc || s.K != args.M) return false;
  auto aligned4 = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 3) == 0; };
  if (!aligned4(s.A) || !aligned4(s.B) || !aligned4(p.C)) return false;
  if (p.M == WG_N && p.N >= 1 && p.N <= 8 && s.A == I.G && s.lda == WG_N && !I.X2) {   // few input columns under the same G
    I.X2 = s.B; I.nx2 = p.N; I.ldx2 = s.ldb; I.dW2 = p.C; I.ldw2 = p.ldc;
    return true;
  }
  if (p.N == WG_N && p.M >= 1 && p.M <= 4 && s.B == I.X && s.ldb == WG_N && !I.G2) {   // few output rows over the same X
    I.G2 = s.A; I.ng2 = p.M; I.ldg2 = s.lda; I.dW3 = p.C; I.ldw3 = p.ldc;
    return true;
  }
  return false;
}
// (the caller re-deals the workgroups with wgrad_stat_balance() once the riders are attached)

double wgrad_stat_flops(const WgArgs &a) {
  double f = 2.0 * a.M * (double)WG_N * WG_N * a.ninst;
  for (int i = 0; i < a.ninst; ++i) f += 2.0 * a.M * (double)WG_N * (a.inst[i].nx2 + a.inst[i].ng2);
  return f;
}

hipError_t wgrad_stat_launch(const WgArgs &a, hipStream_t s) {
  static bool attr[64];
  static std::mutex mu;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
  {
    std::lock_guard<std::mutex> lk(mu);
    if (!attr[dev]) {   // the opt-in to > 64 KiB of dynamic LDS belongs to the (device, function) pair
      e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_wgrad_stat<false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_FLOATS * 4);
      if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_wgrad_stat<true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_FLOATS_RIDERS * 4);
      if (e != hipSuccess) return e;
      attr[dev] = true;
    }
  }
  bool riders = false;
  for (int i = 0; i < a.ninst; ++i) riders = riders || a.inst[i].X2 || a.inst[i].G2;
  if (riders) hipLaunchKernelGGL(k_wgrad_stat<true>, dim3(a.wg_first[a.ninst]), dim3(256), LDS_FLOATS_RIDERS * 4, s, a);
  else hipLaunchKernelGGL(k_wgrad_stat<false>, dim3(a.wg_first[a.ninst]), dim3(256), LDS_FLOATS * 4, s, a);
  return hipGetLastError();
}

}  // namespace fdql
