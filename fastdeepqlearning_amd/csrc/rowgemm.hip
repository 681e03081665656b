// Persistent, software-pipelined row-block fp32 MFMA GEMM for the critic ensemble's dense layers (rowgemm.h).
//
// One 256-thread workgroup per CU walks (instance, 64-row block) tiles.  A tile's 64 x K activation block sits in LDS
// ([64][pitch] image, ds_read_b128 conflict-free), the weights go global -> registers directly as MFMA B fragments (no
// LDS staging of B, no barrier inside the K loop), each of the 4 waves owns 64 of the 256 output columns (2 x 2
// v_mfma_f32_32x32x2_f32 tiles, 64 accumulators).  What a one-tile-per-workgroup kernel does before and after its K loop
// runs UNDER the K loop of the neighbouring tiles here:
//   * the NEXT tile's rows stream global -> registers -> the second LDS image, a few rows per k-group;
//   * the PREVIOUS tile's result waits in a second accumulator set and gets its epilogue (bias, LeakyReLU or its
//     derivative, head-fusion partial sums, column sums) and stores a slice at a time between the k-groups.
// Measured on the prototype (tools/proto/rowblock3.hip, M = 192000, K = N = 256): 115.8 TFLOP/s against 90.8 for the
// 64x64-tile grouped kernel and 101.6 for the same row-block kernel without the overlap.
//
// All memory operations of the loop are straight-line code with compile-time trip counts: with a branch or a loop
// around any of them (or with LDS-DMA loads in flight) hipcc's vmcnt bookkeeping falls back to s_waitcnt vmcnt(0) before
// every consumer of a B fragment, which serialises the loop (profiles/r02_rowgemm_notes.txt).
#include "rowgemm.h"

#include <mutex>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>

namespace fdql {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(1))) v4f *gcf4;
typedef const __attribute__((address_space(1))) v2f *gcf2;
typedef const __attribute__((address_space(1))) float *gcf;
typedef __attribute__((address_space(1))) float *gf;
typedef __attribute__((address_space(1))) v2f *gf2;

constexpr int G = RG_KMAIN / 32;   // 32-k groups of the main segment
// diagnostic: shader cycles and 100 MHz wall ticks of every workgroup's life in the last launch (clock under load)
__device__ unsigned long long g_rg_life[2 * 1024];
#ifdef RG_STAMPS
__device__ unsigned long long g_rg_phase[8 * 1024];   // per workgroup: cycles waiting at the barrier, tile set-up, K loop, tails/roll, tiles
#define RG_T() __builtin_amdgcn_s_memtime()
#endif
constexpr int LD = 256;            // row stride of the main activations, the outputs and the reference (compile-time: the
                                   // 64 store / 16 prefetch addresses of a tile become immediates instead of live registers)

__device__ __forceinline__ unsigned rg_lds_addr(const float *p) {
  return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float *)p;
}
template <int OFF>
__device__ __forceinline__ void rg_rd128(v4f &d, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(d) : "v"(addr), "n"(OFF));
}
template <int N>
__device__ __forceinline__ void rg_lgkm_wait() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ int rg_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <typename T>
__device__ __forceinline__ T *rg_uni(T *p) {   // a pointer read from the instance table: keep it in scalar registers
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (T *)(((unsigned long long)hi << 32) | lo);
}
// The fields of an instance a tile needs, read ONCE per tile: the asm statements of the K loop clobber "memory", so a
// field left in the table would be re-read (by a vector load followed by s_waitcnt vmcnt(0)) at every use.
struct RgPtrs {
  const float *A[1 + RG_MAX_MINOR], *W[1 + RG_MAX_MINOR], *bias, *ref, *hf_w, *fz_h, *fz_w;
  float *C, *C2, *colsum, *hf_out, *hf_out2, *fz_out, *fz_colsum;
};
__device__ __forceinline__ RgPtrs rg_read(const RowGemmArgs &a, int i) {
  const RowGemmInst &I = a.inst[i];   // kernel-argument segment: scalar loads
  RgPtrs r;
#pragma unroll
  for (int s = 0; s < 1 + RG_MAX_MINOR; ++s) { r.A[s] = rg_uni(I.A[s]); r.W[s] = rg_uni(I.W[s]); }
  r.bias = rg_uni(I.bias); r.ref = rg_uni(I.ref); r.hf_w = rg_uni(I.hf_w);
  r.C = rg_uni(I.C); r.C2 = rg_uni(I.C2); r.colsum = rg_uni(I.colsum); r.hf_out = rg_uni(I.hf_out); r.hf_out2 = rg_uni(I.hf_out2);
  r.fz_h = rg_uni(I.fz_h); r.fz_w = rg_uni(I.fz_w); r.fz_out = rg_uni(I.fz_out); r.fz_colsum = rg_uni(I.fz_colsum);
  return r;
}

// Head-fusion partial sums of one 32 x 32 accumulator tile: out[(plane*M + row)*Q + q] = sum over the tile's 32 columns
// of x[row][col] * w[q][col] (GemmProblem::hf_* in common.h).  A lane holds 16 rows of ONE column, so each of the 16*Q
// products is summed over the 32 lanes of its lane half - with DPP adds on the vector ALU (row_shr 1, 2, 4, 8, then
// row_bcast15 into the odd rows): no LDS-crossbar shuffle, hence no lgkmcnt wait stalling the MFMA stream this runs in
// (gemm.hip's butterfly needs 5 dependent ds_bpermute round trips per row group).  Lanes 31 / 63 end up with the sums of
// lane half 0 / 1 and store them.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float rg_dpp(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, ROW_MASK, 0xf, true));
}
__device__ __forceinline__ float rg_sum32(float x) {   // valid in lanes 31 and 63
  x += rg_dpp<0x111, 0xf>(x);   // row_shr:1
  x += rg_dpp<0x112, 0xf>(x);   // row_shr:2
  x += rg_dpp<0x114, 0xf>(x);   // row_shr:4
  x += rg_dpp<0x118, 0xf>(x);   // row_shr:8  -> lane 15 of each 16-lane row holds the row's sum
  x += rg_dpp<0x142, 0xa>(x);   // row_bcast:15 into rows 1 and 3
  return x;
}
template <int Q>
__device__ __forceinline__ void rg_hf_partial(const f32x16 &acc, float bv, const float (&wq)[Q], int lane, gf out, int r_first, int r_count) {
  // out: uniform pointer to the sums of the tile's row 0 in this wave's plane; the lane half adds its 4 rows
  const unsigned vo = (unsigned)(4 * (lane >> 5) * Q);
#pragma unroll
  for (int r = r_first; r < r_first + r_count; ++r) {
    float x = acc[r] + bv;
    x = x > 0.f ? x : 0.01f * x;
    float sum[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) sum[q] = rg_sum32(x * wq[q]);
    if ((lane & 31) == 31) {
      gf dst = &rg_uni(out + ((r & 3) + 8 * (r >> 2)) * Q)[vo];
      if constexpr (Q == 2) {
        *(gf2)dst = v2f{sum[0], sum[1]};
      } else {
#pragma unroll
        for (int q = 0; q < Q; ++q) dst[q] = sum[q];
      }
    }
  }
}

// KS: weights K-strided (dgrad), output column of (tile tn, lane li) = n0 + 2 li + tn; else K-contiguous, n0 + 32 tn + li.
// NMINOR: narrow segments (K <= 8) beside the 256-wide one.  DUAL: two outputs (see rowgemm.h).  HFQ: head-fusion
// outputs per row (0 = off).  GRAD: dgrad epilogue (LeakyReLU' gate from `ref`, column sums) instead of bias + LeakyReLU.
// FUSE (dgrad form, one narrow segment of K = 2 = dY): the main segment's A is formed from (fz_h, dY, fz_w) while it is
// staged (GemmProblem::fz_* in common.h) - the head dgrad of the layer above runs inside this launch's loader.
template <bool KS, int NMINOR, bool DUAL, int HFQ, bool GRAD, bool FUSE = false>
__global__ __launch_bounds__(256, 1) void k_rowgemm(const RowGemmArgs a) {
  static_assert(!FUSE || (GRAD && NMINOR == 1), "the fused head dgrad belongs to the dgrad form with its dY segment");
  static_assert(!DUAL || NMINOR >= 1, "a dual launch emits before its last narrow segment");
  static_assert(!(KS && HFQ), "head fusion belongs to the forward layers");
  static_assert(!(GRAD && DUAL), "one output in the dgrad form");
  extern __shared__ __attribute__((aligned(16))) float lds[];   // two images [64][pitch]
  const int tid = threadIdx.x, lane = tid & 63, wave = rg_uni(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int n0 = wave * 64;
  constexpr int pitch = RG_KMAIN + 8 * NMINOR + 4, img_floats = RG_BM * pitch;   // (pitch / 4) odd: conflict-free ds_read_b128
  const int ntiles = a.ninst * a.blocks_per_inst, bpi = a.blocks_per_inst, M = a.M;

  // pad columns of both images (between a narrow segment's K and its 8-wide slot): never written again
  if constexpr (NMINOR > 0) {
    const int padw = pitch - RG_KMAIN;
    for (int e = tid; e < 2 * RG_BM * padw; e += 256) {
      const int r = e / padw, c = e - r * padw;
      lds[r * pitch + RG_KMAIN + c] = 0.f;   // r runs over the 128 rows of the two images (same pitch)
    }
    __syncthreads();
  }

  float *fzred = lds + 2 * img_floats;   // FUSE: [2][4 waves][256] column-sum partials of the staged tiles
  // narrow segments: which elements of the [64, K] block this thread moves (two per segment at most: 64 * 8 / 256)
  int m_src[NMINOR > 0 ? NMINOR : 1][2], m_dst[NMINOR > 0 ? NMINOR : 1][2];
  bool m_ok[NMINOR > 0 ? NMINOR : 1][2];
#pragma unroll
  for (int s = 0; s < NMINOR; ++s)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int Ks = a.kminor[s], e = tid + 256 * u;
      const int r = e / Ks, c = e - r * Ks;
      m_ok[s][u] = e < RG_BM * Ks;
      m_src[s][u] = m_ok[s][u] ? r * a.lda[1 + s] + c : 0;
      m_dst[s][u] = m_ok[s][u] ? r * pitch + RG_KMAIN + 8 * s + c : RG_KMAIN + 8 * s;
    }

  const unsigned abase = rg_lds_addr(lds) + (unsigned)(li * pitch + 16 * lh) * 4u;
  // Global addresses are split into a UNIFORM base (scalar registers, scalar arithmetic) and a 32-bit per-lane offset
  // that is the same for every row of a tile: the 64 stores / 16 prefetches / reference loads of a tile then need no
  // vector address arithmetic and no live 64-bit address pairs (saddr form of global_load / global_store).
  const unsigned vo_kc[2] = {(unsigned)(4 * lh * LD + n0 + li), (unsigned)(4 * lh * LD + n0 + 32 + li)};   // K-contiguous form: column of tile tn
  const unsigned vo_ks = (unsigned)(4 * lh * LD + n0 + 2 * li);
  const unsigned vo_wks = (unsigned)(16 * lh * LD + n0 + 2 * li);   // K-strided weights: k-row 16 lh of the lane's column pair                                               // K-strided form: column pair
  const int ldw0 = KS ? LD : a.ldw[0];

  // ---- B fragments of the main segment: pointers of the current and of the next tile's instance
  gcf4 wp[2], wpn[2];
  gcf wk = nullptr, wkn = nullptr;
  auto w_pointers = [&](const RgPtrs &I, gcf4 (&p)[2], gcf &k) __attribute__((always_inline)) {
    if constexpr (KS) {
      k = (gcf)I.W[0];   // uniform; lane offset vo_wks
    } else {
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) p[tn] = (gcf4)(I.W[0] + (long long)(n0 + 32 * tn + li) * ldw0 + 16 * lh);
    }
  };
  v4f b[2][2][4];    // K-contiguous: [buffer][tn][j]
  v2f bk[2][4][4];   // K-strided:    [buffer][j][c] = columns (2 li, 2 li + 1)
  // quarter j of the B fragments of k-group g (one of the 4 MFMA steps): the loads are spread under the MFMA steps of
  // the group before, 2 (4 narrow) per step, instead of 8 (16) in a burst that fills the memory pipeline's queue and
  // holds the wave - and with it the MFMA stream - at the issue of the next vector memory instruction
  auto load_b_part = [&](int buf, int g, int j, const gcf4 (&p)[2], gcf k) __attribute__((always_inline)) {
    if constexpr (KS) {
#pragma unroll
      for (int c = 0; c < 4; ++c) bk[buf][j][c] = *(gcf2)(&rg_uni(k + (32 * g + 4 * j + c) * LD)[vo_wks]);
    } else {
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) b[buf][tn][j] = p[tn][8 * g + j];
    }
  };
  auto load_b = [&](int buf, int g, const gcf4 (&p)[2], gcf k) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) load_b_part(buf, g, j, p, k);
  };

  f32x16 acc[2][2][2];         // [set][tm][tn]: the tile being accumulated and the previous one (being stored)
  v4f tp0, tp1;                // DUAL: B fragments of the previous tile's last narrow segment (its second output)
  float bvc[2] = {0.f, 0.f}, bvp[2] = {0.f, 0.f};          // bias of the current / previous tile's columns
  float wqc[2][HFQ > 0 ? HFQ : 1], wqp[2][HFQ > 0 ? HFQ : 1];   // head weights, same
  v2f rv[2][4];                                            // GRAD: reference values of the next / current slice
  float cs[2] = {0.f, 0.f};                                // GRAD: running column sums of the previous tile

  auto group = [&](f32x16 (&ac)[2][2], int buf, int g, int img, auto &&piece) __attribute__((always_inline)) {
    const unsigned ag = abase + (unsigned)(img * img_floats) * 4u + (unsigned)g * 128u, ag1 = ag + (unsigned)(32 * pitch) * 4u;
    v4f af[2][2];
    rg_rd128<0>(af[0][0], ag);
    rg_rd128<0>(af[0][1], ag1);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j < 3) {
        if (j == 0) { rg_rd128<16>(af[1][0], ag); rg_rd128<16>(af[1][1], ag1); }
        if (j == 1) { rg_rd128<32>(af[0][0], ag); rg_rd128<32>(af[0][1], ag1); }
        if (j == 2) { rg_rd128<48>(af[1][0], ag); rg_rd128<48>(af[1][1], ag1); }
        rg_lgkm_wait<2>();
      } else {
        rg_lgkm_wait<0>();
      }
      asm volatile("" : "+v"(af[j & 1][0]), "+v"(af[j & 1][1]));
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
          for (int tn = 0; tn < 2; ++tn) {
            const float bval = KS ? bk[buf][j][c][tn] : b[buf][tn][j][c];
            ac[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j & 1][tm][c], bval, ac[tm][tn], 0, 0, 0);
          }
      piece(j);   // a quarter of the group's side work, free to mix with the 16 MFMAs above
    }
    asm volatile("" ::"v"(ag), "v"(ag1));
  };
  // 8-k step of a narrow segment: lanes lh = 0 / 1 hold k = 0..3 / 4..7
  auto tail_frags = [&](int s, int img, v4f (&af)[2]) __attribute__((always_inline)) {   // A fragments of segment s's step
    const unsigned a0 = abase + (unsigned)(img * img_floats) * 4u + (unsigned)(RG_KMAIN + 8 * s) * 4u - (unsigned)(12 * lh) * 4u;
    const unsigned a1 = a0 + (unsigned)(32 * pitch) * 4u;
    rg_rd128<0>(af[0], a0);
    rg_rd128<0>(af[1], a1);
    rg_lgkm_wait<0>();
    asm volatile("" : "+v"(af[0]), "+v"(af[1]));
    asm volatile("" ::"v"(a0), "v"(a1));
  };
  auto tail_mfma = [&](f32x16 (&ac)[2][2], const v4f (&af)[2], const v4f &t0, const v4f &t1) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      ac[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0][c], t0[c], ac[0][0], 0, 0, 0);
      ac[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0][c], t1[c], ac[0][1], 0, 0, 0);
      ac[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1][c], t0[c], ac[1][0], 0, 0, 0);
      ac[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1][c], t1[c], ac[1][1], 0, 0, 0);
    }
  };
  auto tail = [&](f32x16 (&ac)[2][2], int s, int img, const v4f &t0, const v4f &t1) __attribute__((always_inline)) {
    v4f af[2];
    tail_frags(s, img, af);
    tail_mfma(ac, af, t0, t1);
  };
  // B fragments of a narrow segment's step (guarded: k < K)
  auto load_tail = [&](const RgPtrs &I, int s, v4f &t0, v4f &t1) __attribute__((always_inline)) {
    const int Ks = a.kminor[s], ldw = a.ldw[1 + s];
    const float *W = I.W[1 + s];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int k = 4 * lh + c;
      const int kc = k < Ks ? k : Ks - 1;   // unconditional (clamped) loads: no branch in the memory stream
      if constexpr (KS) {
        const v2f x = *(gcf2)((gcf)W + (long long)kc * ldw + n0 + 2 * li);
        t0[c] = k < Ks ? x.x : 0.f;
        t1[c] = k < Ks ? x.y : 0.f;
      } else {
        const float x0 = ((gcf)W)[(long long)(n0 + li) * ldw + kc], x1 = ((gcf)W)[(long long)(n0 + 32 + li) * ldw + kc];
        t0[c] = k < Ks ? x0 : 0.f;
        t1[c] = k < Ks ? x1 : 0.f;
      }
    }
  };

  // ---- one slice (1/8) of a finished tile: epilogue + stores.  K-contiguous form: slice p = (tm, tn, half of the 16
  // rows); K-strided form: p = (tm, 4 rows) x both columns of the lane's pair (8-byte stores).
  // `first`, `count`: which of the slice's 8 values (K-contiguous form) / 4 value pairs (K-strided form)
  auto store_vals = [&](f32x16 (&pv)[2][2], const RgPtrs &I, bool second, int pblk, int p, int first, int count, const v2f (&rvs)[4]) __attribute__((always_inline)) {
    const int r0 = pblk * RG_BM;
    if constexpr (KS) {
      const int tm = p >> 2, q4 = p & 3;
      gf C = (gf)I.C;
#pragma unroll
      for (int i = first; i < first + count; ++i) {
        const int r = 4 * q4 + i;
        const int rowu = r0 + 32 * tm + (r & 3) + 8 * (r >> 2);   // + 4 lh: in the lane offset
        v2f x = {pv[tm][0][r], pv[tm][1][r]};
        if constexpr (GRAD) {
          x.x = rvs[i].x > 0.f ? x.x : 0.01f * x.x;
          x.y = rvs[i].y > 0.f ? x.y : 0.01f * x.y;
          cs[0] += x.x;
          cs[1] += x.y;
        } else {
          x.x += bvp[0]; x.y += bvp[1];
          x.x = x.x > 0.f ? x.x : 0.01f * x.x;
          x.y = x.y > 0.f ? x.y : 0.01f * x.y;
        }
        *(gf2)(&rg_uni(C + (long long)rowu * LD)[vo_ks]) = x;
      }
    } else {
      const int tm = p >> 2, tn = (p >> 1) & 1, h = p & 1;
      gf C = (gf)(second ? I.C2 : I.C);
#pragma unroll
      for (int rr = first; rr < first + count; ++rr) {
        const int r = 8 * h + rr;
        const int rowu = r0 + 32 * tm + (r & 3) + 8 * (r >> 2);
        float x = pv[tm][tn][r] + bvp[tn];
        x = x > 0.f ? x : 0.01f * x;
        rg_uni(C + (long long)rowu * LD)[vo_kc[tn]] = x;
      }
    }
  };
  auto store_part = [&](f32x16 (&pv)[2][2], const RgPtrs &I, bool second, int pblk, int p, const v2f (&rvs)[4]) __attribute__((always_inline)) {
    store_vals(pv, I, second, pblk, p, 0, KS ? 4 : 8, rvs);
  };
  // head-fusion sums of rows [r_first, r_first + r_count) of tile (tm, tn) of a finished accumulator set
  auto hf_rows = [&](f32x16 (&pv)[2][2], const RgPtrs &I, bool second, int pblk, int tile, int r_first, int r_count) __attribute__((always_inline)) {
    if constexpr (HFQ > 0) {
      const int tm = tile >> 1, tn = tile & 1;
      rg_hf_partial<HFQ>(pv[tm][tn], bvp[tn], wqp[tn], lane,
                         (gf)(second ? I.hf_out2 : I.hf_out) + ((long long)(wave * 2 + tn) * M + pblk * RG_BM + 32 * tm) * HFQ, r_first, r_count);
    }
  };
  auto load_ref_vals = [&](const RgPtrs &I, int blk, int p, int first, int count, v2f (&dst)[4]) __attribute__((always_inline)) {
    const int tm = p >> 2, q4 = p & 3;
#pragma unroll
    for (int i = first; i < first + count; ++i) {
      const int r = 4 * q4 + i;
      const int rowu = blk * RG_BM + 32 * tm + (r & 3) + 8 * (r >> 2);
      dst[i] = *(gcf2)(&rg_uni((gcf)I.ref + (long long)rowu * LD)[vo_ks]);
    }
  };
  auto load_ref = [&](const RgPtrs &I, int blk, int p, v2f (&dst)[4]) __attribute__((always_inline)) { load_ref_vals(I, blk, p, 0, 4, dst); };
  auto finish_colsum = [&](const RgPtrs &I, int pblk) __attribute__((always_inline)) {
    if constexpr (GRAD) {
      v2f t = {cs[0] + __shfl_xor(cs[0], 32), cs[1] + __shfl_xor(cs[1], 32)};
      *(gf2)(&((gf)I.colsum + (long long)pblk * RG_N)[(unsigned)(n0 + 2 * li)]) = t;   // both lane halves hold (and write) the same sums
      cs[0] = 0.f;
      cs[1] = 0.f;
    }
  };

  // ---- FUSE: one staged row slice (this lane's 4 columns of one row): h -> LeakyReLU'(h) * (dY . Wh)
  v4f fw[2];          // head weight rows q = 0, 1 over this lane's 4 columns, of the instance being staged
  v2f dzr[3][3];      // dY of the rows in flight (same ring as stg)
  float fcs[4] = {0.f, 0.f, 0.f, 0.f};   // column sums of the tile being staged (this wave's 16 rows)
  auto fuse_row = [&](v4f h, v2f dz) __attribute__((always_inline)) {
    v4f t;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float x = dz.x * fw[0][c];
      x = fmaf(dz.y, fw[1][c], x);        // k_head_dgrad's order and rounding (fma chain over q)
      x = h[c] > 0.f ? x : 0.01f * x;
      t[c] = x;
      fcs[c] += x;
    }
    return t;
  };
  // ---- next tile's rows: global -> registers, 3 of a wave's 16 rows per k-group (groups 0..5); registers -> the other
  // image two groups later (groups 2..7).  Nothing of it is left after the loop: a wait there would sit behind the
  // stores of the last slices (vmcnt completes in order).
  v4f stg[3][3];
  float stm[NMINOR > 0 ? NMINOR : 1][2];

#ifdef RG_STAMPS
  unsigned long long ph[5] = {0, 0, 0, 0, 0};
#endif
  // One tile.  `ac` receives this tile, `pv` holds the previous one.
  // DUAL: a tile leaves its K loop as the FIRST output's pre-activation; the last narrow segment (pi - a) is added to it
  // one tile later, half way through its stores - first output in k-groups 0..3, then 16 MFMAs on `pv` with the A
  // fragments saved from the old image before the prefetch overwrites it, second output in k-groups 4..7.
  auto block = [&](auto has_prev, f32x16 (&ac)[2][2], f32x16 (&pv)[2][2], int img, int cur, int nxt, int prv) __attribute__((always_inline)) {
    constexpr bool HP = decltype(has_prev)::value;
#ifdef RG_STAMPS
    const unsigned long long q0 = RG_T();
#endif
    // every wave is done with the other image (previous K loop) and this image has landed (the ds_writes of every wave
    // completed before its last lgkmcnt(0) of the previous K loop).  No fence: outstanding stores need not drain.
    asm volatile("s_barrier" ::: "memory");
#ifdef RG_STAMPS
    const unsigned long long q1 = RG_T();
#endif
    v4f pa[2];
    if constexpr (DUAL && HP) tail_frags(NMINOR - 1, img ^ 1, pa);
    // (readfirstlane: the divisions run on the vector ALU; scalar indices make the table reads scalar loads)
    const int cinst = rg_uni(cur / bpi), cblk = cur - cinst * bpi, ninst_ = rg_uni(nxt / bpi), nblk = nxt - ninst_ * bpi;
    const int pinst = rg_uni(prv / bpi), pblk = prv - pinst * bpi;
    const RgPtrs IC = rg_read(a, cinst), IN = rg_read(a, ninst_), IP = rg_read(a, pinst);
    w_pointers(IN, wpn, wkn);
    v4f t0[NMINOR > 0 ? NMINOR : 1], t1[NMINOR > 0 ? NMINOR : 1];
#pragma unroll
    for (int s = 0; s < NMINOR; ++s) load_tail(IC, s, t0[s], t1[s]);
    if constexpr (!GRAD) {
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
        const int col = KS ? n0 + 2 * li + tn : n0 + 32 * tn + li;
        bvc[tn] = ((gcf)IC.bias)[col];
        if constexpr (HFQ > 0) {
#pragma unroll
          for (int q = 0; q < HFQ; ++q) wqc[tn][q] = ((gcf)IC.hf_w)[(long long)q * a.hf_ldw + col];
        }
      }
    }
    const float *nsrc = (FUSE ? IN.fz_h : IN.A[0]) + (long long)nblk * RG_BM * LD;   // uniform; the lane adds its 16 bytes of the row
    if constexpr (FUSE) {
      // this tile was staged during the previous one: the column sums of its formed A block (4 partials per column)
      const float *red = fzred + img * (4 * RG_N);
      const float t = (red[tid] + red[RG_N + tid]) + (red[2 * RG_N + tid] + red[3 * RG_N + tid]);
      rg_uni((gf)IC.fz_colsum + (long long)cblk * RG_N)[(unsigned)tid] = t;
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int c = 0; c < 4; ++c) fw[q][c] = ((gcf)(IN.fz_w + (long long)q * a.fz_ldw))[(unsigned)(lane * 4 + c)];
#pragma unroll
      for (int c = 0; c < 4; ++c) fcs[c] = 0.f;
    }
    float *ndst = lds + (img ^ 1) * img_floats + lane * 4;
    // side work of k-group g, quarter j (runs under the 16 MFMAs of that step)
    auto piece = [&](int g, int j, int lbuf, int lg, const gcf4 (&lp)[2], gcf lk) __attribute__((always_inline)) {
      load_b_part(lbuf, lg, j, lp, lk);                 // B fragments of the next k-group
      // next tile's rows: row 3 g + j of the wave's 16 (groups 0..5), written to the other image two groups later
      if (j < 3 && g < 6 && 3 * g + j < RG_BM / 4) {
        const int row = wave + 4 * (3 * g + j);
        stg[g % 3][j] = rg_uni((gcf4)(nsrc + row * LD))[(unsigned)lane];
        if constexpr (FUSE) dzr[g % 3][j] = *(gcf2)rg_uni(IN.A[1] + ((long long)nblk * RG_BM + row) * a.lda[1]);
      }
      if (j < 3 && g >= 2 && 3 * (g - 2) + j < RG_BM / 4) {
        const int row = wave + 4 * (3 * (g - 2) + j);
        v4f val = stg[(g - 2) % 3][j];
        if constexpr (FUSE) {
          val = fuse_row(val, dzr[(g - 2) % 3][j]);
          rg_uni((__attribute__((address_space(1))) v4f *)(IN.fz_out + ((long long)nblk * RG_BM + row) * LD))[(unsigned)lane] = val;
        }
        *reinterpret_cast<v4f *>(ndst + row * pitch) = val;
      }
      if constexpr (FUSE) {
        if (g == 7 && j == 2) {   // all 16 rows of this wave are in: its column-sum partials for the next tile's start
          float *red = fzred + (img ^ 1) * (4 * RG_N) + wave * RG_N + lane * 4;
          *reinterpret_cast<v4f *>(red) = v4f{fcs[0], fcs[1], fcs[2], fcs[3]};
        }
      }
      if (j == 3 && g == 0) {
#pragma unroll
        for (int s = 0; s < NMINOR; ++s)
#pragma unroll
          for (int u = 0; u < 2; ++u)
            stm[s][u] = ((gcf)(IN.A[1 + s] + (long long)nblk * RG_BM * a.lda[1 + s]))[m_src[s][u]];
      }
      if (j == 3 && g == 2) {
#pragma unroll
        for (int s = 0; s < NMINOR; ++s)
#pragma unroll
          for (int u = 0; u < 2; ++u)
            if (m_ok[s][u]) lds[(img ^ 1) * img_floats + m_dst[s][u]] = stm[s][u];
      }
      if constexpr (HP && DUAL) {                       // two slices per group: first output in groups 0..3, second in 4..7
        store_vals(pv, IP, g >= 4, pblk, 2 * (g & 3) + (j >> 1), 4 * (j & 1), 4, rv[0]);
        hf_rows(pv, IP, g >= 4, pblk, g & 3, 4 * j, 4);   // tile g & 3: 4 of its 16 rows per step
      }
      if constexpr (HP && !DUAL) {
        store_vals(pv, IP, false, pblk, g, KS ? j : 2 * j, KS ? 1 : 2, rv[g & 1]);
        hf_rows(pv, IP, false, pblk, g >> 1, 8 * (g & 1) + 2 * j, 2);   // tile g / 2: 2 of its 16 rows per step
        if constexpr (GRAD) {
          if (g < 7) load_ref_vals(IP, pblk, g + 1, j, 1, rv[(g + 1) & 1]);
          else if (j == 3) finish_colsum(IP, pblk);
        }
      }
      if constexpr (GRAD) {
        if (g == 7) load_ref_vals(IC, cblk, 0, j, 1, rv[0]);   // first slice of THIS tile, consumed in the next tile's group 0
      }
    };
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) ac[i][j][r] = 0.f;
#ifdef RG_STAMPS
    const unsigned long long q2 = RG_T();
#endif
#pragma unroll
    for (int g = 0; g < G; g += 2) {
      if constexpr (HP && DUAL) {
        if (g == 4) tail_mfma(pv, pa, tp0, tp1);   // the previous tile becomes its second output
      }
      group(ac, 0, g, img, [&](int j) __attribute__((always_inline)) { piece(g, j, 1, g + 1, wp, wk); });
      if (g + 2 < G) group(ac, 1, g + 1, img, [&](int j) __attribute__((always_inline)) { piece(g + 1, j, 0, g + 2, wp, wk); });
      else group(ac, 1, g + 1, img, [&](int j) __attribute__((always_inline)) { piece(g + 1, j, 0, 0, wpn, wkn); });   // group 0 of the next tile
    }
#ifdef RG_STAMPS
    const unsigned long long q3 = RG_T();
#endif
#pragma unroll
    for (int s = 0; s < (DUAL ? NMINOR - 1 : NMINOR); ++s) tail(ac, s, img, t0[s], t1[s]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's writes to the other image have landed (no-tail forms)
    if constexpr (DUAL) { tp0 = t0[NMINOR - 1]; tp1 = t1[NMINOR - 1]; }
    // roll the per-tile state on
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      wp[tn] = wpn[tn];
      bvp[tn] = bvc[tn];
      if constexpr (HFQ > 0) {
#pragma unroll
        for (int q = 0; q < HFQ; ++q) wqp[tn][q] = wqc[tn][q];
      }
    }
    wk = wkn;
#ifdef RG_STAMPS
    const unsigned long long q4 = RG_T();
    ph[0] += q1 - q0; ph[1] += q2 - q1; ph[2] += q3 - q2; ph[3] += q4 - q3; ph[4] += 1;
#endif
  };
  using T = std::true_type;
  using F = std::false_type;

  // Tile order: round-robin over the workgroups (tile t, t + grid, ...): at any moment the chip streams ONE contiguous
  // region of the activations.  Giving each workgroup its own contiguous run of tiles instead (to stay on one instance's
  // weights) was measured slower, 92 against 107 TFLOP/s on a single instance: 256 separate streams through HBM.
  const int ntile_end = ntiles;
  int cur = blockIdx.x;
  if (cur >= ntile_end) return;
  const unsigned long long life_c0 = __builtin_amdgcn_s_memtime(), life_w0 = wall_clock64();
  const int stride = gridDim.x;

  {   // first tile's image, not overlapped (once per workgroup)
    const int cinst = rg_uni(cur / bpi), cblk = cur - cinst * bpi;
    const RgPtrs IC = rg_read(a, cinst);
    w_pointers(IC, wp, wk);
    load_b(0, 0, wp, wk);
    const float *src = (FUSE ? IC.fz_h : IC.A[0]) + (long long)cblk * RG_BM * LD;
    if constexpr (FUSE) {
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int c = 0; c < 4; ++c) fw[q][c] = ((gcf)(IC.fz_w + (long long)q * a.fz_ldw))[(unsigned)(lane * 4 + c)];
    }
#pragma unroll
    for (int i = 0; i < RG_BM / 4; ++i) {
      const int r = wave + 4 * i;
      v4f val = ((gcf4)(src + r * LD))[(unsigned)lane];
      if constexpr (FUSE) {
        val = fuse_row(val, *(gcf2)(IC.A[1] + ((long long)cblk * RG_BM + r) * a.lda[1]));
        ((__attribute__((address_space(1))) v4f *)(IC.fz_out + ((long long)cblk * RG_BM + r) * LD))[(unsigned)lane] = val;
      }
      *reinterpret_cast<v4f *>(lds + r * pitch + lane * 4) = val;
    }
    if constexpr (FUSE) *reinterpret_cast<v4f *>(fzred + wave * RG_N + lane * 4) = v4f{fcs[0], fcs[1], fcs[2], fcs[3]};
#pragma unroll
    for (int s = 0; s < NMINOR; ++s)
#pragma unroll
      for (int u = 0; u < 2; ++u)
        if (m_ok[s][u]) lds[m_dst[s][u]] = ((gcf)(IC.A[1 + s] + (long long)cblk * RG_BM * a.lda[1 + s]))[m_src[s][u]];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  // a workgroup's last tile prefetches itself again (nobody consumes it): keeps the loop free of branches
  int nxt = cur + stride < ntile_end ? cur + stride : cur;
  block(F(), acc[0], acc[1], 0, cur, nxt, cur);
  int prv = cur, set = 1;
  cur += stride;
#pragma unroll 1
  while (cur < ntile_end) {
    nxt = cur + stride < ntile_end ? cur + stride : cur;
    block(T(), acc[1], acc[0], 1, cur, nxt, prv);
    prv = cur;
    cur += stride;
    set = 0;
    if (cur >= ntile_end) break;
    nxt = cur + stride < ntile_end ? cur + stride : cur;
    block(T(), acc[0], acc[1], 0, cur, nxt, prv);
    prv = cur;
    cur += stride;
    set = 1;
  }
  // the last tile's result: all slices now
  {
    const int pinst = rg_uni(prv / bpi), pblk = prv - pinst * bpi;
    const RgPtrs IP = rg_read(a, pinst);
    auto flush = [&](f32x16 (&pv)[2][2], int img) __attribute__((always_inline)) {
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        if constexpr (GRAD) {
          if (p > 0) load_ref(IP, pblk, p, rv[p & 1]);   // slice 0 was requested in the tile's last k-group
        }
        store_part(pv, IP, false, pblk, p, rv[p & 1]);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) hf_rows(pv, IP, false, pblk, t, 0, 16);
      if constexpr (DUAL) {   // the tile's image is still in place (its workgroup prefetched into the other one)
        v4f pa[2];
        tail_frags(NMINOR - 1, img, pa);
        tail_mfma(pv, pa, tp0, tp1);
#pragma unroll
        for (int p = 0; p < 8; ++p) store_part(pv, IP, true, pblk, p, rv[0]);
#pragma unroll
        for (int t = 0; t < 4; ++t) hf_rows(pv, IP, true, pblk, t, 0, 16);
      }
      finish_colsum(IP, pblk);
    };
    if (set == 1) flush(acc[0], 0);
    else flush(acc[1], 1);
  }
#ifdef RG_STAMPS
  if (tid == 0 && blockIdx.x < 1024)
    for (int i = 0; i < 5; ++i) g_rg_phase[8 * blockIdx.x + i] = ph[i];
#endif
  if (tid == 0 && blockIdx.x < 1024) {
    g_rg_life[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - life_c0;
    g_rg_life[2 * blockIdx.x + 1] = wall_clock64() - life_w0;
  }
}

}  // namespace

bool rowgemm_from_problems(const GemmProblem *probs, int nprob, RowGemmArgs &args) {
  if (nprob < 1 || nprob > RG_MAX_INST) return false;
  memset(&args, 0, sizeof(args));
  const GemmProblem &p0 = probs[0];
  // the 256-wide segment may sit anywhere in the list; the narrow ones keep their order (the dual's extra one is last)
  auto main_of = [](const GemmProblem &p) {
    int m = -1;
    for (int s = 0; s < p.nseg; ++s)
      if (p.seg[s].K == RG_KMAIN) { if (m >= 0) return -1; m = s; }
    return m;
  };
  const int main0 = main_of(p0);
  if (main0 >= 0 && (p0.seg[main0].lda != LD || p0.ldc != LD || (p0.C2 && p0.ldc2 != LD) || (p0.ref && p0.ldref != LD) ||
                     (!p0.seg[main0].b_kc && p0.seg[main0].ldb != LD)))
    return false;
  if (main0 < 0 || p0.N != RG_N || p0.M % RG_BM || p0.M < RG_BM || p0.ksplit != 1) return false;
  const int nminor = p0.nseg - 1;
  if (nminor > RG_MAX_MINOR) return false;
  const bool dual = p0.emit_seg >= 0 && p0.emit_seg < p0.nseg - 1;
  if (dual && (p0.emit_seg != p0.nseg - 2 || main0 == p0.nseg - 1)) return false;
  const int ks = p0.seg[main0].b_kc ? 0 : 1;
  const bool grad = p0.epi == EPI_LRELU_GRAD;
  if (!grad && p0.epi != EPI_LRELU) return false;
  if (grad && (p0.bias || !ks || dual || p0.hf_w || !p0.ref || !p0.colsum)) return false;
  if (!grad && (ks || p0.colsum || !p0.bias)) return false;
  if (p0.hf_w && p0.hf_q != 2) return false;
  args.M = p0.M; args.ninst = nprob; args.blocks_per_inst = p0.M / RG_BM;
  args.nminor = nminor; args.ks = ks; args.grad = grad; args.dual = dual;
  args.hf_q = p0.hf_w ? p0.hf_q : 0; args.hf_ldw = p0.hf_ldw;
  const bool fz = p0.fz_h != nullptr;
  if (fz && (!grad || nminor != 1 || !p0.fz_w || !p0.fz_out || !p0.fz_colsum)) return false;
  args.fz = fz; args.fz_ldw = p0.fz_ldw;
  args.ldc = p0.ldc; args.ldc2 = p0.ldc2; args.ldref = p0.ldref;
  for (int i = 0; i < nprob; ++i) {
    const GemmProblem &p = probs[i];
    if (p.M != p0.M || p.N != p0.N || p.nseg != p0.nseg || p.ksplit != 1 || p.epi != p0.epi || p.emit_seg != p0.emit_seg ||
        main_of(p) != main0 || (p.hf_w != nullptr) != (p0.hf_w != nullptr) || p.hf_q != p0.hf_q || p.hf_ldw != p0.hf_ldw ||
        p.ldc != p0.ldc || p.ldc2 != p0.ldc2 || p.ldref != p0.ldref || (p.bias != nullptr) != (p0.bias != nullptr) ||
        (p.ref != nullptr) != (p0.ref != nullptr) || (p.colsum != nullptr) != (p0.colsum != nullptr))
      return false;
    if (dual && (!p.C2 || (p.hf_w && !p.hf_out2))) return false;
    if ((p.fz_h != nullptr) != fz || p.fz_ldw != p0.fz_ldw) return false;
    if (fz && (!p.fz_w || !p.fz_out || !p.fz_colsum || (reinterpret_cast<uintptr_t>(p.fz_h) & 15) || (reinterpret_cast<uintptr_t>(p.fz_out) & 15)))
      return false;
    RowGemmInst I;
    memset(&I, 0, sizeof(I));
    int m = 0;
    for (int s = 0; s < p.nseg; ++s) {
      const GemmSeg &sg = p.seg[s], &s0 = p0.seg[s];
      if (!sg.a_kc || sg.b_kc != (ks ? 0 : 1) || sg.K != s0.K || sg.lda != s0.lda || sg.ldb != s0.ldb) return false;
      const int slot = s == main0 ? 0 : 1 + m++;
      if (slot > 0 && (sg.K < 1 || sg.K > 8)) return false;
      if (fz && slot > 0 && (sg.K != 2 || sg.lda % 2 || (reinterpret_cast<uintptr_t>(sg.A) & 7))) return false;   // dY rows read as float2
      if (slot == 0 && (sg.lda % 4 || (reinterpret_cast<uintptr_t>(sg.A) & 15))) return false;
      // (K-contiguous weights: any row pitch - gfx950 serves dwordx4 loads from 4-byte aligned addresses, and the critics'
      // layer-0 weights have rows of 256 + act_dim floats)
      if (ks && (sg.ldb % 2 || (reinterpret_cast<uintptr_t>(sg.B) & 7))) return false;
      I.A[slot] = sg.A; I.W[slot] = sg.B;
      if (i == 0) {
        args.lda[slot] = sg.lda; args.ldw[slot] = sg.ldb;
        if (slot > 0) args.kminor[slot - 1] = sg.K;
      }
    }
    if (ks && (p.ldc % 2 || (reinterpret_cast<uintptr_t>(p.C) & 7) || (grad && (p.ldref % 2 || (reinterpret_cast<uintptr_t>(p.ref) & 7)))))
      return false;
    I.bias = p.bias; I.C = p.C; I.C2 = p.C2; I.ref = p.ref; I.colsum = p.colsum;
    I.hf_w = p.hf_w; I.hf_out = p.hf_out; I.hf_out2 = p.hf_out2;
    I.fz_h = p.fz_h; I.fz_w = p.fz_w; I.fz_out = p.fz_out; I.fz_colsum = p.fz_colsum;
    args.inst[i] = I;
  }
  // instantiated forms; FDQL_ROWGEMM_FORMS (bit mask, tuning hook): 1 forward, 2 dgrad, 4 dual forward
  const char *fe = getenv("FDQL_ROWGEMM_FORMS");   // (read per plan build / test call)
  const int forms = fe ? atoi(fe) : 2;   // the forward form with head fusion is slower than the tile kernel in the update so far
  if (grad) return (forms & 2) && nminor == 1;
  if (dual) return (forms & 4) && nminor == 2 && args.hf_q == 2;   // (spills registers so far: off by default)
  if (!(forms & 1)) return false;
  return nminor <= 1 && (args.hf_q == 2 || args.hf_q == 0);
}

int rowgemm_read_life(unsigned long long *out, int cap) {
  static unsigned long long h[2 * 1024];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_rg_life), sizeof(h)) != hipSuccess) return -1;
  const int n = cap < 2 * 1024 ? cap : 2 * 1024;
  for (int i = 0; i < n; ++i) out[i] = h[i];
#ifdef RG_STAMPS
  static unsigned long long q[8 * 1024];
  if (hipMemcpyFromSymbol(q, HIP_SYMBOL(g_rg_phase), sizeof(q)) == hipSuccess) {
    double t[5] = {0, 0, 0, 0, 0};
    for (int w = 0; w < 1024; ++w) for (int i = 0; i < 5; ++i) t[i] += (double)q[8 * w + i];
    if (t[4] > 0)
      fprintf(stderr, "rowgemm phases, cycles per tile: barrier wait %.0f, set-up %.0f, K loop %.0f, tails + roll %.0f (%.0f tiles)\n",
              t[0] / t[4], t[1] / t[4], t[2] / t[4], t[3] / t[4], t[4]);
  }
#endif
  return n;
}

double rowgemm_flops(const RowGemmArgs &a) {
  double k = RG_KMAIN;
  for (int s = 0; s < a.nminor; ++s) k += a.kminor[s];
  double f = 2.0 * a.M * (double)RG_N * k * a.ninst;
  return f;
}

template <bool KS, int NMINOR, bool DUAL, int HFQ, bool GRAD, bool FUSE = false>
static hipError_t rg_launch(const RowGemmArgs &a, int grid, int lds_bytes, hipStream_t s) {
  // the opt-in to > 64 KiB of dynamic LDS belongs to the (device, function) pair
  static bool attr[64];
  static std::mutex mu;
  auto kern = &k_rowgemm<KS, NMINOR, DUAL, HFQ, GRAD, FUSE>;
  {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lk(mu);
    if (!attr[dev]) {
      e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
      if (e != hipSuccess) return e;
      attr[dev] = true;
    }
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds_bytes, s, a);
  return hipGetLastError();
}

hipError_t rowgemm_launch(const RowGemmArgs &a, hipStream_t s) {
  static int ncu_of[64];
  static std::mutex ncu_mu;
  int ncu = 0;
  {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorUnknown;
    std::lock_guard<std::mutex> lk(ncu_mu);
    if (!ncu_of[dev]) {
      hipDeviceProp_t p;
      if (hipGetDeviceProperties(&p, dev) != hipSuccess) return hipErrorUnknown;
      ncu_of[dev] = p.multiProcessorCount;
    }
    ncu = cu_budget(ncu_of[dev]);
  }
  const int ntiles = a.ninst * a.blocks_per_inst;
  const int grid = ntiles < ncu ? ntiles : ncu;
  const int lds_bytes = 2 * RG_BM * (RG_KMAIN + 8 * a.nminor + 4) * 4 + (a.fz ? 2 * 4 * RG_N * 4 : 0);
  if (a.grad && a.fz) return rg_launch<true, 1, false, 0, true, true>(a, grid, lds_bytes, s);
  if (a.grad) return rg_launch<true, 1, false, 0, true>(a, grid, lds_bytes, s);
  if (a.dual) return rg_launch<false, 2, true, 2, false>(a, grid, lds_bytes, s);
  if (a.nminor == 0) return a.hf_q ? rg_launch<false, 0, false, 2, false>(a, grid, lds_bytes, s) : rg_launch<false, 0, false, 0, false>(a, grid, lds_bytes, s);
  return a.hf_q ? rg_launch<false, 1, false, 2, false>(a, grid, lds_bytes, s) : rg_launch<false, 1, false, 0, false>(a, grid, lds_bytes, s);
}

}  // namespace fdql
