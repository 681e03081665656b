// Row-block chain kernel: several Linear layers per launch with the activations resident in LDS.
//
// Why (measured on MI355X, tools/proto/rowblock.hip, profiles/r02_*): the grouped GEMM of gemm.hip runs one layer per
// launch and sends every activation through HBM/L2 between layers; at the update's sizes (12.5 k rows, K = N = 256) a
// third of each launch is tile start-up, operand staging and the epilogue's round trip.  Here a 256-thread workgroup
// (one wave per SIMD, one workgroup per CU) owns 64 rows of the batch and interprets a short program of operations
// (chain.h) on them:
//   CH_LOAD    global rows (the K-segments of a torch.cat) -> an LDS image [64][pitch]
//   CH_GEMM    acc[64 x N<=256] (+)= sum over K-segments of image x W^T; wave w owns columns [64w, 64w+64) as 2x2 MFMA
//              tiles (v_mfma_f32_32x32x2_f32), computed TRANSPOSED (the weights are the MFMA's A operand, the activations
//              its B operand): a lane then holds, for ITS batch row, four consecutive output columns per register quad,
//              so the epilogue (bias, LeakyReLU) writes the next layer's LDS image with ds_write_b128 and global memory
//              with global_store_dwordx4 straight from the accumulators (round 3; before: 64 ds_write_b32 per lane, a
//              barrier, and a second pass image -> memory with its index arithmetic: 10 k of a layer's 50 k cycles)
//   CH_NARROW  skip heads and other N <= 32 outputs: the K range is dealt round-robin to the four waves, partial
//              32-column tiles are summed through LDS in wave order
// The A operand is read from LDS by ds_read_b128 ((pitch / 4) odd: the 32 rows of a fragment hit distinct banks); the B
// operand (weights) goes global -> registers directly as MFMA fragments: lane (li, lh) of a column tile holds, for a
// 32-k group, the 16 k's 32g + 16 lh + 0..15 of ITS column - K-contiguous weights: 64 contiguous bytes per lane
// (4 x dwordx4).  Activations and weights pair the same k's in every MFMA
// step, so the fp32 sum runs over a fixed permutation of k (bitwise a k-ordered fma chain, like gemm.hip).  No LDS
// staging of B, no barrier inside a K loop; the next group's fragments are requested before the current group's 64 MFMAs.
#include "chain.h"

#include <mutex>

#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace fdql {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(1))) float *gcf;
typedef __attribute__((address_space(1))) float *gf;
typedef const __attribute__((address_space(1))) v4f *gcf4;
typedef const __attribute__((address_space(1))) v2f *gcf2;

static_assert(CH_LDS_FLOATS * 4 + CH_MAX_OPS * sizeof(ChainOp) + 256 <= 163840, "LDS budget of the chain kernel");

__device__ __forceinline__ unsigned ch_lds_addr(const float *p) {
  return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float *)p;
}
// early-clobber destination: a fragment can never be allocated on top of its own address register (gemm.hip, finding 2)
__device__ __forceinline__ void ch_rd128(v4f &d, unsigned addr) { asm volatile("ds_read_b128 %0, %1" : "=&v"(d) : "v"(addr)); }
template <int N>
__device__ __forceinline__ void ch_lgkm_wait() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ int ch_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ const float *ch_uni(const float *p) {
  const uintptr_t u = reinterpret_cast<uintptr_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
  return reinterpret_cast<const float *>(((uintptr_t)hi << 32) | lo);
}

// Diagnostic stamps (fdql_debug_chain_stamps(NULL, 1) -> chain_read_stamps): the workgroup in the middle of a launch records
// s_memtime at its entry, after the program fetch and after every operation.
__device__ unsigned long long g_ch_stamps[8 * CH_MAX_OPS + 4];
__device__ int g_ch_stamps_on = 0;

// One workgroup per CU (512 registers per lane: the accumulators live in AccVGPRs, two weight-fragment buffers in VGPRs).
// TM = row tiles of 32 per workgroup: 2 (64 rows: the weights a workgroup streams are shared by two tiles) or 1 (32 rows:
// twice the workgroups - small batches, e.g. one rank's share of a data-parallel step - and half the LDS per image, which
// is what lets a 376-column observation image and a hidden image live together).
template <int TM>
__global__ __launch_bounds__(CH_THREADS, 1) void k_chain(const ChainProblem *__restrict__ probs, int nprob,
                                                         const ChainOp *__restrict__ ops_all) {
  constexpr int BM = 32 * TM;
  __shared__ ChainOp s_ops[CH_MAX_OPS];
  extern __shared__ __attribute__((aligned(16))) float lds[];

  const int tid = threadIdx.x, lane = tid & 63, wave = ch_uni(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int bid = blockIdx.x;
  // which program, which rows: ONE load round - lane i reads problem i's header, ballot, the owning lane's fields are
  // broadcast (every dependent global access before the first MFMA is a memory round trip this workgroup sits out)
  const bool stamp = g_ch_stamps_on && bid == (int)(gridDim.x / 2) && tid == 0;
  int nstamp = 0;
#define CH_STAMP() do { if (stamp) g_ch_stamps[nstamp++] = __builtin_amdgcn_s_memtime(); } while (0)
  if (stamp) g_ch_stamps[nstamp++] = __builtin_amdgcn_s_memtime();
  int rows = 0, blk = 0, op_start = 0, nops = 0;
  for (int base = 0; base < nprob; base += 64) {
    const int i = base + lane;
    int bs = 0x7fffffff, rw = 0, os = 0, no = 0;
    if (i < nprob) {
      const __attribute__((address_space(1))) ChainProblem *pp = (const __attribute__((address_space(1))) ChainProblem *)probs + i;
      bs = pp->block_start; rw = pp->rows; os = pp->op_start; no = pp->nops;
    }
    const unsigned long long m = __ballot(bid >= bs);
    if (m) {
      const int src = 63 - __builtin_clzll(m);
      rows = __builtin_amdgcn_readlane(rw, src);
      op_start = __builtin_amdgcn_readlane(os, src);
      nops = __builtin_amdgcn_readlane(no, src);
      blk = bid - __builtin_amdgcn_readlane(bs, src);
    }
  }
  const int r0 = blk * BM;
  {   // the program -> LDS in one coalesced round (fields are then read from LDS, not through L2)
    const int *src = reinterpret_cast<const int *>(ops_all + op_start);
    int *dst = reinterpret_cast<int *>(s_ops);
    constexpr int WORDS = (int)(sizeof(ChainOp) / 4);
    const int total = min(nops, CH_MAX_OPS) * WORDS;
    for (int e = tid; e < total; e += CH_THREADS) dst[e] = ((const __attribute__((address_space(1))) int *)src)[e];
    __syncthreads();
  }
  const unsigned lds0 = ch_lds_addr(lds);
  // The weight slices of every CH_NARROW operation -> their staging areas, once, before anything else is in flight:
  // fetched inside the operation they would cost a memory round trip (behind the layer's own stores) per operation.
  // Slice of a segment: [N][wp = roundup(K, 16) + 4] floats, k >= K zero.  Whole 16-byte quads when rows allow it (all of
  // a slice's requests in flight at once, then the LDS writes), single floats otherwise.
  {
    bool any = false;
    for (int ip = 0; ip < nops; ++ip) {
      const ChainOp &op = s_ops[ip];
      if (ch_uni(op.kind) != CH_NARROW) continue;
      any = true;
      const int N = ch_uni(op.N), nseg = ch_uni(op.nseg);
      int wslot = ch_uni(op.slot);
      for (int s = 0; s < nseg; ++s) {
        gcf W = (gcf)ch_uni(op.seg[s].W);
        const int ldw = ch_uni(op.seg[s].ldw), K = ch_uni(op.seg[s].K);
        const int k16 = (K + 15) & ~15, wp = k16 + 4;   // (wp / 4) odd: conflict-free B fragments
        if (((K | ldw) & 3) == 0 && (reinterpret_cast<uintptr_t>(op.seg[s].W) & 15) == 0) {
          const int kq = k16 >> 2, total = N * kq;   // quads
          for (int base = 0; base < total; base += 8 * CH_THREADS) {
            v4f v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const int e = min(base + tid + u * CH_THREADS, total - 1);
              const int n = e / kq, k = (e - n * kq) * 4;
              v[u] = *(gcf4)(W + (long long)n * ldw + min(k, K - 4));
              if (k >= K) v[u] = v4f{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const int e = base + tid + u * CH_THREADS;
              const int n = e / kq, k = (e - n * kq) * 4;
              if (e < total) *reinterpret_cast<v4f *>(&lds[wslot + n * wp + k]) = v[u];
            }
          }
        } else {
          const int total = N * k16;
          for (int base = 0; base < total; base += 8 * CH_THREADS) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const int e = base + tid + u * CH_THREADS;
              const int n = e / k16, k = e - n * k16;
              v[u] = (e < total && k < K) ? W[(long long)n * ldw + k] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const int e = base + tid + u * CH_THREADS;
              const int n = e / k16, k = e - n * k16;
              if (e < total) lds[wslot + n * wp + k] = v[u];
            }
          }
        }
        wslot += N * wp;
      }
    }
    if (any) __syncthreads();
  }
  if (stamp) g_ch_stamps[nstamp++] = __builtin_amdgcn_s_memtime();

  f32x16 acc[TM][2];   // [tm][tn]: rows 32 tm + ..., columns of column tile tn
  v4f hacc[2];        // narrow-head accumulators: this wave's 16 rows x 16 columns per tile (v_mfma_f32_16x16x4_f32)
  v4f hodd[2];        // CH_NARROW: the odd 16-k steps' sums (two independent MFMA chains; added at CHF_FINISH)
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[a][0][r] = 0.f; acc[a][1][r] = 0.f; }
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    hacc[a] = v4f{0.f, 0.f, 0.f, 0.f};
    hodd[a] = v4f{0.f, 0.f, 0.f, 0.f};
  }

  // ---------------------------------------------------------------- 32-k group of MFMAs: A from LDS, B in registers
  // a0 / a1: byte addresses of this lane's row in row tile 0 / 1 at the group's first k (+ 16 lh floats)
  // af: the group's first fragments (j = 0), already requested by the previous group (or the segment prologue); the
  // group requests the NEXT group's first fragments (na0 / na1) under its last MFMA step, so no LDS latency is exposed
  // at a group boundary.  RIDER: the rider's two A fragments are requested first and consumed after the 64 main MFMAs.
  // The NEXT group's weight fragments (pa / pb / ph0 / ph1 -> bn / rbn) are requested two or three at a time under each
  // MFMA step, not in one burst at the group boundary: a wave issues in order, and a burst of 12 cache-line-scattered
  // loads holds the instruction stream (and with it the matrix pipe) for ~1 k cycles while the texture path takes them in.
  auto mfma_group_kc = [&](unsigned a0, unsigned a1, unsigned na0, unsigned na1, v4f (&af)[2], const v4f (&b)[2][4], unsigned a16g,
                           const v4f (&rb)[2][2], gcf pa, gcf pb, gcf ph0, gcf ph1, v4f (&bn)[2][4], v4f (&rbn)[2][2],
                           auto rider_tag) __attribute__((always_inline)) {
    constexpr int NTH = decltype(rider_tag)::value;   // head tiles riding: 0, 1 or 2
    constexpr bool RIDER = NTH > 0;
    v4f a[2][2];   // [parity of j][tm]
    v4f ra[2], an[2];
    if constexpr (RIDER) {
      ch_rd128(ra[0], a16g);
      ch_rd128(ra[1], a16g + 64u);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j < 3) {
        ch_rd128(a[(j + 1) & 1][0], a0 + 16 * (j + 1));
        if constexpr (TM > 1) ch_rd128(a[(j + 1) & 1][1], a1 + 16 * (j + 1));
        ch_lgkm_wait<TM>();
      } else {
        ch_lgkm_wait<0>();
        ch_rd128(an[0], na0);
        if constexpr (TM > 1) ch_rd128(an[1], na1);
      }
      if (j == 0) {
        asm volatile("" : "+v"(af[0]));
        if constexpr (TM > 1) asm volatile("" : "+v"(af[1]));
        if constexpr (RIDER) asm volatile("" : "+v"(ra[0]), "+v"(ra[1]));
      } else {
        asm volatile("" : "+v"(a[j & 1][0]));
        if constexpr (TM > 1) asm volatile("" : "+v"(a[j & 1][1]));
      }
      bn[0][j] = *(gcf4)(pa + 4 * j);
      bn[1][j] = *(gcf4)(pb + 4 * j);
      if constexpr (RIDER) {
        if (j < 2) rbn[j][0] = *(gcf4)(ph0 + 16 * j);
        if constexpr (NTH > 1) { if (j >= 2) rbn[j - 2][1] = *(gcf4)(ph1 + 16 * (j - 2)); }
      }
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < 2; ++tn)
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[tn][j][c], j == 0 ? af[tm][c] : a[j & 1][tm][c], acc[tm][tn], 0, 0, 0);
    }
    if constexpr (RIDER) {
#pragma unroll
      for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          hacc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[hh][c], rb[hh][0][c], hacc[0], 0, 0, 0);
          if constexpr (NTH > 1) hacc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[hh][c], rb[hh][1][c], hacc[1], 0, 0, 0);
        }
    }
    af[0] = an[0];
    if constexpr (TM > 1) af[1] = an[1];
    asm volatile("" ::"v"(a0), "v"(a1), "v"(na0), "v"(na1), "v"(a16g));
  };
  // 8-k tail group: lanes lh = 0 / 1 hold k = kb + 0..3 / kb + 4..7 (component c -> MFMA step c)
  auto mfma_tail = [&](unsigned a0, unsigned a1, const v4f &b0, const v4f &b1) __attribute__((always_inline)) {
    v4f a[2];
    ch_rd128(a[0], a0);
    if constexpr (TM > 1) ch_rd128(a[1], a1);
    ch_lgkm_wait<0>();
    asm volatile("" : "+v"(a[0]));
    if constexpr (TM > 1) asm volatile("" : "+v"(a[1]));
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        acc[tm][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[c], a[tm][c], acc[tm][0], 0, 0, 0);
        acc[tm][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b1[c], a[tm][c], acc[tm][1], 0, 0, 0);
      }
    asm volatile("" ::"v"(a0), "v"(a1));
  };

  // ---------------------------------------------------------------- one K-segment of a GEMM, K-contiguous weights
  // RIDER (hw != null): the segment's block of a narrow skip head rides in the same K loop - per 32-k group each wave
  // also multiplies ITS 16 rows of the image by the head's columns (8 v_mfma_f32_16x16x4_f32 per 16 head columns beside
  // the 64 v_mfma_f32_32x32x2_f32), operands requested a group ahead like the main ones: the head costs ~6 % of the
  // layer instead of a latency-bound pass of its own.
  auto gemm_seg_kc = [&](const ChainSeg &S, int N, int n0, const float *hw, int hldw, int hN, auto rider_tag) __attribute__((always_inline)) {
    constexpr int NTH = decltype(rider_tag)::value;
    constexpr bool RIDER = NTH > 0;
    const float *W = ch_uni(S.W);
    const int ldw = ch_uni(S.ldw), K = ch_uni(S.K), pitch = ch_uni(S.pitch), slot = ch_uni(S.slot);
    const int na = min(n0 + li, N - 1), nb = min(n0 + 32 + li, N - 1);   // clamped: rows beyond N repeat row N-1, never stored
    gcf wpa = (gcf)(W + (long long)na * ldw + 16 * lh), wpb = (gcf)(W + (long long)nb * ldw + 16 * lh);
    const unsigned a0 = lds0 + (unsigned)(slot + li * pitch + 16 * lh) * 4u, a1 = a0 + (unsigned)(32 * pitch) * 4u;
    const int G = K >> 5, ntail = ((K & 31) + 7) >> 3;
    // rider operands: lane (lj, kq) of head tile t reads head row min(16 t + lj, hN - 1), k's 16 h + 4 kq + 0..3
    const int lj = lane & 15, kq = lane >> 4;
    constexpr bool two = NTH > 1;
    gcf hp0 = nullptr, hp1 = nullptr;
    unsigned a16 = 0;
    if constexpr (RIDER) {
      hp0 = (gcf)(hw + (long long)min(lj, hN - 1) * hldw + 4 * kq);
      hp1 = (gcf)(hw + (long long)min(16 + lj, hN - 1) * hldw + 4 * kq);
      a16 = lds0 + (unsigned)(slot + (16 * (wave & (2 * TM - 1)) + lj) * pitch + 4 * kq) * 4u;   // (32-row blocks: waves 2, 3 repeat 0, 1; never stored)
    }
    // the tail groups' fragments (at most four 8-k groups) are requested before the full groups: their latency hides
    // behind them, one request round instead of a memory round trip per tail group (with a rider: only the first one
    // up front - the rider's operands need the registers)
    constexpr int NPRE = RIDER ? 1 : 4;
    v4f tb0[NPRE], tb1[NPRE];
    auto load_tail = [&](int t, v4f &d0, v4f &d1) __attribute__((always_inline)) {
      const int kb = 32 * G + 8 * t + 4 * lh;     // this lane's 4 k's
      gcf pa = (gcf)(W + (long long)na * ldw + kb), pb = (gcf)(W + (long long)nb * ldw + kb);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        d0[c] = kb + c < K ? pa[c] : 0.f;
        d1[c] = kb + c < K ? pb[c] : 0.f;
      }
    };
#pragma unroll
    for (int t = 0; t < NPRE; ++t) {
      tb0[t] = v4f{0.f, 0.f, 0.f, 0.f};
      tb1[t] = v4f{0.f, 0.f, 0.f, 0.f};
      if (t < ntail) load_tail(t, tb0[t], tb1[t]);
    }
    if (G > 0) {
      v4f b[2][2][4];   // [buffer][tn][j]
      v4f rb[2][2][2];  // rider: [buffer][half group][head tile]
      auto load_b = [&](int buf, int g) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) b[buf][0][j] = *(gcf4)(wpa + 32 * g + 4 * j);
#pragma unroll
        for (int j = 0; j < 4; ++j) b[buf][1][j] = *(gcf4)(wpb + 32 * g + 4 * j);
        if constexpr (RIDER) {
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            rb[buf][hh][0] = *(gcf4)(hp0 + 32 * g + 16 * hh);
            if constexpr (two) rb[buf][hh][1] = *(gcf4)(hp1 + 32 * g + 16 * hh);
          }
        }
      };
      load_b(0, 0);
      v4f af[2];
      ch_rd128(af[0], a0);
      if constexpr (TM > 1) ch_rd128(af[1], a1);
      int g = 0;
      for (; g + 2 <= G; g += 2) {   // branch-free pairs: the last pair re-requests group G-1 (harmless)
        mfma_group_kc(a0 + 128u * g, a1 + 128u * g, a0 + 128u * (g + 1), a1 + 128u * (g + 1), af, b[0], a16 + 128u * g, rb[0],
                      wpa + 32 * (g + 1), wpb + 32 * (g + 1), hp0 + 32 * (g + 1), hp1 + 32 * (g + 1), b[1], rb[1], rider_tag);
        const int g2 = min(g + 2, G - 1);
        mfma_group_kc(a0 + 128u * (g + 1), a1 + 128u * (g + 1), a0 + 128u * g2, a1 + 128u * g2, af, b[1], a16 + 128u * (g + 1), rb[1],
                      wpa + 32 * g2, wpb + 32 * g2, hp0 + 32 * g2, hp1 + 32 * g2, b[0], rb[0], rider_tag);
      }
      if (g < G) {   // odd count: the last group (its fragments were requested by the pair before it, or by the prologue)
        v4f bd[2][4], rbd[2][2];
        mfma_group_kc(a0 + 128u * g, a1 + 128u * g, a0 + 128u * g, a1 + 128u * g, af, b[0], a16 + 128u * g, rb[0],
                      wpa + 32 * g, wpb + 32 * g, hp0 + 32 * g, hp1 + 32 * g, bd, rbd, rider_tag);
        asm volatile("" ::"v"(bd[0][0]), "v"(bd[1][3]));
      }
      ch_lgkm_wait<0>();   // the last group's look-ahead reads
      asm volatile("" : "+v"(af[0]));
      if constexpr (TM > 1) asm volatile("" : "+v"(af[1]));
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (t < ntail) {
        if (t >= NPRE) load_tail(t, tb0[0], tb1[0]);
        // tail A fragment: k = 32 G + 8 t + 4 lh + c: the lane-half offset inside a tail group is 4 floats, not 16
        const unsigned off = (unsigned)(32 * G + 8 * t) * 4u - (unsigned)(12 * lh) * 4u;
        mfma_tail(a0 + off, a1 + off, tb0[t < NPRE ? t : 0], tb1[t < NPRE ? t : 0]);
      }
    }
    if constexpr (RIDER) {   // the rider's own tail: the k's past the last full group, in guarded 16-k steps
      for (int kb0 = 32 * G; kb0 < K; kb0 += 16) {
        const int kb = kb0 + 4 * kq;
        v4f ra, r0v, r1v = {0.f, 0.f, 0.f, 0.f};
        ch_rd128(ra, a16 + (unsigned)kb0 * 4u);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          r0v[c] = kb + c < K ? hp0[kb0 + c] : 0.f;
          if constexpr (two) r1v[c] = kb + c < K ? hp1[kb0 + c] : 0.f;
        }
        ch_lgkm_wait<0>();
        asm volatile("" : "+v"(ra));
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float av = kb + c < K ? ra[c] : 0.f;   // image columns past K's 8-padding are not defined
          hacc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, r0v[c], hacc[0], 0, 0, 0);
          if constexpr (two) hacc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, r1v[c], hacc[1], 0, 0, 0);
        }
      }
      asm volatile("" ::"v"(a16));
    }
  };
  // ---------------------------------------------------------------- epilogue of a GEMM
  // Accumulator layout (transposed tile): lane (li, lh) holds, for batch row 32 tm + li, the outputs
  //   n0 + 32 tn + 8 q + 4 lh + c       of register 4 q + c of acc[tm][tn]  -  four consecutive columns per register quad.
  auto epilogue = [&](const ChainOp &op, bool fast) __attribute__((always_inline)) {
    const int N = ch_uni(op.N), act = ch_uni(op.act);
    const int out_slot = ch_uni(op.out_slot), out_pitch = ch_uni(op.out_pitch);
    const int row_lo = ch_uni(op.row_lo), row_hi = ch_uni(op.row_hi), shift = ch_uni(op.row_shift);
    const int ldo = ch_uni(op.ldo);
    gcf bias = (gcf)ch_uni(op.bias);
    gf out = (gf)ch_uni(op.out);
    const int n0 = wave * 64;
    // the fast path's bias quads: requested before the barrier, which hides most of their latency
    v4f bq[2][4];
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int q = 0; q < 4; ++q) bq[tn][q] = v4f{0.f, 0.f, 0.f, 0.f};
    if (fast && bias) {
      gcf bp = bias + n0 + 4 * lh;
      if ((reinterpret_cast<uintptr_t>(op.bias) & 15) == 0) {
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
          for (int q = 0; q < 4; ++q) bq[tn][q] = *(gcf4)(bp + 32 * tn + 8 * q);
      } else {
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int c = 0; c < 4; ++c) bq[tn][q][c] = bp[32 * tn + 8 * q + c];
      }
    }
    if (out_slot >= 0) __syncthreads();   // every wave has finished reading the images this op may overwrite
    CH_STAMP();
    if (fast) {
      // the common case (a full 64-column tile per wave, 16-byte aligned rows): straight-line, 16 ds_write_b128 and
      // 16 global_store_dwordx4 per lane, no per-element predicates
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        const int rl = 32 * tm + li, grow = r0 + rl;
        const bool rok = out && grow >= row_lo && grow < row_hi;
        float *drow = &lds[out_slot + rl * out_pitch + n0 + 4 * lh];
        gf orow = out + (long long)(grow - shift) * ldo + n0 + 4 * lh;
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            v4f x;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              x[c] = acc[tm][tn][4 * q + c] + bq[tn][q][c];
              if (act == CHA_LRELU) x[c] = fmaxf(x[c], 0.01f * x[c]);   // == x > 0 ? x : 0.01 x
            }
            *reinterpret_cast<v4f *>(drow + 32 * tn + 8 * q) = x;
            if (rok) *reinterpret_cast<__attribute__((address_space(1))) v4f *>(orow + 32 * tn + 8 * q) = x;
          }
      }
    } else {
      // everything else - partial column tiles, unaligned rows, no LDS image: one element at a time
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int col = n0 + 32 * tn + 8 * (r >> 2) + 4 * lh + (r & 3);
          const bool cok = col < N;
          const float bv = (bias && cok) ? bias[col] : 0.f;
#pragma unroll
          for (int tm = 0; tm < TM; ++tm) {
            const int rl = 32 * tm + li, grow = r0 + rl;
            float x = acc[tm][tn][r] + bv;
            if (act == CHA_LRELU) x = x > 0.f ? x : 0.01f * x;
            if (out_slot >= 0 && cok) lds[out_slot + rl * out_pitch + col] = x;
            if (out && cok && grow >= row_lo && grow < row_hi) out[(long long)(grow - shift) * ldo + col] = x;
          }
        }
    }
    CH_STAMP();
    if (out_slot >= 0) {
      // columns [N, next multiple of 8) of the image stay zero: the next layer's last k-group reads them
      const int npad = ((N + 7) & ~7) - N;
      for (int e = tid; e < BM * npad; e += CH_THREADS) {
        const int r = e / npad, c = e - r * npad;
        lds[out_slot + r * out_pitch + N + c] = 0.f;
      }
      __syncthreads();
      CH_STAMP();
    }
  };
  // is this GEMM's epilogue the straight-line one?  (wave-uniform)
  auto epilogue_fast = [&](const ChainOp &op) __attribute__((always_inline)) {
    const int N = ch_uni(op.N), out_slot = ch_uni(op.out_slot), ldo = ch_uni(op.ldo);
    const float *out = ch_uni(op.out);
    const bool out_ok = !out || ((ldo & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0);
    return out_slot >= 0 && out_ok && wave * 64 + 64 <= N;
  };

  // ---------------------------------------------------------------- the program
  for (int ip = 0; ip < CH_MAX_OPS; ++ip) {
    const ChainOp &op = s_ops[ip];
    const int kind = ch_uni(op.kind);
    if (kind == CH_END) break;
    if (kind == CH_LOAD) {
      const int slot = ch_uni(op.slot), pitch = ch_uni(op.pitch), kpad = ch_uni(op.kpad), nseg = ch_uni(op.nseg);
      __syncthreads();   // readers of the image being replaced are done
      int ktot = 0;
      for (int s = 0; s < nseg; ++s) {
        gcf src = (gcf)ch_uni(op.ld[s].src);
        const int ld = ch_uni(op.ld[s].ld), width = ch_uni(op.ld[s].width), col = ch_uni(op.ld[s].col);
        const bool vec = ((width | ld | col) & 3) == 0 && (reinterpret_cast<uintptr_t>(op.ld[s].src) & 15) == 0;
        // eight loads in flight per thread, then the eight LDS stores (a load -> store loop would expose one memory
        // round trip per iteration: 16 of them for a 256-wide block)
        if (vec) {
          const int wq = width >> 2, total = BM * wq;
          for (int base = 0; base < total; base += 16 * CH_THREADS) {
            v4f v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
              const int e = base + tid + u * CH_THREADS;
              const int r = e / wq, c = (e - r * wq) * 4;
              const int grow = min(r0 + min(r, BM - 1), rows - 1);   // rows beyond the batch repeat its last row (never stored)
              v[u] = *(gcf4)(src + (long long)grow * ld + c);
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
              const int e = base + tid + u * CH_THREADS;
              const int r = e / wq, c = (e - r * wq) * 4;
              if (e < total) *reinterpret_cast<v4f *>(&lds[slot + r * pitch + col + c]) = v[u];
            }
          }
        } else {
          const int total = BM * width;
          for (int base = 0; base < total; base += 8 * CH_THREADS) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const int e = base + tid + u * CH_THREADS;
              const int r = e / width, c = e - r * width;
              const int grow = min(r0 + min(r, BM - 1), rows - 1);
              v[u] = src[(long long)grow * ld + c];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const int e = base + tid + u * CH_THREADS;
              const int r = e / width, c = e - r * width;
              if (e < total) lds[slot + r * pitch + col + c] = v[u];
            }
          }
        }
        ktot = max(ktot, col + width);
      }
      const int npad = kpad - ktot;
      for (int e = tid; e < BM * npad; e += CH_THREADS) {
        const int r = e / npad, c = e - r * npad;
        lds[slot + r * pitch + ktot + c] = 0.f;
      }
      __syncthreads();
    } else if (kind == CH_GEMM) {
      const int N = ch_uni(op.N), flags = ch_uni(op.flags), nseg = ch_uni(op.nseg);
      const int n0 = wave * 64;
      if (flags & CHF_ZERO) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int r = 0; r < 16; ++r) { acc[a][0][r] = 0.f; acc[a][1][r] = 0.f; }
      }
      const float *hw = ch_uni(op.hw);
      if (hw && (flags & CHF_HBEGIN)) { hacc[0] = v4f{0.f, 0.f, 0.f, 0.f}; hacc[1] = v4f{0.f, 0.f, 0.f, 0.f}; }
      const bool fast = (flags & CHF_EMIT) && epilogue_fast(op);
      if (n0 < N) {   // wave-uniform: a wave whose 64 columns lie beyond N has no tile
        if (hw) {   // (the builder attaches a rider only when every wave has a tile: N > 192)
          const int hldw = ch_uni(op.hldw), hN = ch_uni(op.hN);
          int kcol = 0;
          for (int s = 0; s < nseg; ++s) {
            if (hN > 16) gemm_seg_kc(op.seg[s], N, n0, hw + kcol, hldw, hN, std::integral_constant<int, 2>{});
            else gemm_seg_kc(op.seg[s], N, n0, hw + kcol, hldw, hN, std::integral_constant<int, 1>{});
            kcol += ch_uni(op.seg[s].K);
          }
        } else {
          for (int s = 0; s < nseg; ++s) gemm_seg_kc(op.seg[s], N, n0, nullptr, 0, 0, std::integral_constant<int, 0>{});
        }
      }
      if (stamp) g_ch_stamps[nstamp++] = __builtin_amdgcn_s_memtime();
      if (flags & CHF_EMIT) epilogue(op, fast);
    } else if (kind == CH_NARROW) {
      // Narrow outputs (skip heads, N <= 32): every wave computes the COMPLETE K sum for its own 16 rows with
      // v_mfma_f32_16x16x4_f32 - lane (i = l & 15, kq = l >> 4) holds A[row 16 w + i][16 h + 4 kq + c] and
      // B[..][col l & 15] for MFMA step c of 16-k half group h - so there is no cross-wave reduction, no scratch and
      // no barrier, and the accumulators are 4 registers per 16 columns.
      const int N = ch_uni(op.N), flags = ch_uni(op.flags), nseg = ch_uni(op.nseg);
      const int lj = lane & 15, kq = lane >> 4;
      if (flags & CHF_BEGIN) { hacc[0] = v4f{0.f, 0.f, 0.f, 0.f}; hacc[1] = v4f{0.f, 0.f, 0.f, 0.f}; }
      const int n0c = min(lj, N - 1), n1c = min(16 + lj, N - 1);   // clamped columns (never stored when shifted)
      const bool two = N > 16;
      int wslot = ch_uni(op.slot);   // this op's weight slices, staged at program start: per segment [N][wp]
      for (int s = 0; s < nseg; ++s) {
        const int K = ch_uni(op.seg[s].K), pitch = ch_uni(op.seg[s].pitch), slot = ch_uni(op.seg[s].slot);
        const int k16 = (K + 15) & ~15, wp = k16 + 4;
        const unsigned a0 = lds0 + (unsigned)(slot + (16 * (wave & (2 * TM - 1)) + lj) * pitch + 4 * kq) * 4u;
        const unsigned w0 = lds0 + (unsigned)(wslot + n0c * wp + 4 * kq) * 4u, w1 = lds0 + (unsigned)(wslot + n1c * wp + 4 * kq) * 4u;
        wslot += N * wp;
        const int nh = k16 >> 4;
        for (int h0 = 0; h0 < nh; h0 += 4) {   // four 16-k steps per pass: 8 or 12 fragment reads in flight, then their MFMAs
          v4f a[4], b0[4], b1[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const unsigned off = (unsigned)min(h0 + u, nh - 1) * 64u;
            ch_rd128(a[u], a0 + off);
            ch_rd128(b0[u], w0 + off);
            if (two) ch_rd128(b1[u], w1 + off);
          }
          ch_lgkm_wait<0>();
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            asm volatile("" : "+v"(a[u]), "+v"(b0[u]));
            if (two) asm volatile("" : "+v"(b1[u]));
            // steps past the end contribute nothing; image columns beyond K's 8-padding are not defined
            const int kb = 16 * (h0 + u) + 4 * kq;
            if (16 * (h0 + u) + 16 > K) {
#pragma unroll
              for (int c = 0; c < 4; ++c) a[u][c] = (h0 + u < nh && kb + c < K) ? a[u][c] : 0.f;
            }
          }
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              if (u & 1) {
                hodd[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][c], b0[u][c], hodd[0], 0, 0, 0);
                if (two) hodd[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][c], b1[u][c], hodd[1], 0, 0, 0);
              } else {
                hacc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][c], b0[u][c], hacc[0], 0, 0, 0);
                if (two) hacc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][c], b1[u][c], hacc[1], 0, 0, 0);
              }
            }
        }
        asm volatile("" ::"v"(a0), "v"(w0), "v"(w1));
      }
      if (flags & CHF_FINISH) {   // D: column l & 15, rows 4 (l >> 4) + reg of the wave's 16
        const int row_lo = ch_uni(op.row_lo), row_hi = ch_uni(op.row_hi), shift = ch_uni(op.row_shift), ldo = ch_uni(op.ldo);
        gcf bias = (gcf)ch_uni(op.bias);
        gf out = (gf)ch_uni(op.out);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int col = 16 * t + lj;
          if (col < N && wave < 2 * TM) {   // (32-row blocks: the rows live in waves 0 and 1)
            const float bv = bias ? bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int grow = r0 + 16 * wave + 4 * kq + r;
              if (grow >= row_lo && grow < row_hi) out[(long long)(grow - shift) * ldo + col] = (hacc[t][r] + hodd[t][r]) + bv;
            }
          }
        }
        hodd[0] = v4f{0.f, 0.f, 0.f, 0.f};
        hodd[1] = v4f{0.f, 0.f, 0.f, 0.f};
      }
    }
    if (stamp) g_ch_stamps[nstamp++] = __builtin_amdgcn_s_memtime();
  }
  if (stamp) g_ch_stamps[8 * CH_MAX_OPS + 3] = nstamp;
}

int chain_read_stamps(unsigned long long *out, int cap) {
  unsigned long long h[8 * CH_MAX_OPS + 4];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_ch_stamps), sizeof(h)) != hipSuccess) return -1;
  const int n = (int)h[8 * CH_MAX_OPS + 3];
  for (int i = 0; i < n && i < cap; ++i) out[i] = h[i];
  return n < cap ? n : cap;
}
void chain_enable_stamps(int on) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ch_stamps_on), &on, sizeof(on)); }

int chain_finalize(ChainProblem *probs, int nprob, int bm) {
  int total = 0;
  for (int i = 0; i < nprob; ++i) {
    probs[i].block_start = total;
    total += (probs[i].rows + bm - 1) / bm;
  }
  return total;
}

hipError_t chain_launch(const ChainProblem *probs_dev, int nprob, const ChainOp *ops_dev, int total_blocks, int lds_floats, int bm,
                        hipStream_t stream) {
  if (total_blocks <= 0) return hipSuccess;
  if (bm != 32 && bm != CH_BM) return hipErrorInvalidValue;
  // dynamic LDS beyond 64 KiB has to be allowed once per (device, function): a process may drive several GPUs, and two
  // agents may launch from two threads
  static bool attr_set[64];
  static std::mutex attr_mu;
  {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lk(attr_mu);
    if (!attr_set[dev]) {
      e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_chain<2>), hipFuncAttributeMaxDynamicSharedMemorySize, CH_LDS_FLOATS * 4);
      if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_chain<1>), hipFuncAttributeMaxDynamicSharedMemorySize, CH_LDS_FLOATS * 4);
      if (e != hipSuccess) return e;
      attr_set[dev] = true;
    }
  }
  if (bm == 32) hipLaunchKernelGGL(k_chain<1>, dim3(total_blocks), dim3(CH_THREADS), (size_t)lds_floats * 4, stream, probs_dev, nprob, ops_dev);
  else hipLaunchKernelGGL(k_chain<2>, dim3(total_blocks), dim3(CH_THREADS), (size_t)lds_floats * 4, stream, probs_dev, nprob, ops_dev);
  return hipGetLastError();
}

double chain_op_flops(const ChainOp &op, int rows) {
  if (op.kind != CH_GEMM && op.kind != CH_NARROW) return 0.0;
  double k = 0;
  for (int s = 0; s < op.nseg; ++s) k += op.seg[s].K;
  return 2.0 * rows * (double)op.N * k;
}

}  // namespace fdql
