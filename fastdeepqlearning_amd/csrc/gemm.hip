// Grouped, K-segmented fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32).
//
// One launch runs a LIST of independent problems (e.g. layer i of all 15 critic MLPs);
// each problem sums over a list of K-segments so torch.cat((x, h1, h2)) @ W.T never
// materialises the concat (franQ/Agent/models/mlp.py:88-94) and the five critics'
// contributions to d(state) reduce inside one accumulator.  Operands may be K-contiguous
// (activations, torch Linear weights in the forward) or K-strided (weights in dgrad,
// both operands in wgrad).  fp32 in / fp32 accumulate: the MFMA result is bitwise a
// k-ordered fmaf chain, which is what keeps the 1e-5 parity budget against the CPU path.
//
// Tile (default, dense problems): 64x64 per 256-thread workgroup, 4 waves as 2x2, each wave ONE 32x32 MFMA tile
// (16 accumulator VGPRs, 63 VGPR in all -> 8 waves per SIMD; the larger 64x128 / 128x128 tiles stay selectable and
// measured slower: occupancy, not operand reuse, is what this kernel is short of).  K advances in chunks of 16
// through a double-buffered LDS image laid out [k][row] with pitch rows+4 floats, so every fragment read is a
// conflict-free ds_read_b32 of 32 consecutive rows for one k.  Global loads of chunk c+1 are issued before the
// MFMAs of chunk c (interior chunks: all addresses first, then the loads back to back) and written to the other
// LDS buffer after them (one barrier per chunk); a chunk's MFMA k-steps run at raised wave priority.
//
// Two kernels live here: k_gemm_grouped (register-staged, described above: the default for every shape) and, further
// down.  (Round 2's LDS-DMA build of the dense shapes, k_gemm_dma, lost against this kernel on every stage of the update for three
// rounds and was removed in round 5: profiles/r02_* keep its measurements.)
#include "common.h"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>

namespace fdql {

constexpr int GEMM_THREADS = 256;
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Tile shapes (rows x cols per workgroup; 4 waves, each TM x TN MFMA tiles of 32x32):
//   GEMM_64x64  : waves 2x2, wave tile 1x1  — the dense layers (default)
//   GEMM_128x128: waves 2x2, wave tile 2x2  — selectable (FDQL_GEMM_DENSE_SHAPE)
//   GEMM_128x32 : waves 4x1, wave tile 1x1  — narrow outputs (heads with N<=32, d pi, few-column wgrads)
//   GEMM_32x128 : waves 1x4, wave tile 1x1  — few-row outputs (head weight gradients, K-split)
//   GEMM_64x128 : waves 2x2, wave tile 1x2  — dense layers of a single network (M x 256 gives only
//                 196 tiles of 128x128 on 256 CUs; halving the tile doubles the workgroups)
// The narrow shapes waste MFMA lanes on padding but those problems are bandwidth-bound: what
// matters is that they stream their big operand through the same coalesced LDS staging.
#ifndef FDQL_MFMA_PRIO
#define FDQL_MFMA_PRIO 1   // s_setprio around a chunk's MFMA k-steps: waves that have their operands go before waves still
#endif                   // forming addresses (1.770 -> 1.729 ms/update; priority on the operand requests instead: 1.82)
#ifndef FDQL_NARROW_MINB
#define FDQL_NARROW_MINB 7   // occupancy target of the narrow (bandwidth-bound) shapes: 72 VGPR, 7 waves/SIMD (92 / 5 without)
#endif
#ifndef FDQL_DUAL_TN
#define FDQL_DUAL_TN 1   // dual-output tiles: 64 x (64 * FDQL_DUAL_TN)
#endif
template <int SHAPE> struct TileCfg;
template <> struct TileCfg<GEMM_128x128> { static constexpr int WM = 2, WN = 2, TM = 2, TN = 2; static constexpr bool DUAL = false; static constexpr int MINB = 2; static constexpr bool HF = false; };
template <> struct TileCfg<GEMM_128x32> { static constexpr int WM = 4, WN = 1, TM = 1, TN = 1; static constexpr bool DUAL = false; static constexpr int MINB = FDQL_NARROW_MINB; static constexpr bool HF = false; };
template <> struct TileCfg<GEMM_32x128> { static constexpr int WM = 1, WN = 4, TM = 1, TN = 1; static constexpr bool DUAL = false; static constexpr int MINB = FDQL_NARROW_MINB; static constexpr bool HF = false; };
template <> struct TileCfg<GEMM_64x128> { static constexpr int WM = 2, WN = 2, TM = 1, TN = 2; static constexpr bool DUAL = false; static constexpr int MINB = 2; static constexpr bool HF = false; };
// 64x128 tiles, two outputs: C = f(sum over segments <= emit_seg), C2 = f(sum over all segments) -
// critic layer 0 of q(s, a) and q(s, pi) in one pass over s.Ws
//   (no second accumulator: the tile is stored, the tail segments are added, the tile is stored again)
// 64x64: for launches with too few 64x128 tiles to give every SIMD more than one wave (a single network's
// layer at 12.5 k rows is 392 tiles on 256 CUs): twice the workgroups, one 32x32 MFMA tile per wave
template <> struct TileCfg<GEMM_64x64> { static constexpr int WM = 2, WN = 2, TM = 1, TN = 1; static constexpr bool DUAL = false; static constexpr int MINB = 2; static constexpr bool HF = false; };
// 64x64 with the head-fusion epilogue (GemmProblem::hf_*): the hidden layers >= 1 of the critics
template <> struct TileCfg<GEMM_64x64_HF> { static constexpr int WM = 2, WN = 2, TM = 1, TN = 1; static constexpr bool DUAL = false; static constexpr int MINB = 2; static constexpr bool HF = true; };
template <> struct TileCfg<GEMM_64x128_DUAL> { static constexpr int WM = 2, WN = 2, TM = 1, TN = FDQL_DUAL_TN; static constexpr bool DUAL = true; static constexpr int MINB = 8; static constexpr bool HF = true; };

// Pointers that come out of the problem tables are generic to the compiler, which would emit
// FLAT loads: those also count on lgkmcnt, so the `s_waitcnt lgkmcnt(0)` in front of the MFMAs
// (meant for the LDS fragment reads) would wait for the NEXT chunk's prefetch and serialise the
// pipeline.  Casting to the global address space yields global_load/global_store (vmcnt only).
typedef const __attribute__((address_space(1))) float *gcf;
typedef __attribute__((address_space(1))) float *gf;
typedef float v4f __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) v4f *gcf4;

// Values read from the problem table are wave-uniform, but the table lives in memory the kernel may
// also write (hipcc cannot prove otherwise), so they arrive through vector loads.  Pinning them into
// SGPRs makes every branch on them a scalar branch and lets base pointers stay in scalar registers.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ const float *uni(const float *p) {
  const uintptr_t u = reinterpret_cast<uintptr_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
  return reinterpret_cast<const float *>(((uintptr_t)hi << 32) | lo);
}


// Load this thread's share of a [R rows x BKT k] operand chunk: NV float4 slots per thread
// (NV = ceil(R*BKT/4/256)).  kc: element (r,k) at P[r*ld + k]; slot -> row = slot / (BKT/4),
// k = (slot % (BKT/4))*4 + j  (BKT = 32: eight lanes cover one full 128-byte line of a row).
template <int R, int NV, int BKT>
__device__ __forceinline__ void load_chunk_kc(const float *__restrict__ P, int ld, int Rmax, int kend, int r0, int k0,
                                              int tid, float (&v)[4 * NV]) {
  constexpr int KQ = BKT / 4;
#pragma unroll
  for (int h = 0; h < NV; ++h) {
    const int slot = tid + GEMM_THREADS * h;
    const int kq = k0 + (slot % KQ) * 4;
    const int r = r0 + (slot / KQ);
    const bool in_tile = (slot / KQ) < R;
    gcf p = (gcf)(P + (long long)r * ld + kq);
    if (in_tile && r < Rmax && kq + 3 < kend) {
      const v4f x = *(gcf4)p;   // any 4-byte aligned address (see operand_fast)
      v[4 * h + 0] = x.x; v[4 * h + 1] = x.y; v[4 * h + 2] = x.z; v[4 * h + 3] = x.w;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[4 * h + j] = (in_tile && r < Rmax && kq + j < kend) ? p[j] : 0.f;
    }
  }
}

// ks: element (r,k) at P[k*ld + r]; slot -> k = slot / (R/4), rows (slot % (R/4))*4 + j.
template <int R, int NV, int BKT>
__device__ __forceinline__ void load_chunk_ks(const float *__restrict__ P, int ld, int Rmax, int kend, int r0, int k0,
                                              int tid, float (&v)[4 * NV]) {
  constexpr int RQ = R / 4;
#pragma unroll
  for (int h = 0; h < NV; ++h) {
    const int slot = tid + GEMM_THREADS * h;
    const int kl = slot / RQ;
    const int rq = r0 + (slot - kl * RQ) * 4;
    const int k = k0 + kl;
    const bool in_tile = kl < BKT;
    gcf p = (gcf)(P + (long long)k * ld + rq);
    if (in_tile && k < kend && rq + 3 < Rmax) {
      const v4f x = *(gcf4)p;   // any 4-byte aligned address (see operand_fast)
      v[4 * h + 0] = x.x; v[4 * h + 1] = x.y; v[4 * h + 2] = x.z; v[4 * h + 3] = x.w;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[4 * h + j] = (in_tile && k < kend && rq + j < Rmax) ? p[j] : 0.f;
    }
  }
}

// Both operand chunks -> registers.  Interior, aligned chunks (the common case: the tile inside the
// matrix, a whole K-chunk; any row pitch) take a straight-line path: every address is
// formed FIRST, then the loads issue back to back.  The order matters: hipcc reuses the (dead)
// destination registers of the staging loads as address temporaries, and a temporary written while
// an earlier load of this chunk is in flight costs an `s_waitcnt vmcnt(0)` - one exposed global-load
// latency per chunk (seen in the ISA of the guarded loaders below, which remain for edge tiles and
// ragged operands).  All conditions are wave-uniform (problem table + tile position only).
template <int R, int BKT>
__device__ __forceinline__ bool operand_fast(const float *P, int ld, int Rmax, int kend, int r0, int k0) {
  constexpr bool EXACT = (R * BKT / 4) % GEMM_THREADS == 0;
  // gfx950 serves a dwordx4 load from any 4-byte aligned address (HSA runs with unaligned access enabled), so row pitches
  // such as 262 or 273 floats need no special case: requiring ld % 4 == 0 and a 16-byte base here sent every chunk of the
  // [256][262] critic layer-0 weights through the guarded loader and cost layer 0 and d state 12-14 % each
  return EXACT && (r0 + R <= Rmax) && (k0 + BKT <= kend);
}
// per-lane element offset of slot 0 of an operand chunk (constant for a segment) and the uniform
// distance between a thread's slots
template <int R, int BKT>
__device__ __forceinline__ unsigned operand_lane_off(int ld, int kc, int tid) {
  constexpr int KQ = BKT / 4, RQ = R / 4;
  return kc ? (unsigned)((tid / KQ) * ld + (tid % KQ) * 4) : (unsigned)((tid / RQ) * ld + (tid % RQ) * 4);
}
template <int R, int BKT>
__device__ __forceinline__ long long operand_slot_stride(int ld, int kc) {
  constexpr int KQ = BKT / 4, RQ = R / 4;
  return (long long)(kc ? GEMM_THREADS / KQ : GEMM_THREADS / RQ) * ld;
}

template <int BMT, int BNT, int NVA, int NVB, int BKT>
__device__ __forceinline__ void load_operands(const float *A, int lda, int akc, int M, const float *B, int ldb, int bkc,
                                              int N, int kend, int r0, int c0, int k0, int tid, unsigned voa, unsigned vob,
                                              float (&va)[4 * NVA], float (&vb)[4 * NVB]) {
  if (operand_fast<BMT, BKT>(A, lda, M, kend, r0, k0) && operand_fast<BNT, BKT>(B, ldb, N, kend, c0, k0)) {
    gcf baseA = (gcf)(akc ? A + (long long)r0 * lda + k0 : A + (long long)k0 * lda + r0);   // scalar
    gcf baseB = (gcf)(bkc ? B + (long long)c0 * ldb + k0 : B + (long long)k0 * ldb + c0);
    const long long hsa = operand_slot_stride<BMT, BKT>(lda, akc), hsb = operand_slot_stride<BNT, BKT>(ldb, bkc);
    gcf4 pa[NVA], pb[NVB];
#pragma unroll
    for (int h = 0; h < NVA; ++h) pa[h] = (gcf4)(baseA + h * hsa + voa);
#pragma unroll
    for (int h = 0; h < NVB; ++h) pb[h] = (gcf4)(baseB + h * hsb + vob);
    __builtin_amdgcn_sched_barrier(0);
    v4f xa[NVA], xb[NVB];
#pragma unroll
    for (int h = 0; h < NVA; ++h) xa[h] = *pa[h];
#pragma unroll
    for (int h = 0; h < NVB; ++h) xb[h] = *pb[h];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int h = 0; h < NVA; ++h) { va[4 * h] = xa[h].x; va[4 * h + 1] = xa[h].y; va[4 * h + 2] = xa[h].z; va[4 * h + 3] = xa[h].w; }
#pragma unroll
    for (int h = 0; h < NVB; ++h) { vb[4 * h] = xb[h].x; vb[4 * h + 1] = xb[h].y; vb[4 * h + 2] = xb[h].z; vb[4 * h + 3] = xb[h].w; }
    return;
  }
  // Mixed case (the narrow shapes: a 32-row operand is never a whole number of slots per thread; an edge tile in one
  // dimension only): the guarded operand first - its loads wait on each other - then the interior one straight-line, so
  // that its loads are the ones left in flight under the MFMAs.
  if constexpr (BMT != 32 && BNT != 32) {   // dense shapes: both guarded (keeps them at 64 VGPRs)
    if (akc) load_chunk_kc<BMT, NVA, BKT>(A, lda, M, kend, r0, k0, tid, va); else load_chunk_ks<BMT, NVA, BKT>(A, lda, M, kend, r0, k0, tid, va);
    if (bkc) load_chunk_kc<BNT, NVB, BKT>(B, ldb, N, kend, c0, k0, tid, vb); else load_chunk_ks<BNT, NVB, BKT>(B, ldb, N, kend, c0, k0, tid, vb);
    return;
  }
  const bool fa = operand_fast<BMT, BKT>(A, lda, M, kend, r0, k0), fb = operand_fast<BNT, BKT>(B, ldb, N, kend, c0, k0);
  if (!fa) { if (akc) load_chunk_kc<BMT, NVA, BKT>(A, lda, M, kend, r0, k0, tid, va); else load_chunk_ks<BMT, NVA, BKT>(A, lda, M, kend, r0, k0, tid, va); }
  if (!fb) { if (bkc) load_chunk_kc<BNT, NVB, BKT>(B, ldb, N, kend, c0, k0, tid, vb); else load_chunk_ks<BNT, NVB, BKT>(B, ldb, N, kend, c0, k0, tid, vb); }
  if (fa) {
    gcf baseA = (gcf)(akc ? A + (long long)r0 * lda + k0 : A + (long long)k0 * lda + r0);
    const long long hsa = operand_slot_stride<BMT, BKT>(lda, akc);
    gcf4 pa[NVA];
#pragma unroll
    for (int h = 0; h < NVA; ++h) pa[h] = (gcf4)(baseA + h * hsa + voa);
    __builtin_amdgcn_sched_barrier(0);
    v4f xa[NVA];
#pragma unroll
    for (int h = 0; h < NVA; ++h) xa[h] = *pa[h];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int h = 0; h < NVA; ++h) { va[4 * h] = xa[h].x; va[4 * h + 1] = xa[h].y; va[4 * h + 2] = xa[h].z; va[4 * h + 3] = xa[h].w; }
  }
  if (fb) {
    gcf baseB = (gcf)(bkc ? B + (long long)c0 * ldb + k0 : B + (long long)k0 * ldb + c0);
    const long long hsb = operand_slot_stride<BNT, BKT>(ldb, bkc);
    gcf4 pb[NVB];
#pragma unroll
    for (int h = 0; h < NVB; ++h) pb[h] = (gcf4)(baseB + h * hsb + vob);
    __builtin_amdgcn_sched_barrier(0);
    v4f xb[NVB];
#pragma unroll
    for (int h = 0; h < NVB; ++h) xb[h] = *pb[h];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int h = 0; h < NVB; ++h) { vb[4 * h] = xb[h].x; vb[4 * h + 1] = xb[h].y; vb[4 * h + 2] = xb[h].z; vb[4 * h + 3] = xb[h].w; }
  }
}

template <int R, int NV, int BKT>
__device__ __forceinline__ void store_chunk_kc(float *__restrict__ lds, int tid, const float (&v)[4 * NV]) {
  constexpr int PITCH = R + 4, KQ = BKT / 4;
#pragma unroll
  for (int h = 0; h < NV; ++h) {
    const int slot = tid + GEMM_THREADS * h;
    const int kq = (slot % KQ) * 4, r = slot / KQ;
    if (r < R) {
#pragma unroll
      for (int j = 0; j < 4; ++j) lds[(kq + j) * PITCH + r] = v[4 * h + j];
    }
  }
}

template <int R, int NV, int BKT>
__device__ __forceinline__ void store_chunk_ks(float *__restrict__ lds, int tid, const float (&v)[4 * NV]) {
  constexpr int PITCH = R + 4, RQ = R / 4;
#pragma unroll
  for (int h = 0; h < NV; ++h) {
    const int slot = tid + GEMM_THREADS * h;
    const int kl = slot / RQ, rq = (slot - kl * RQ) * 4;
    if (kl < BKT)
      *reinterpret_cast<float4 *>(&lds[kl * PITCH + rq]) = make_float4(v[4 * h], v[4 * h + 1], v[4 * h + 2], v[4 * h + 3]);
  }
}

// ---- LDS fragment reads (see the comment in the main loop)
__device__ __forceinline__ unsigned lds_addr(const float *p) {
  return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float *)p;
}
template <int OFF_BYTES>
__device__ __forceinline__ void lds_read_b32(float &d, unsigned addr) {
  static_assert(OFF_BYTES >= 0 && OFF_BYTES < 65536, "ds_read_b32 offset field is 16 bits");
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=&v"(d) : "v"(addr), "n"(OFF_BYTES));
}
// T fragments of k-step KK: element t at (2*KK*PITCH + 32*t) floats past the lane's base
template <int T, int KK, int PITCH>
__device__ __forceinline__ void frag_read(float (&f)[T], unsigned addr) {
  lds_read_b32<(2 * KK * PITCH) * 4>(f[0], addr);
  if constexpr (T > 1) lds_read_b32<(2 * KK * PITCH + 32) * 4>(f[1], addr);
  if constexpr (T > 2) lds_read_b32<(2 * KK * PITCH + 64) * 4>(f[2], addr);
  if constexpr (T > 3) lds_read_b32<(2 * KK * PITCH + 96) * 4>(f[3], addr);
}
// kk is a compile-time constant after unrolling; dispatch it onto the template parameter
template <int T, int PITCH>
__device__ __forceinline__ void frag_read_dyn(float (&f)[T], unsigned addr, int kk) {
  switch (kk) {
    case 1: frag_read<T, 1, PITCH>(f, addr); break;
    case 2: frag_read<T, 2, PITCH>(f, addr); break;
    case 3: frag_read<T, 3, PITCH>(f, addr); break;
    case 4: frag_read<T, 4, PITCH>(f, addr); break;
    case 5: frag_read<T, 5, PITCH>(f, addr); break;
    case 6: frag_read<T, 6, PITCH>(f, addr); break;
    case 7: frag_read<T, 7, PITCH>(f, addr); break;
    case 8: frag_read<T, 8, PITCH>(f, addr); break;
    case 9: frag_read<T, 9, PITCH>(f, addr); break;
    case 10: frag_read<T, 10, PITCH>(f, addr); break;
    case 11: frag_read<T, 11, PITCH>(f, addr); break;
    case 12: frag_read<T, 12, PITCH>(f, addr); break;
    case 13: frag_read<T, 13, PITCH>(f, addr); break;
    case 14: frag_read<T, 14, PITCH>(f, addr); break;
    case 15: frag_read<T, 15, PITCH>(f, addr); break;
    default: frag_read<T, 0, PITCH>(f, addr); break;
  }
}
template <int OFF_BYTES>
__device__ __forceinline__ void frag_read_b128(v4f &d, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(d) : "v"(addr), "n"(OFF_BYTES));
}
template <int N>
__device__ __forceinline__ void lds_wait() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}
// orders the uses of the fragments after the wait above (an asm the values pass through)
template <int T>
__device__ __forceinline__ void frag_ready(float (&f)[T]) {
#pragma unroll
  for (int t = 0; t < T; ++t) asm volatile("" : "+v"(f[t]));
}

// Head fusion (GemmProblem::hf_*): a wave holds 32 rows x 32 columns of an activation tile - 16 rows per lane half,
// one column per lane (D layout of the 32x32 MFMA tile).  Per group of 8/Q rows the 8 products x * w[q] are summed
// over the 32 column lanes by a halving butterfly (4 + 2 + 1 exchanges, then 2 full steps): lane bits (4,3,2) pick
// which of the 8 sums a lane ends up with, lanes with (lane & 3) == 0 store it.  x is recomputed from the accumulator.
template <int Q>
__device__ __forceinline__ void hf_partial(const f32x16 &acc, float bv, int epi, const float (&wq)[8], bool cok, int lane,
                                           int M, int rbase, gf out, int plane) {
  constexpr int GROWS = 8 / Q;
  const int b4 = (lane >> 4) & 1, b3 = (lane >> 3) & 1, b2 = (lane >> 2) & 1;
  const int jmine = 4 * b4 + 2 * b3 + b2, rr_mine = jmine / Q, q_mine = jmine - rr_mine * Q;
#pragma unroll
  for (int g0 = 0; g0 < 16; g0 += GROWS) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float x = acc[g0 + j / Q] + bv;
      if (epi == EPI_LRELU) x = x > 0.f ? x : 0.01f * x;
      v[j] = cok ? x * wq[j % Q] : 0.f;
    }
    float h4[4], h2[2], h1;
#pragma unroll
    for (int j = 0; j < 4; ++j) h4[j] = (b4 ? v[4 + j] : v[j]) + __shfl_xor(b4 ? v[j] : v[4 + j], 16);
#pragma unroll
    for (int j = 0; j < 2; ++j) h2[j] = (b3 ? h4[2 + j] : h4[j]) + __shfl_xor(b3 ? h4[j] : h4[2 + j], 8);
    h1 = (b2 ? h2[1] : h2[0]) + __shfl_xor(b2 ? h2[0] : h2[1], 4);
    h1 += __shfl_xor(h1, 2);
    h1 += __shfl_xor(h1, 1);
    const int r = g0 + rr_mine;
    const int row = rbase + (r & 3) + 8 * (r >> 2);
    if ((lane & 3) == 0 && row < M) out[((long long)plane * M + row) * Q + q_mine] = h1;
  }
}

template <int SHAPE, int BK>
__global__ __launch_bounds__(GEMM_THREADS, TileCfg<SHAPE>::MINB) void k_gemm_grouped(const GemmProblem *__restrict__ probs, int nprob) {
  using Cfg = TileCfg<SHAPE>;
  constexpr int BM = Cfg::WM * Cfg::TM * 32, BN = Cfg::WN * Cfg::TN * 32;
  constexpr int PA = BM + 4, PB = BN + 4;
  constexpr int NVA = (BM * BK / 4 + GEMM_THREADS - 1) / GEMM_THREADS, NVB = (BN * BK / 4 + GEMM_THREADS - 1) / GEMM_THREADS;
  constexpr int TM = Cfg::TM, TN = Cfg::TN;
  __shared__ __attribute__((aligned(16))) float lds[2][BK * (PA + PB)];

  const int tid = threadIdx.x;
  const int bid = blockIdx.x;
  const int pi = find_problem<GemmProblem, &GemmProblem::tile_start>(probs, nprob, bid, tid & 63);
  const GemmProblem &P = probs[pi];

  const int M = uni(P.M), N = uni(P.N), nseg = uni(P.nseg), ksplit = uni(P.ksplit);
  const int tiles_n = uni(P.tiles_n);
  const int local = bid - uni(P.tile_start);
  const int tiles_mn = uni(P.tiles_m) * tiles_n;
  const int split = local / tiles_mn;
  const int rem = local - split * tiles_mn;
  // Tile order inside a problem.  Workgroup ids go round-robin over the 8 XCDs (each with its own L2), so with the
  // plain order the tiles_n column tiles of one row tile sit on different XCDs and every one of them pulls the same
  // A rows through the fabric.  Rows are therefore taken in groups of 8: ids g*8*tiles_n + [0, 8) are the 8 row tiles
  // of column 0, the next 8 ids column 1, ... - the column tiles of a row tile are 8 ids apart = on ONE XCD, a few
  // dispatch slots from each other, and share the A tile in that L2.  The XCD load stays what it was.
  int tile_m, tile_n;
  {
    const int tiles_m = uni(P.tiles_m), full = (tiles_m >> 3) << 3;
    if (rem < full * tiles_n) {
      const int grp = rem / (8 * tiles_n), r = rem - grp * 8 * tiles_n;
      tile_n = r >> 3;
      tile_m = grp * 8 + (r & 7);
    } else {
      const int r = rem - full * tiles_n;
      tile_m = full + r / tiles_n;
      tile_n = r - (r / tiles_n) * tiles_n;
    }
  }
  const int r0 = tile_m * BM, c0 = tile_n * BN;

  // K range of segment 0 when the problem is K-split (wgrad); whole segments otherwise.
  const int K0 = uni(P.seg[0].K);
  int kb0 = 0, ke0 = K0;
  if (ksplit > 1) {
    const int per = ((K0 + ksplit - 1) / ksplit + BK - 1) / BK * BK;
    kb0 = split * per;
    ke0 = min(K0, kb0 + per);
  }

  constexpr bool DUAL = Cfg::DUAL;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  // DUAL: the pipelined main loop covers segments <= emit_seg; the (narrow) tail segments follow the
  // first tile store in a plain load / stage / multiply loop
  const int nseg_all = nseg;
  const int nseg_main = DUAL ? uni(P.emit_seg) + 1 : nseg;

  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
  const int li = lane & 31, lh = lane >> 5;

  // first non-empty chunk
  int s = 0, k = kb0, ke = ke0;
  while (s < nseg_main && k >= ke) {
    ++s;
    if (s < nseg_main) { k = 0; ke = uni(P.seg[s].K); }
  }
  bool have = (s < nseg_main) && (ksplit == 1 || s == 0);

  // Descriptor of the segment chunks are being loaded from, held in (scalar) registers and re-read
  // from the problem table only at a segment switch - not once per chunk.
  const float *sA = nullptr, *sB = nullptr;
  int slda = 0, sldb = 0, sakc = 1, sbkc = 1;
  unsigned voa = 0, vob = 0;   // per-lane offsets of the straight-line loader, constant for a segment
  auto fetch_seg = [&](int idx) {
    const GemmSeg &S = P.seg[idx];
    sA = uni(S.A); sB = uni(S.B); slda = uni(S.lda); sldb = uni(S.ldb); sakc = uni(S.a_kc); sbkc = uni(S.b_kc);
    voa = operand_lane_off<BM, BK>(slda, sakc, tid);
    vob = operand_lane_off<BN, BK>(sldb, sbkc, tid);
  };
  float va[4 * NVA], vb[4 * NVB];
  auto stage = [&](int buf, int akc, int bkc) {
    if (akc) store_chunk_kc<BM, NVA, BK>(lds[buf], tid, va);
    else store_chunk_ks<BM, NVA, BK>(lds[buf], tid, va);
    if (bkc) store_chunk_kc<BN, NVB, BK>(lds[buf] + BK * PA, tid, vb);
    else store_chunk_ks<BN, NVB, BK>(lds[buf] + BK * PA, tid, vb);
  };
  if (have) {
    fetch_seg(s);
    load_operands<BM, BN, NVA, NVB, BK>(sA, slda, sakc, M, sB, sldb, sbkc, N, ke, r0, c0, k, tid, voa, vob, va, vb);
    stage(0, sakc, sbkc);
  }
  __syncthreads();

  int cur = 0;
  int c_akc = sakc, c_bkc = sbkc;   // operand forms of the chunk staged in lds[cur] (F128 picks its fragment reads by them)
  // MFMAs of the chunk staged in lds[cur] (k, ke: its position in the current segment)
  auto run_chunk = [&]() {
    // Fragment reads are written as inline `ds_read_b32` with EARLY-CLOBBER destinations and explicit
    // lgkmcnt waits.  Reason (observed twice on gfx950, reproducible, LDS contents verified by a dump):
    // when hipcc allocates a fragment's destination VGPR on top of that read's own address VGPR
    // (`ds_read_b32 v45, v45 offset:...`), lanes 27/31/59/63 of the wave can come back with wrong data,
    // depending on what surrounds the read - parity tests then fail on output rows 27 and 31 of a
    // tile.  With "=&v" the destination can never alias the address, whatever the allocator does;
    // immediates (<= 64 KB) address the whole staged tile from two base registers.
    const unsigned la = lds_addr(lds[cur] + wm * (TM * 32) + li + lh * PA);
    const unsigned lb = lds_addr(lds[cur] + BK * PA + wn * (TN * 32) + li + lh * PB);
    constexpr int NS = BK / 2;
    // k-steps that hold data: a ragged last chunk (or a narrow segment such as the Q columns of dz or
    // the 6 action columns) stops early instead of multiplying staged zeros
    const int ksteps = min(NS, (ke - k + 1) >> 1);
    float a[2][TM], b[2][TN];
    frag_read<TM, 0, PA>(a[0], la);
    frag_read<TN, 0, PB>(b[0], lb);
#if FDQL_MFMA_PRIO
    __builtin_amdgcn_s_setprio(FDQL_MFMA_PRIO);
#endif
#pragma unroll
    for (int kk = 0; kk < NS; ++kk) {
      if (kk >= ksteps) {   // wave-uniform
        lds_wait<0>();      // drain the prefetch issued by the previous step
        break;
      }
      float(&ac)[TM] = a[kk & 1];
      float(&bc)[TN] = b[kk & 1];
      if (kk + 1 < NS) {
        // next k-step's fragments are requested before this step's MFMAs issue; LDS returns in order,
        // so "at most TM+TN reads outstanding" means this step's fragments have landed
        frag_read_dyn<TM, PA>(a[(kk + 1) & 1], la, kk + 1);
        frag_read_dyn<TN, PB>(b[(kk + 1) & 1], lb, kk + 1);
        lds_wait<TM + TN>();
      } else {
        lds_wait<0>();
      }
      frag_ready<TM>(ac);
      frag_ready<TN>(bc);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[tm], bc[tn], acc[tm][tn], 0, 0, 0);
    }
    // the two base registers stay reserved until every read of the chunk has returned, so no
    // fragment destination is ever allocated on top of an address still in use by an in-flight read
    asm volatile("" ::"v"(la), "v"(lb));
#if FDQL_MFMA_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
  };
  while (have) {
    // locate the next chunk
    int ns = s, nk = k + BK, nke = ke;
    if (nk >= ke) {
      ns = s + 1;
      nk = 0;
      while (ns < nseg_main && uni(P.seg[ns].K) <= 0) ++ns;
      if (ns < nseg_main) nke = uni(P.seg[ns].K);
    }
    const bool has_next = (ns < nseg_main) && (ksplit == 1 || ns == 0);
    if (has_next) {
      if (ns != s) fetch_seg(ns);
      load_operands<BM, BN, NVA, NVB, BK>(sA, slda, sakc, M, sB, sldb, sbkc, N, nke, r0, c0, nk, tid, voa, vob, va, vb);
    }
    const int n_akc = sakc, n_bkc = sbkc;   // layout of the chunk just requested (stored below)

    run_chunk();

    if (!has_next) break;
    stage(cur ^ 1, n_akc, n_bkc);
    __syncthreads();
    cur ^= 1;
    c_akc = n_akc; c_bkc = n_bkc;
    s = ns; k = nk; ke = nke;
  }

  // epilogue: D[i][j] of a 32x32 tile sits at col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  const int epi = P.epi;
  gcf bias = (gcf)P.bias;
  gcf ref = (gcf)P.ref;
  const int ldref = P.ldref;
  float csum[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) csum[tn] = 0.f;
  auto store_tile = [&](gf C, int ldc, bool second) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        const int col = c0 + (wn * TN + tn) * 32 + li;
        if (col >= N) continue;
        const float bv = bias ? bias[col] : 0.f;
        // the 16 reference values of the tile first, then the stores: C and ref may alias as far as hipcc knows, so a
        // load placed after a store waits for it - one exposed memory round trip per element
        float rv[16];
        if (epi == EPI_LRELU_GRAD || epi == EPI_ADD_REF) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = r0 + (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            rv[r] = row < M ? ref[(long long)row * ldref + col] : 0.f;
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = r0 + (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (row >= M) continue;
          float x = acc[tm][tn][r];
          x += bv;
          if (epi == EPI_LRELU) x = x > 0.f ? x : 0.01f * x;
          else if (epi == EPI_LRELU_GRAD) x = rv[r] > 0.f ? x : 0.01f * x;
          else if (epi == EPI_ADD_REF) x += rv[r];
          C[(long long)row * ldc + col] = x;
          if (!second) csum[tn] += x;
        }
      }
    }
    if constexpr (Cfg::HF && TM == 1 && TN == 1) {
      gcf hw = (gcf)P.hf_w;
      if (hw) {
        const int Q = P.hf_q, ldw = P.hf_ldw;
        gf out = (gf)(second ? P.hf_out2 : P.hf_out);
        const int col = c0 + wn * 32 + li;
        const bool cok = col < N;
        const float bv = (bias && cok) ? bias[col] : 0.f;
        float wq[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) wq[q] = (q < Q && cok) ? hw[(long long)q * ldw + col] : 0.f;
        const int plane = (c0 / 64) * 2 + wn, rbase = r0 + wm * 32 + 4 * lh;
        if (Q == 2) hf_partial<2>(acc[0][0], bv, epi, wq, cok, lane, M, rbase, out, plane);
        else if (Q == 1) hf_partial<1>(acc[0][0], bv, epi, wq, cok, lane, M, rbase, out, plane);
        else if (Q == 4) hf_partial<4>(acc[0][0], bv, epi, wq, cok, lane, M, rbase, out, plane);
        else hf_partial<8>(acc[0][0], bv, epi, wq, cok, lane, M, rbase, out, plane);
      }
    }
  };
  store_tile((gf)(P.C + (long long)split * P.split_stride), P.ldc, false);
  if constexpr (DUAL) {
    // tail segments (e.g. the 6 columns of pi - a): accumulate on top of the stored sum, store again
    for (int ts = nseg_main; ts < nseg_all; ++ts) {
      fetch_seg(ts);
      ke = uni(P.seg[ts].K);
      for (k = 0; k < ke; k += BK) {
        __syncthreads();   // every wave is done reading the staging buffer
        load_operands<BM, BN, NVA, NVB, BK>(sA, slda, sakc, M, sB, sldb, sbkc, N, ke, r0, c0, k, tid, voa, vob, va, vb);
        stage(0, sakc, sbkc);
        c_akc = sakc; c_bkc = sbkc;
        __syncthreads();
        cur = 0;
        run_chunk();
      }
    }
    if (nseg_all > nseg_main) store_tile((gf)P.C2, P.ldc2, true);   // a problem without tail is an ordinary one
  }
  // optional column sums (bias gradients): fixed-order reduction through LDS, one partial row
  // per 64 output rows: colsum[row/64][N]
  if (P.colsum) {
    constexpr int RW = TM * 32;                       // rows per wave
    constexpr int WPG = RW >= 64 ? 1 : 64 / RW;       // waves (along M) per 64-row block
    constexpr int GROUPS = (BM + 63) / 64;
    __syncthreads();  // every wave is done reading the staging buffers
    float *red = lds[0];  // [WM][2][BN]
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) red[(wm * 2 + lh) * BN + (wn * TN + tn) * 32 + li] = csum[tn];
    __syncthreads();
    for (int e = tid; e < GROUPS * BN; e += GEMM_THREADS) {
      const int g = e / BN, c = e - g * BN;
      if (c0 + c < N && r0 + g * 64 < M) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < WPG * 2; ++j) t += red[(g * WPG * 2 + j) * BN + c];
        ((gf)P.colsum)[(long long)(r0 / 64 + g) * N + c0 + c] = t;
      }
    }
  }
}

static void shape_dims(int shape, int &bm, int &bn) {
  if (shape == GEMM_SMALL) { bm = 64; bn = 32; return; }
  bm = shape == GEMM_32x128 ? 32 : ((shape == GEMM_64x128 || shape == GEMM_64x128_DUAL || shape == GEMM_64x64 || shape == GEMM_64x64_HF) ? 64 : 128);
  bn = shape == GEMM_128x32 ? 32 : ((shape == GEMM_64x64 || shape == GEMM_64x64_HF) ? 64 : (shape == GEMM_64x128_DUAL ? 64 * FDQL_DUAL_TN : 128));
}

int gemm_finalize(GemmProblem *probs, int nprob, int shape) {
  int bm, bn;
  shape_dims(shape, bm, bn);
  int total = 0;
  for (int i = 0; i < nprob; ++i) {
    GemmProblem &p = probs[i];
    p.tiles_m = (p.M + bm - 1) / bm;
    p.tiles_n = (p.N + bn - 1) / bn;
    if (p.ksplit < 1) p.ksplit = 1;
    p.tile_start = total;
    total += p.tiles_m * p.tiles_n * p.ksplit;
  }
  return total;
}

// Dense problems run on 64x128 tiles: measured on MI355X (config 2) they beat 128x128 on every
// stage (84->88 TF on the 15-critic layers, 53->64 TF on the dgrad with the LeakyReLU' epilogue):
// more, smaller workgroups (5 per CU at 92 VGPRs / 25.6 KB LDS) hide the per-chunk load latency
// better than the bigger tile's higher MFMA:LDS ratio helps.  `prefer_128` keeps the big tile
// selectable for experiments (FDQL_GEMM_DENSE_SHAPE).
static int g_dense_shape = GEMM_64x64;
void gemm_set_dense_shape(int shape) { g_dense_shape = shape; }
int gemm_dense_shape() { return g_dense_shape; }   // (test / tuning hook: fdql_debug_set_gemm_dense_shape)

int gemm_pick_shape(const GemmProblem &p, int dense_shape) {
  if (dense_shape == GEMM_SMALL) {   // test hook: the small-batch kernel wherever it has the form, else the default shapes
    if (gemm_small_takes(p)) return GEMM_SMALL;
    dense_shape = GEMM_64x64;
  }
  if (p.emit_seg >= 0) return GEMM_64x128_DUAL;
  if (p.hf_w) return GEMM_64x64_HF;
  if (p.N <= 32) return GEMM_128x32;
  if (p.M <= 32 && !p.colsum) return GEMM_32x128;
  return dense_shape;
}

bool gemm_shape_is_dense(int shape) { return shape == GEMM_128x128 || shape == GEMM_64x128 || shape == GEMM_64x64; }

double gemm_flops(const GemmProblem &p) {
  double k = 0;
  for (int s = 0; s < p.nseg; ++s) k += p.seg[s].K;
  return 2.0 * p.M * (double)p.N * k;
}

double gemm_bytes(const GemmProblem &p) {
  double b = 0;
  for (int s = 0; s < p.nseg; ++s) b += 4.0 * p.seg[s].K * ((double)p.M + p.N);
  return b + 4.0 * p.M * (double)p.N * p.ksplit;
}

// K-chunk 32 for a SMALL launch of the 64x64 shapes (at most two workgroups per CU: the launch is as long as one workgroup's K
// loop, and half the chunks and barriers shorten that - temporal_len 2 0.380 -> 0.353 ms, one rank's share of config 4 at 128
// windows 1.068 -> 1.058 ms); K-chunk 16 for big launches (81 instead of 63 VGPR cost them two waves per SIMD: config 4 at
// B = 1024 6.09 -> 6.26 ms with 32 everywhere) and for the narrow shapes (116 VGPR at 32).  (The other builds of the main loop -
// K-chunk 8, b128 fragments, no fragment prefetch - lost every measurement of rounds 2-4 and are gone.)
template <int SHAPE>
static void launch_shape(const GemmProblem *probs_dev, int nprob, int total_blocks, hipStream_t stream) {
  const dim3 g(total_blocks), b(GEMM_THREADS);
  if constexpr (SHAPE == GEMM_64x64 || SHAPE == GEMM_64x64_HF) {
    if (total_blocks <= 512) { hipLaunchKernelGGL((k_gemm_grouped<SHAPE, 32>), g, b, 0, stream, probs_dev, nprob); return; }
  }
  hipLaunchKernelGGL((k_gemm_grouped<SHAPE, 16>), g, b, 0, stream, probs_dev, nprob);
}

hipError_t gemm_launch(const GemmProblem *probs_dev, int nprob, int total_blocks, int shape, hipStream_t stream) {
  if (total_blocks <= 0) return hipSuccess;
  if (shape == GEMM_SMALL) return gemm_small_launch(probs_dev, nprob, total_blocks, stream);
  if (shape == GEMM_128x128) launch_shape<GEMM_128x128>(probs_dev, nprob, total_blocks, stream);
  else if (shape == GEMM_128x32) launch_shape<GEMM_128x32>(probs_dev, nprob, total_blocks, stream);
  else if (shape == GEMM_64x128) launch_shape<GEMM_64x128>(probs_dev, nprob, total_blocks, stream);
  else if (shape == GEMM_64x128_DUAL) launch_shape<GEMM_64x128_DUAL>(probs_dev, nprob, total_blocks, stream);
  else if (shape == GEMM_64x64) launch_shape<GEMM_64x64>(probs_dev, nprob, total_blocks, stream);
  else if (shape == GEMM_64x64_HF) launch_shape<GEMM_64x64_HF>(probs_dev, nprob, total_blocks, stream);
  else launch_shape<GEMM_32x128>(probs_dev, nprob, total_blocks, stream);
  return hipGetLastError();
}

}  // namespace fdql
