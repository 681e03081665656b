// Non-GEMM kernels of the SAC/TQC update: skinny heads, policy sampling, the TQC/SAC
// loss with its analytic gradient, slab reduction, Adam + polyak.  gfx950, wave64.
#include "common.h"
#include "update_kernels.h"
#include "dmath.h"
#include <algorithm>
#include <type_traits>

namespace fdql {

// ======================================================================================
// Skinny wgrad / column sums: dW[q, k] (slab) = sum_{m in split} dY[m, q] * X[m, k]
// Block = 64 columns x 4 row lanes (one wave per row lane, so dY[m, q] is wave-uniform);
// LDS reduction over the 4 row lanes; one slab per split (deterministic, no atomics).
// dY == null: dY = 1 (column sums of X -> bias gradients), Nout = 1.
// ======================================================================================
constexpr int SW_COLS = 64, SW_THREADS = 256;

template <int NOUT_MAX>
__global__ __launch_bounds__(SW_THREADS) void k_skinny_wgrad(const SkinnyWgradProblem *__restrict__ probs, int nprob) {
  __shared__ float red[3][NOUT_MAX][SW_COLS];
  const int bid = blockIdx.x;
  const int pi = find_problem<SkinnyWgradProblem, &SkinnyWgradProblem::block_start>(probs, nprob, bid, threadIdx.x & 63);
  const SkinnyWgradProblem &P = probs[pi];
  const int local = bid - P.block_start;
  const int split = local / P.col_blocks;
  const int cb = local - split * P.col_blocks;
  const int tid = threadIdx.x, lane = tid & 63, rg = tid >> 6;
  const int k = cb * SW_COLS + lane;
  const int Nout = P.Nout;
  const int per = (P.M + P.nsplit - 1) / P.nsplit;
  const int m0 = split * per, m1 = min(P.M, m0 + per);
  const bool kin = k < P.K;

  typedef const __attribute__((address_space(1))) float *gcf;   // table pointers: force global_load (not flat)
  typedef __attribute__((address_space(1))) float *gf;
  gcf X = (gcf)P.X;
  gcf dY = (gcf)P.dY;
  if (!dY && P.K <= 16 && P.col_blocks == 1) {
    // narrow column sums (d z, d logits): threads take different ROWS, then reduce
    float a16[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) a16[c] = 0.f;
    for (int m = m0 + tid; m < m1; m += SW_THREADS) {
      gcf x = X + (long long)m * P.ldx;
#pragma unroll
      for (int c = 0; c < 16; ++c)
        if (c < P.K) a16[c] += x[c];
    }
    __shared__ float nred[SW_THREADS / 64][16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      float v = a16[c];
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
      if (lane == 0) nred[rg][c] = v;
    }
    __syncthreads();
    if (tid < P.K) {
      gf dst = (gf)(P.dW + (long long)split * P.split_stride);
      dst[(long long)tid * P.sk] = ((nred[0][tid] + nred[1][tid]) + nred[2][tid]) + nred[3][tid];
    }
    return;
  }
  if (dY && Nout <= 2 && P.K <= 8 && P.col_blocks == 1) {
    // tiny outer products (a skip head's rows over the action columns: 2 x 6): threads take different ROWS, then reduce -
    // one column per thread would be 100 dependent round trips for a few kilobytes
    float a2[2][8];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int c = 0; c < 8; ++c) a2[q][c] = 0.f;
    for (int m = m0 + tid; m < m1; m += SW_THREADS) {
      gcf x = X + (long long)m * P.ldx, d = dY + (long long)m * P.lddy;
      float xv[8], dv[2];
#pragma unroll
      for (int c = 0; c < 8; ++c) xv[c] = c < P.K ? x[c] : 0.f;
#pragma unroll
      for (int q = 0; q < 2; ++q) dv[q] = q < Nout ? d[q] : 0.f;
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int c = 0; c < 8; ++c) a2[q][c] = fmaf(dv[q], xv[c], a2[q][c]);
    }
    __shared__ float pred[SW_THREADS / 64][16];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        float v = a2[q][c];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0) pred[rg][8 * q + c] = v;
      }
    __syncthreads();
    if (tid < 16 && (tid >> 3) < Nout && (tid & 7) < P.K) {
      gf dst = (gf)(P.dW + (long long)split * P.split_stride);
      dst[(long long)(tid >> 3) * P.sq + (long long)(tid & 7) * P.sk] = ((pred[0][tid] + pred[1][tid]) + pred[2][tid]) + pred[3][tid];
    }
    return;
  }
  float acc[NOUT_MAX];
#pragma unroll
  for (int q = 0; q < NOUT_MAX; ++q) acc[q] = 0.f;
  // four rows per round (rows m, m + 4, m + 8, m + 12 of this row lane), their loads requested together: a loop of single
  // rows is one memory round trip per row
  if (!dY) {   // column sums: sixteen rows per round (one register per request)
    for (int m = m0 + rg; m < m1; m += 64) {
      float xv[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) xv[u] = (kin && X) ? X[(long long)min(m + 4 * u, m1 - 1) * P.ldx + k] : 1.f;
      asm volatile("" ::: "memory");
#pragma unroll
      for (int u = 0; u < 16; ++u) acc[0] += m + 4 * u < m1 ? xv[u] : 0.f;
    }
  } else
  for (int m = m0 + rg; m < m1; m += 16) {
    float xv[4];
    gcf dp[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int mc = min(m + 4 * u, m1 - 1);
      xv[u] = (kin && X) ? X[(long long)mc * P.ldx + k] : 1.f;
      dp[u] = dY + (long long)mc * P.lddy;
    }
    if (dY) {
      float dv[4][NOUT_MAX];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int q = 0; q < NOUT_MAX; ++q) dv[u][q] = q < Nout ? dp[u][q] : 0.f;
      asm volatile("" ::: "memory");
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float xu = m + 4 * u < m1 ? xv[u] : 0.f;
#pragma unroll
        for (int q = 0; q < NOUT_MAX; ++q) acc[q] = fmaf(dv[u][q], xu, acc[q]);
      }
    } else {
      asm volatile("" ::: "memory");
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[0] += m + 4 * u < m1 ? xv[u] : 0.f;
    }
  }
  if (rg > 0) {
#pragma unroll
    for (int q = 0; q < NOUT_MAX; ++q)
      if (q < Nout) red[rg - 1][q][lane] = acc[q];
  }
  __syncthreads();
  if (rg == 0 && kin) {
    gf dst = (gf)(P.dW + (long long)split * P.split_stride);
#pragma unroll
    for (int q = 0; q < NOUT_MAX; ++q)
      if (q < Nout) dst[(long long)q * P.sq + (long long)k * P.sk] = ((acc[q] + red[0][q][lane]) + red[1][q][lane]) + red[2][q][lane];
  }
}

int skinny_wgrad_finalize(SkinnyWgradProblem *p, int n) {
  int total = 0;
  for (int i = 0; i < n; ++i) {
    p[i].col_blocks = (p[i].K + SW_COLS - 1) / SW_COLS;
    p[i].block_start = total;
    total += p[i].col_blocks * p[i].nsplit;
  }
  return total;
}

hipError_t skinny_wgrad_launch_host(const SkinnyWgradProblem *host, const SkinnyWgradProblem *dev, int n,
                                    int total_blocks, hipStream_t s) {
  if (total_blocks <= 0) return hipSuccess;
  int max_out = 1;
  for (int i = 0; i < n; ++i) max_out = host[i].Nout > max_out ? host[i].Nout : max_out;
  if (max_out <= 4) hipLaunchKernelGGL(k_skinny_wgrad<4>, dim3(total_blocks), dim3(SW_THREADS), 0, s, dev, n);
  else if (max_out <= 16) hipLaunchKernelGGL(k_skinny_wgrad<16>, dim3(total_blocks), dim3(SW_THREADS), 0, s, dev, n);
  else hipLaunchKernelGGL(k_skinny_wgrad<32>, dim3(total_blocks), dim3(SW_THREADS), 0, s, dev, n);
  return hipGetLastError();
}

// ======================================================================================
// Streaming narrow weight gradients: dW[q, k] (slab) = sum_{m in split} dY[m, q] * X[m, k] for X of exactly 256 columns
// (row pitch 256) and Nout <= 32 - the skip heads' rows over a 256-wide activation, the few input columns (observation,
// action) of a 256-wide layer with the roles swapped.  HBM-bound by construction: a single-wave workgroup = 64 columns
// [64 w, 64 w + 64) of one K-split slab of one problem for ALL rows of the split, so nothing is reduced across waves,
// nothing goes through LDS, and the number of waves does not hang on the K-split alone.  A row is ONE
// v_mfma_f32_4x4x1_16b_f32 per four outputs (8 cycles): lane l holds X[m][64 w + l] (a 256-byte coalesced load per row) as
// the B operand of its block l / 4, and the A operand - dY[m][4 g .. 4 g + 3] for all 16 blocks - is BROADCAST by the
// instruction itself (CBSZ = 4, ABID = r) from a register that holds the dY quads of 16 rows (one load per 16 rows).  The
// result needs no rearrangement: register i of lane l is dW[4 g + i][64 w + l].  (First version, round 3: row pairs as
// v_mfma_f32_32x32x2_f32 - 64 cycles per row on two dependent accumulator chains, 18 us of MFMA latency per wave.)
// ======================================================================================
#ifndef FDQL_SW2_R
#define FDQL_SW2_R 32
#endif
constexpr int SW2_R = FDQL_SW2_R;   // rows per request round and wave (two rounds in flight); multiple of 16

template <int I, int N, typename F>
__device__ __forceinline__ void sw2_for(F &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    sw2_for<I + 1, N>(f);
  }
}

template <int NG>   // groups of four outputs
__device__ __forceinline__ void stream_wgrad_body(const SkinnyWgradProblem &P, int split, int wave) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  typedef const __attribute__((address_space(1))) float *gcf;
  typedef __attribute__((address_space(1))) float *gf;
  const int lane = threadIdx.x;
  const int Nout = P.Nout, M = P.M, lddy = P.lddy;
  const int per = (M + P.nsplit - 1) / P.nsplit;
  const int m0 = split * per, m1 = min(M, m0 + per);
  gcf X = (gcf)P.X + 64 * wave + lane;
  // dY quads of 16 rows: lane l <-> (row l / 4 of the 16, output 4 g + l % 4)
  const int dq = lane & 3, dr = lane >> 2;
  gcf dY = (gcf)P.dY;
  v4f acc[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) acc[g] = v4f{0.f, 0.f, 0.f, 0.f};
  // rows past the split read its last row (a valid address); their dY is zeroed where it is USED (a select next to the load
  // would make every request wait for its own answer)
  auto load = [&](int r0, float (&x)[SW2_R], float (&d)[NG][SW2_R / 16]) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < SW2_R; ++u) x[u] = X[(long long)min(r0 + u, m1 - 1) * 256];
#pragma unroll
    for (int t = 0; t < SW2_R / 16; ++t)
#pragma unroll
      for (int g = 0; g < NG; ++g) d[g][t] = dY[(long long)min(r0 + 16 * t + dr, m1 - 1) * lddy + min(4 * g + dq, Nout - 1)];
  };
  auto use = [&](int r0, const float (&x)[SW2_R], const float (&d)[NG][SW2_R / 16]) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < SW2_R / 16; ++t) {
      float dv[NG];
#pragma unroll
      for (int g = 0; g < NG; ++g) dv[g] = (r0 + 16 * t + dr < m1 && 4 * g + dq < Nout) ? d[g][t] : 0.f;
      sw2_for<0, 16>([&](auto rc) __attribute__((always_inline)) {   // (ABID is an immediate)
        constexpr int r = decltype(rc)::value;
#pragma unroll
        for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(dv[g], x[16 * t + r], acc[g], 4, r, 0);
      });
    }
  };
  if (m0 < m1) {
    float xa[SW2_R], xb[SW2_R], da[NG][SW2_R / 16], db[NG][SW2_R / 16];
    // (the empty asm statements pin the order requests-then-uses: the scheduler otherwise sinks every load next to its use)
    load(m0, xa, da);
    asm volatile("" ::: "memory");
#pragma unroll 1
    for (int g = m0; g < m1; g += 2 * SW2_R) {
      load(g + SW2_R, xb, db);   // (past the end: clamped rows, used with dY = 0 or not at all)
      asm volatile("" ::: "memory");
      use(g, xa, da);
      asm volatile("" ::: "memory");
      if (g + SW2_R < m1) {
        load(g + 2 * SW2_R, xa, da);
        asm volatile("" ::: "memory");
        use(g + SW2_R, xb, db);
        asm volatile("" ::: "memory");
      }
    }
  }
  // register i of lane l = dW[4 g + i][64 w + l].  A split without rows writes zeros (the slab sum runs over every slab).
  gf dst = (gf)(P.dW + (long long)split * P.split_stride);
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (4 * g + i < Nout) dst[(long long)(4 * g + i) * P.sq + (long long)(64 * wave + lane) * P.sk] = acc[g][i];
}

// Both sides narrow (K <= 32 input columns, Nout <= 32 outputs: a 25-quantile head's rows over 17 action columns): one wave per
// slab, a row pair = one v_mfma_f32_32x32x2_f32 (A: lane (q, parity) = dY[m + parity][q], B: lane (k, parity) = X[m + parity][k]),
// 32 pairs per request round.  (On the tile kernel these were a 0.16 ms launch for 42 MB at config 4: one memory round trip per
// 16 rows; one column per thread in k_skinny_wgrad: 0.18 ms.)
constexpr int STREAM_COLSUM_NARROW_MAX = 40;   // column sums with lanes over rows: up to this many columns
constexpr int SW2_TP = 16;   // row pairs per request round
__device__ __forceinline__ void stream_wgrad_tiny(const SkinnyWgradProblem &P, int split) {
  typedef float f32x16 __attribute__((ext_vector_type(16)));
  typedef const __attribute__((address_space(1))) float *gcf;
  typedef __attribute__((address_space(1))) float *gf;
  const int lane = threadIdx.x, li = lane & 31, lh = lane >> 5;
  const int Nout = P.Nout, K = P.K, M = P.M;
  const int per = (((M + P.nsplit - 1) / P.nsplit) + 1) & ~1;   // even: row pairs
  const int m0 = split * per, m1 = min(M, m0 + per);
  gcf X = (gcf)P.X + min(li, K - 1), dY = (gcf)P.dY + min(li, Nout - 1);
  const int ldx = P.ldx, lddy = P.lddy;
  const bool qok = li < Nout, kok = li < K;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  auto load = [&](int g, float (&x)[SW2_TP], float (&d)[SW2_TP]) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < SW2_TP; ++u) {
      const int mc = min(g + 2 * u + lh, m1 - 1);
      x[u] = X[(long long)mc * ldx];
      d[u] = dY[(long long)mc * lddy];
    }
  };
  auto use = [&](int g, const float (&x)[SW2_TP], const float (&d)[SW2_TP]) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < SW2_TP; ++u) {
      const bool in = g + 2 * u + lh < m1;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32((qok && in) ? d[u] : 0.f, (kok && in) ? x[u] : 0.f, acc, 0, 0, 0);
    }
  };
  if (m0 < m1) {
    float xa[SW2_TP], xb[SW2_TP], da[SW2_TP], db[SW2_TP];
    load(m0, xa, da);
    asm volatile("" ::: "memory");
#pragma unroll 1
    for (int g = m0; g < m1; g += 4 * SW2_TP) {
      load(g + 2 * SW2_TP, xb, db);
      asm volatile("" ::: "memory");
      use(g, xa, da);
      asm volatile("" ::: "memory");
      if (g + 2 * SW2_TP < m1) {
        load(g + 4 * SW2_TP, xa, da);
        asm volatile("" ::: "memory");
        use(g + 2 * SW2_TP, xb, db);
        asm volatile("" ::: "memory");
      }
    }
  }
  // D[q][k]: lane (k = li, lh), register r = output q = 8 (r / 4) + 4 lh + r % 4
  gf dst = (gf)(P.dW + (long long)split * P.split_stride);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int q = 8 * (r >> 2) + 4 * lh + (r & 3);
    if (q < Nout && kok) dst[(long long)q * P.sq + (long long)li * P.sk] = acc[r];
  }
}

// Column sums (dY == null: the bias gradients - sums of the per-tile partial rows the dgrad launches left, or of d z / d logits
// over all rows) as problems of the same launch: they were a launch of their own behind this one (k_skinny_wgrad, 13 us at
// config 2 for 4.6 MB).  256-wide X: a wave = 64 columns of one slab, sixteen rows in flight; K <= 16: a wave = one slab,
// lanes take different rows, one butterfly at the end.
__device__ __forceinline__ void stream_colsum_wide(const SkinnyWgradProblem &P, int grp, int wave) {
  typedef const __attribute__((address_space(1))) float *gcf;
  typedef __attribute__((address_space(1))) float *gf;
  // G = P.lddy row groups (stream_wgrad_finalize: about 64 rows each, at most one per slab - this launch's register count
  // allows eight waves per CU, so a wave per slab and 64 columns of a 25-row table would spend a dispatch round on nothing):
  // group g's sum goes to slab g, slabs g + G, g + 2 G, ... get its zeros
  const int lane = threadIdx.x, M = P.M, G = P.lddy;
  const int per = (M + G - 1) / G;
  const int m0 = grp * per, m1 = min(M, m0 + per);
  gcf X = (gcf)P.X + 64 * wave + lane;
  float acc = 0.f;
  for (int g = m0; g < m1; g += 16) {
    float x[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) x[u] = X[(long long)min(g + u, m1 - 1) * 256];
#pragma unroll
    for (int u = 0; u < 16; ++u) acc += g + u < m1 ? x[u] : 0.f;
  }
  for (int sl = grp; sl < P.nsplit; sl += G) {
    ((gf)(P.dW + (long long)sl * P.split_stride))[(long long)(64 * wave + lane) * P.sk] = acc;
    acc = 0.f;
  }
}
template <int KMAX, int U = 4>   // U x 64 rows in flight
__device__ __forceinline__ void stream_colsum_narrow(const SkinnyWgradProblem &P, int split) {
  typedef const __attribute__((address_space(1))) float *gcf;
  typedef __attribute__((address_space(1))) float *gf;
  const int lane = threadIdx.x, M = P.M, K = P.K, ldx = P.ldx;
  const int per = (M + P.nsplit - 1) / P.nsplit;
  const int m0 = split * per, m1 = min(M, m0 + per);
  gcf X = (gcf)P.X;
  float acc[KMAX];
#pragma unroll
  for (int c = 0; c < KMAX; ++c) acc[c] = 0.f;
  for (int g = m0; g < m1; g += U * 64) {
    float x[U][KMAX];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      gcf row = X + (long long)min(g + 64 * u + lane, m1 - 1) * ldx;
#pragma unroll
      for (int c = 0; c < KMAX; ++c) x[u][c] = row[min(c, K - 1)];
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int c = 0; c < KMAX; ++c) acc[c] += g + 64 * u + lane < m1 ? x[u][c] : 0.f;
  }
  gf dst = (gf)(P.dW + (long long)split * P.split_stride);
#pragma unroll
  for (int c = 0; c < KMAX; ++c) {
    float v = acc[c];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == c && c < K) dst[(long long)c * P.sk] = v;
  }
}

// MAXNG: the widest problem's groups of four outputs (1, 3, 5, 8 or 9); TINY: the launch has both-ways-narrow problems.  The
// register count of a launch is that of its widest path - 224 with everything in, two waves per SIMD; a launch whose problems
// are all narrow (config 2: two or six outputs, at most twelve) gets the instantiation without the wide bodies: more single-wave
// workgroups in flight for an HBM-latency-bound kernel.
template <int MAXNG, bool TINY>
__global__ __launch_bounds__(64) void k_stream_wgrad(const SkinnyWgradProblem *__restrict__ probs, int nprob) {
  const int bid = blockIdx.x;
  const int pi = find_problem<SkinnyWgradProblem, &SkinnyWgradProblem::block_start>(probs, nprob, bid, threadIdx.x & 63);
  const SkinnyWgradProblem &P = probs[pi];
  if (!P.dY) {   // (workgroup-uniform) column sums
    if (P.K <= 2) stream_colsum_narrow<2>(P, bid - P.block_start);
    else if (P.K <= 8) stream_colsum_narrow<8>(P, bid - P.block_start);
    else if (P.K <= 16) stream_colsum_narrow<16>(P, bid - P.block_start);
    else if (P.K <= STREAM_COLSUM_NARROW_MAX) stream_colsum_narrow<STREAM_COLSUM_NARROW_MAX, 1>(P, bid - P.block_start);   // (25 atoms, 34 logits)
    else stream_colsum_wide(P, (bid - P.block_start) >> 2, (bid - P.block_start) & 3);
    return;
  }
  if (P.K <= 32) {   // (workgroup-uniform)
    if constexpr (TINY) stream_wgrad_tiny(P, bid - P.block_start);
    return;
  }
  const int split = (bid - P.block_start) >> 2, wave = (bid - P.block_start) & 3;   // a workgroup is ONE wave: 64 columns of one slab
  const int ng = (P.Nout + 3) >> 2;   // workgroup-uniform: one of the loop bodies
  if (ng <= 1) stream_wgrad_body<1>(P, split, wave);
  else if (MAXNG >= 3 && ng <= 3) stream_wgrad_body<3>(P, split, wave);
  else if (MAXNG >= 5 && ng <= 5) stream_wgrad_body<5>(P, split, wave);
  else if (MAXNG >= 8 && ng <= 8) stream_wgrad_body<8>(P, split, wave);
  else if (MAXNG >= 9) stream_wgrad_body<9>(P, split, wave);   // (33..36 outputs: the 2 x 17 logits of config 4's actor head)
}

int stream_wgrad_finalize(SkinnyWgradProblem *p, int n) {
  int total = 0;
  for (int i = 0; i < n; ++i) {
    p[i].col_blocks = (p[i].K <= 32 || (!p[i].dY && p[i].K <= STREAM_COLSUM_NARROW_MAX)) ? 1 : 4;
    p[i].block_start = total;
    if (!p[i].dY && p[i].K > STREAM_COLSUM_NARROW_MAX) {   // column sums of a 256-wide table: row groups instead of slabs (stream_colsum_wide)
      const int g = std::max(1, std::min(p[i].nsplit, (p[i].M + 63) / 64));
      p[i].lddy = g;
      total += 4 * g;
      continue;
    }
    total += p[i].col_blocks * p[i].nsplit;
  }
  return total;
}

bool stream_wgrad_takes(const SkinnyWgradProblem &p) {
  if (!p.X || p.M < 1 || p.nsplit < 1) return false;
  if (!p.dY) return p.Nout == 1 && ((p.K == 256 && p.ldx == 256) || (p.K >= 1 && p.K <= STREAM_COLSUM_NARROW_MAX));   // column sums
  if (p.Nout < 1 || p.Nout > STREAM_WGRAD_MAX_OUT) return false;
  return (p.K == 256 && p.ldx == 256) || (p.K >= 1 && p.K <= 32 && p.Nout <= 32);   // 256-wide X, or narrow both ways
}

hipError_t stream_wgrad_launch(const SkinnyWgradProblem *host, const SkinnyWgradProblem *dev, int n, int total_blocks, hipStream_t s) {
  if (total_blocks <= 0) return hipSuccess;
  int ng = 1;
  bool tiny = false;
  for (int i = 0; i < n; ++i) {
    if (!host[i].dY) continue;
    if (host[i].K <= 32) tiny = true;
    else ng = std::max(ng, (host[i].Nout + 3) >> 2);
  }
  const dim3 g(total_blocks), b(64);
  if (ng <= 3 && !tiny) hipLaunchKernelGGL((k_stream_wgrad<3, false>), g, b, 0, s, dev, n);
  else if (ng <= 5) hipLaunchKernelGGL((k_stream_wgrad<5, true>), g, b, 0, s, dev, n);
  else hipLaunchKernelGGL((k_stream_wgrad<9, true>), g, b, 0, s, dev, n);
  return hipGetLastError();
}

// ======================================================================================
// d(pre-activation) of the last hidden layer under a narrow head (rank-Q outer product)
//   block = 64 rows x 256 columns of one problem; thread = one column: its Q head weights sit in
//   registers, the 64 x Q dY block in LDS, h is read and dpre written row by row (1 KB coalesced).
// ======================================================================================
int head_dgrad_finalize(HeadDgradProblem *p, int n) {
  int total = 0;
  for (int i = 0; i < n; ++i) {
    p[i].block_start = total;
    p[i].col_blocks = (p[i].N + 255) / 256;
    total += ((p[i].M + 63) / 64) * p[i].col_blocks;
  }
  return total;
}

__global__ __launch_bounds__(256) void k_head_dgrad(const HeadDgradProblem *__restrict__ probs, int nprob) {
  constexpr int MAXQ = HEAD_DGRAD_MAXQ;
  __shared__ float dys[64 * MAXQ];
  const int pi = find_problem<HeadDgradProblem, &HeadDgradProblem::block_start>(probs, nprob, (int)blockIdx.x, threadIdx.x & 63);
  const HeadDgradProblem P = probs[pi];
  const int local = blockIdx.x - P.block_start;
  const int rb = local / P.col_blocks, cb = local - rb * P.col_blocks;
  const int r0 = rb * 64, n = cb * 256 + threadIdx.x;
  const int Q = P.Q;
  for (int e = threadIdx.x; e < 64 * Q; e += 256) {
    const int r = e / Q, q = e - r * Q;
    dys[e] = (r0 + r < P.M) ? P.dY[(long long)(r0 + r) * P.lddy + q] : 0.f;
  }
  float w[MAXQ];
#pragma unroll
  for (int q = 0; q < MAXQ; ++q) w[q] = (q < Q && n < P.N) ? P.Wh[(long long)q * P.ldw + n] : 0.f;
  __syncthreads();
  if (n >= P.N) return;
  const int nr = min(64, P.M - r0);
  const float *h = P.h + (long long)r0 * P.N + n;
  float *out = P.dpre + (long long)r0 * P.N + n;
  float csum = 0.f;
  // 32 rows per round, their activations requested together (a loop of single rows is a memory round trip per row: at 256
  // rows - temporal_len 2 - the kernel was 64 dependent round trips long)
  for (int rb0 = 0; rb0 < nr; rb0 += 32) {
    float hv[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) hv[u] = h[(long long)min(rb0 + u, nr - 1) * P.N];
    asm volatile("" ::: "memory");
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      const int r = rb0 + u;
      if (r < nr) {
        float g = 0.f;
#pragma unroll
        for (int q = 0; q < MAXQ; ++q)
          if (q < Q) g = fmaf(dys[r * Q + q], w[q], g);
        const float x = hv[u] > 0.f ? g : 0.01f * g;
        out[(long long)r * P.N] = x;
        csum += x;
      }
    }
  }
  if (P.colsum) P.colsum[(long long)rb * P.N + n] = csum;
}

hipError_t head_dgrad_launch(const HeadDgradProblem *dev, int n, int total_blocks, hipStream_t s) {
  if (total_blocks <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_head_dgrad, dim3(total_blocks), dim3(256), 0, s, dev, n);
  return hipGetLastError();
}

// ======================================================================================
// Philox4x32-10 (counter-based RNG for perf runs; parity runs pass noise in)
// ======================================================================================
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ float u32_to_unit_open(uint32_t x) {  // (0, 1)
  return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f);
}

// one N(0,1) (continuous) or U(0,1) (discrete) draw for element e of stream `which`
__device__ __forceinline__ float device_noise(uint64_t seed, uint32_t step, uint32_t which, uint32_t e, bool normal) {
  uint32_t r[4];
  philox4x32_10(e >> 1, step, which, 0x5eedu, (uint32_t)seed, (uint32_t)(seed >> 32), r);
  if (!normal) return u32_to_unit_open(r[(e & 1) * 2]);
  const float u1 = u32_to_unit_open(r[0]), u2 = u32_to_unit_open(r[1]);
  const float rad = sqrtf(-2.0f * logf(u1));
  float sn, cs;
  sincosf(6.28318530717958647692f * u2, &sn, &cs);
  return (e & 1) ? rad * sn : rad * cs;
}

// ======================================================================================
// prep: mask / is_contiguous / per-window normaliser -> per-row loss weight
//   deepQlearning.py:201-203, 222-225, 249
// ======================================================================================
// block = 16 windows x 16 time lanes; consecutive threads -> consecutive b (coalesced rows)
// (workgroup `blk` of (B + 15) / 16, 256 threads; also runs as extra workgroups of the policy-forward launch: PrepArgs)
__device__ __forceinline__ void prep_block(int blk, const float *__restrict__ task_done, const float *__restrict__ episode_step,
                                           int T, int B, int burn_in, int cumprod, float inv_global_batch,
                                           float *__restrict__ w, float *__restrict__ contig, DevState *st,
                                           const float *log_alpha) {
  // get_losses() side effect in the reference: curr_alpha <- exp(log_alpha) AFTER it was used (soft_actor_critic.py:152);
  // rides in this launch (nothing here reads alpha) instead of a one-thread kernel of its own
  if (st && blk == 0 && threadIdx.x == 0) {
    st->alpha_cur = st->alpha_next;
    st->alpha_next = expf(*log_alpha);
  }
  __shared__ float cnt[16][17];
  const int bl = threadIdx.x & 15, tl = threadIdx.x >> 4;
  const int b = blk * 16 + bl;
  float c_local = 0.f;
  if (b < B) {
    if (cumprod) {
      // GRU joiner (encoder.py:80): is_contiguous <- cumprod over t, THEN the burn-in rows are zeroed
      // (deepQlearning.py:219-220): a sequential walk per window, done by its first time lane
      if (tl == 0) {
        float run = 1.f;
        for (int t = 0; t < T - 1; ++t) {
          const int m = t * B + b;
          const bool c = (episode_step[m + B] == episode_step[m] + 1.0f) && (task_done[m] == 0.f);
          run = c ? run : 0.f;
          const float v = t >= burn_in ? run : 0.f;
          contig[m] = v;
          c_local += v;
        }
      }
    } else {
      for (int t = tl; t < T - 1; t += 16) {
        const int m = t * B + b;
        // deepQlearning.py:219-220: the first burn_in rows carry no loss
        const bool c = t >= burn_in && (episode_step[m + B] == episode_step[m] + 1.0f) && (task_done[m] == 0.f);
        contig[m] = c ? 1.f : 0.f;
        c_local += c ? 1.f : 0.f;
      }
    }
  }
  cnt[tl][bl] = c_local;
  __syncthreads();   // also orders lane 0's contig[] stores before the other lanes' reads below (same block)
  float total = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) total += cnt[j][bl];   // exact: a count of at most T-1 ones
  const float denom = total + 1e-4f;
  if (b < B) {
    for (int t = tl; t < T - 1; t += 16) {
      const int m = t * B + b;
      w[m] = ((contig[m] / denom) * inv_global_batch) / (float)T;
    }
  }
}

__global__ __launch_bounds__(256) void k_prep(const float *__restrict__ task_done, const float *__restrict__ episode_step,
                                              int T, int B, int burn_in, int cumprod, float inv_global_batch,
                                              float *__restrict__ w, float *__restrict__ contig, DevState *st,
                                              const float *log_alpha) {
  prep_block((int)blockIdx.x, task_done, episode_step, T, B, burn_in, cumprod, inv_global_batch, w, contig, st, log_alpha);
}

// ======================================================================================
// tick: optimiser step counter, Adam bias corrections (double, like the Python floats in
// torch.optim.Adam), and the one-step-lagged alpha (soft_actor_critic.py:41,152)
// ======================================================================================
// Runs on one thread of k_loss_finish (the step's one single-block kernel): corrections of the step ABOUT to be
// applied, step + 1; the counter itself is advanced by k_adam_polyak.
__device__ void tick_adam(DevState *st, double lr, double b1, double b2) {
  const double step = (double)(st->step + 1);
  const double bc1 = 1.0 - pow(b1, step);
  const double bc2 = 1.0 - pow(b2, step);
  st->neg_step_size = (float)(-(lr / bc1));
  st->bc2_sqrt = (float)sqrt(bc2);
}

// ======================================================================================
// tanh-Gaussian policy head: sample, log-prob (gaussian_mlp.py:15-39) and its backward
// ======================================================================================
// One thread per (row, action): the per-element work is four double-precision transcendentals (~600 instructions) - with a
// thread per ROW (six elements in sequence, 392 waves for 12 544 rows x 2 passes) the kernel was one long dependent
// instruction stream per SIMD: 12 us.  AG = lanes per row (power of two >= A); the row's log-prob is summed by the row's first
// lane in the order j = 0, 1, ... (the order of the sequential loop this replaces).
// pr.nblocks > 0: the launch carries that many extra 256-thread workgroups behind its own, which do k_prep's work (nothing
// here reads what prep writes: row weights, is_contiguous, the lagged alpha) - one launch less in a latency-bound step.
template <int AG>
__global__ __launch_bounds__(256) void k_policy_fwd(PolicyFwdArgs a0, PolicyFwdArgs a1, int nprob, int M, int A, const DevState *st,
                                                    uint64_t seed, PrepArgs pr, int own_blocks) {
#pragma clang fp contract(off)
  if ((int)blockIdx.x >= own_blocks) {   // uniform
    prep_block((int)blockIdx.x - own_blocks, pr.task_done, pr.episode_step, pr.T, pr.B, pr.burn_in, pr.cumprod, pr.inv_global_batch,
               pr.w, pr.contig, pr.st, pr.log_alpha);
    return;
  }
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int rowid = gid / AG, j = gid - rowid * AG;
  const int p = rowid / M;
  const bool live = p < nprob && j < A;
  const int m = rowid - p * M;
  const PolicyFwdArgs &a = p == 0 ? a0 : a1;
  float lp = 0.f;
  if (live) {
    const float *lo = a.logits + (long long)m * 2 * A;
    const float mean = lo[j];
    const float ls = fminf(fmaxf(lo[A + j], -20.f), 2.f);
    // transcendental functions through double, rounded once to f32: the log-prob below is
    // ill-conditioned near |a| -> 1 (d logp = 2a/(1-a^2+1e-4) da), so every ulp of exp/tanh/log
    // matters when comparing with the CPU path's (<= 1 ulp) vector math library.
    const float sd = (float)dm_exp((double)ls);
    const float eps = a.noise ? a.noise[(long long)m * A + j]
                              : device_noise(seed, (uint32_t)st->step, a.which, (uint32_t)(m * A + j), true);
    if (a.noise_out) a.noise_out[(long long)m * A + j] = eps;
    const float x = mean + eps * sd;
    const float d = x - mean;
    lp = -(d * d) / (2.f * (sd * sd)) - (float)dm_log((double)sd) - 0.91893853320467274178f;
    const float act = (float)dm_tanh((double)x);
    lp -= (float)dm_log((double)((1.f - act * act) + 1e-4f));
    a.action[(long long)m * A + j] = act;
    if (a.diff) a.diff[(long long)m * A + j] = act - a.sub[(long long)m * A + j];
  }
  // sum over the row's lanes, in order (every lane of the wave takes part in the shuffles)
  const int base = (threadIdx.x & 63) - j;
  float logp = 0.f;
  for (int k = 0; k < A; ++k) logp += __shfl(lp, base + k, 64);
  if (live && j == 0) a.logp[m] = logp;
}

// d logits from d pi (through the frozen critics) and d logp = w*alpha.  One thread per (row, action).
// fin.nblocks > 0: one extra 256-thread workgroup behind the launch's own does k_loss_finish's work (sums the loss partials,
// publishes the scalars and d log_alpha, the Adam bias corrections): nothing before the optimiser reads those - one launch less.
__device__ __forceinline__ void loss_finish_block(const float *__restrict__ partials, int nblocks, int M, int Nq, DevState *st,
                                                  float *__restrict__ scalars, float *__restrict__ dlog_alpha, double lr, double b1, double b2);
struct LossFinishRider {
  const float *partials;
  LossFinishArgs f;
};
__global__ __launch_bounds__(256) void k_policy_bwd(const float *__restrict__ logits, const float *__restrict__ noise,
                                                    const float *__restrict__ action, const float *__restrict__ dpi_parts, int nparts,
                                                    float *__restrict__ dpi_sum, const float *__restrict__ w, const DevState *st, int M, int A,
                                                    float *__restrict__ dlogits, LossFinishRider fin, int own_blocks) {
  if ((int)blockIdx.x >= own_blocks) {   // (uniform)
    loss_finish_block(fin.partials, fin.f.nblocks, fin.f.M, fin.f.Nq, fin.f.st, fin.f.scalars, fin.f.dlog_alpha, fin.f.lr, fin.f.b1, fin.f.b2);
    return;
  }
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int m = (int)(gid / A), j = (int)(gid - (long long)m * A);
  if (m >= M) return;
  const float glp = w[m] * st->alpha_cur;
  const float *lo = logits + (long long)m * 2 * A;
  const float lsr = lo[A + j];
  const float ls = fminf(fmaxf(lsr, -20.f), 2.f);
  const float sd = (float)dm_exp((double)ls);
  const float eps = noise[(long long)m * A + j];
  const float act = action[(long long)m * A + j];
  const float om = 1.f - act * act;
  float g = 0.f, gp[8];   // d loss / d pi_j = sum over the frozen critics' partials, fixed order (all requested before the first add)
#pragma unroll
  for (int c = 0; c < 8; ++c) gp[c] = dpi_parts[((long long)min(c, nparts - 1) * M + m) * A + j];
#pragma unroll
  for (int c = 0; c < 8; ++c) g += c < nparts ? gp[c] : 0.f;
  for (int c = 8; c < nparts; ++c) g += dpi_parts[((long long)c * M + m) * A + j];
  dpi_sum[(long long)m * A + j] = g;
  const float dx = g * om + glp * (2.f * act * om / (om + 1e-4f));
  const float dsd = dx * eps - glp / sd;
  float dls = dsd * sd;
  if (lsr < -20.f || lsr > 2.f) dls = 0.f;
  dlogits[(long long)m * 2 * A + j] = dx;
  dlogits[(long long)m * 2 * A + A + j] = dls;
}

// The policy backward of 64 rows and, in the same launch, the pre-activation gradient of the actor's LAST hidden layer under
// its narrow head (franQ: autograd through mlp.py:88-94 / soft_actor_critic.py:136-154):
//   d logits[m, 0..2A)  as k_policy_bwd;   dpre[m, n] = LeakyReLU'(h[m, n]) * sum_k d logits[m, k] Wh[k, n]   (n < 256, fma chain over k)
// and the per-64-row column sums of dpre (the bias gradient's partials, the tile kernel's layout).  The K = 2A product was a
// GEMM launch of its own (12 us at config 2 on tiles that are 5 % MFMA work) behind a 6 us launch that produced its operand:
// here the 64 rows' d logits stay in LDS, a lane owns four columns (Wh's 2A x 4 values in registers), a wave four rows.
// 1024 threads per 64 rows: phase 1 has 64 A (row, action) items, phase 2 streams 64 KiB in and out per workgroup.
constexpr int PBD_ROWS = 64, PBD_N = 256, PBD_MAXK = 16;
struct PolicyBwdDpreArgs {
  const float *Wh;      // head weights of the hidden layer's columns: element (k, n) at Wh[k * ldw + n]
  int ldw;
  const float *h;       // [M, 256] the layer's activations (gate reference)
  float *dpre;          // [M, 256]
  float *colsum;        // [ceil(M / 64), 256]
};
__global__ __launch_bounds__(1024) void k_policy_bwd_dpre(const float *__restrict__ logits, const float *__restrict__ noise,
                                                          const float *__restrict__ action, const float *__restrict__ dpi_parts, int nparts,
                                                          float *__restrict__ dpi_sum, const float *__restrict__ w, const DevState *st, int M, int A,
                                                          float *__restrict__ dlogits, PolicyBwdDpreArgs d, LossFinishRider fin, int own_blocks) {
  if ((int)blockIdx.x >= own_blocks) {   // (uniform; every thread of the workgroup enters: the finish has workgroup barriers)
    loss_finish_block(fin.partials, fin.f.nblocks, fin.f.M, fin.f.Nq, fin.f.st, fin.f.scalars, fin.f.dlog_alpha, fin.f.lr, fin.f.b1, fin.f.b2);
    return;
  }
  typedef float pv4 __attribute__((ext_vector_type(4)));
  __shared__ __attribute__((aligned(16))) float dl[PBD_ROWS][PBD_MAXK];
  __shared__ __attribute__((aligned(16))) float cs[16][PBD_N];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r0 = blockIdx.x * PBD_ROWS, K = 2 * A;
  // the gate references of this wave's rows and the head's K x 256 weights (-> LDS, rows beyond K zero): requested before
  // phase 1's dependent loads
  __shared__ __attribute__((aligned(16))) float wl[PBD_MAXK][PBD_N];
  pv4 hq[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int m = min(r0 + 4 * wave + u, M - 1);
    hq[u] = *reinterpret_cast<const pv4 *>(d.h + (long long)m * PBD_N + lane * 4);
  }
  {
    float wv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = 4 * i + (tid >> 8), n = tid & 255;   // (k, n) of element tid + 1024 i
      wv[i] = d.Wh[(long long)min(k, K - 1) * d.ldw + n];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = 4 * i + (tid >> 8);
      wl[k][tid & 255] = k < K ? wv[i] : 0.f;
    }
  }
  // ---- phase 1: d logits of the 64 rows (k_policy_bwd's arithmetic), kept in LDS
  if (tid < PBD_ROWS * A) {
    const int rl = tid / A, j = tid - rl * A, m = r0 + rl;
    float dx = 0.f, dls = 0.f;
    if (m < M) {
      const float glp = w[m] * st->alpha_cur;
      const float *lo = logits + (long long)m * 2 * A;
      const float lsr = lo[A + j];
      const float ls = fminf(fmaxf(lsr, -20.f), 2.f);
      const float sd = (float)dm_exp((double)ls);
      const float eps = noise[(long long)m * A + j];
      const float act = action[(long long)m * A + j];
      const float om = 1.f - act * act;
      float g = 0.f, gp[8];   // (the frozen critics' partials, fixed order; all requested before the first add)
#pragma unroll
      for (int c = 0; c < 8; ++c) gp[c] = dpi_parts[((long long)min(c, nparts - 1) * M + m) * A + j];
#pragma unroll
      for (int c = 0; c < 8; ++c) g += c < nparts ? gp[c] : 0.f;
      for (int c = 8; c < nparts; ++c) g += dpi_parts[((long long)c * M + m) * A + j];
      dpi_sum[(long long)m * A + j] = g;
      dx = g * om + glp * (2.f * act * om / (om + 1e-4f));
      const float dsd = dx * eps - glp / sd;
      dls = dsd * sd;
      if (lsr < -20.f || lsr > 2.f) dls = 0.f;
      dlogits[(long long)m * 2 * A + j] = dx;
      dlogits[(long long)m * 2 * A + A + j] = dls;
    }
    dl[rl][j] = dx;
    dl[rl][A + j] = dls;
  }
  for (int e = tid; e < PBD_ROWS * (PBD_MAXK - K); e += 1024) dl[e / (PBD_MAXK - K)][K + e % (PBD_MAXK - K)] = 0.f;
  __syncthreads();
  // ---- phase 2: this wave's four rows
  pv4 wk[PBD_MAXK];
#pragma unroll
  for (int k = 0; k < PBD_MAXK; ++k) wk[k] = *reinterpret_cast<const pv4 *>(&wl[k][lane * 4]);
  pv4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int rl = 4 * wave + u, m = r0 + rl;
    pv4 dq[PBD_MAXK / 4];
#pragma unroll
    for (int q = 0; q < PBD_MAXK / 4; ++q) dq[q] = *reinterpret_cast<const pv4 *>(&dl[rl][4 * q]);   // (same address in every lane: broadcast)
    pv4 x = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < PBD_MAXK; ++k)   // (k >= K: 0 * 0 added)
#pragma unroll
      for (int c = 0; c < 4; ++c) x[c] = fmaf(dq[k >> 2][k & 3], wk[k][c], x[c]);
#pragma unroll
    for (int c = 0; c < 4; ++c) x[c] = hq[u][c] > 0.f ? x[c] : 0.01f * x[c];
    if (m < M) {
      *reinterpret_cast<pv4 *>(d.dpre + (long long)m * PBD_N + lane * 4) = x;
      sum += x;
    }
  }
  *reinterpret_cast<pv4 *>(&cs[wave][lane * 4]) = sum;
  __syncthreads();
  if (tid < PBD_N) {
    float t = 0.f;
#pragma unroll
    for (int v = 0; v < 16; ++v) t += cs[v][tid];
    d.colsum[(long long)blockIdx.x * PBD_N + tid] = t;
  }
}

// k_head_dgrad_masked (update_kernels.h, HeadDgradMaskedArgs): a 512-thread workgroup = 4 x 64 rows of one instance; the head's
// Q x 256 weights through LDS into registers (a lane owns four columns), the 64 rows' dY in LDS (read as broadcasts), a wave
// forms eight rows.  The gates of a row's four columns are a nibble of the row's mask dword in the forward wave that owned
// the columns (the layout k_wstat_grad's fused loader reads, wstat.hip).
#ifndef FDQL_HDM_CHUNKS
#define FDQL_HDM_CHUNKS 4
#endif
constexpr int HDM_CHUNKS = FDQL_HDM_CHUNKS;   // 64-row chunks per workgroup: the head's weights are staged once per 256 rows
__global__ __launch_bounds__(512) void k_head_dgrad_masked(const HeadDgradMaskedArgs a) {
  typedef float hv4 __attribute__((ext_vector_type(4)));
  __shared__ __attribute__((aligned(16))) float wl[HDM_MAXQ][256];
  __shared__ __attribute__((aligned(16))) float dl[2][64][HDM_MAXQ];
  __shared__ __attribute__((aligned(16))) float cs[8][256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int inst = blockIdx.y, M = a.M, Q = a.Q;
  const float *Wh = a.Wh[inst], *dY = a.dY[inst];
  const unsigned *gm = a.gm[inst];
  float *dpre = a.dpre[inst];
  const unsigned mlane = (unsigned)((lane >> 4) * 64 + 32 * (lane & 1)), lsh = 4u * ((unsigned)(lane & 15) >> 1);
  // (every staging round: all requests, then all LDS writes - a load -> store loop is one memory round trip per iteration)
  auto stage_dy = [&](int buf, int r0) __attribute__((always_inline)) {
    float v[64 * HDM_MAXQ / 512];
#pragma unroll
    for (int t = 0; t < 64 * HDM_MAXQ / 512; ++t) {
      const int e = tid + 512 * t, r = e / HDM_MAXQ, q = e - r * HDM_MAXQ;
      v[t] = dY[(long long)min(r0 + r, M - 1) * a.lddy + min(q, Q - 1)];
    }
#pragma unroll
    for (int t = 0; t < 64 * HDM_MAXQ / 512; ++t) {
      const int e = tid + 512 * t, r = e / HDM_MAXQ, q = e - r * HDM_MAXQ;
      dl[buf][r][q] = (q < Q && r0 + r < M) ? v[t] : 0.f;
    }
  };
  {
    float v[HDM_MAXQ * 256 / 512];
#pragma unroll
    for (int t = 0; t < HDM_MAXQ * 256 / 512; ++t) {
      const int e = tid + 512 * t, q = e >> 8, n = e & 255;
      v[t] = Wh[(long long)min(q, Q - 1) * a.ldw + n];
    }
#pragma unroll
    for (int t = 0; t < HDM_MAXQ * 256 / 512; ++t) {
      const int e = tid + 512 * t, q = e >> 8, n = e & 255;
      wl[q][n] = q < Q ? v[t] : 0.f;
    }
  }
  const int rb = blockIdx.x * 64 * HDM_CHUNKS;
  stage_dy(0, rb);
  __syncthreads();
#pragma unroll 1
  for (int ch = 0; ch < HDM_CHUNKS; ++ch) {
    const int r0 = rb + 64 * ch, buf = ch & 1;
    if (r0 >= M) break;   // (uniform)
    // the masks of this wave's eight rows (the only dependent-address loads) and the next chunk's dY: requested first
    unsigned mk[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int m = min(r0 + 8 * wave + u, M - 1);
      mk[u] = gm[(long long)(m >> 5) * 256 + mlane + (m & 31)];
    }
    if (ch + 1 < HDM_CHUNKS) stage_dy(buf ^ 1, r0 + 64);
    // x[u] as two packed pairs: v_pk_fma_f32 forms two columns per instruction (the plain fp32 VALU rate bounds this kernel)
    typedef float hv2 __attribute__((ext_vector_type(2)));
    hv2 xa[8], xb[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { xa[u] = hv2{0.f, 0.f}; xb[u] = hv2{0.f, 0.f}; }
#pragma unroll
    for (int q0 = 0; q0 < HDM_MAXQ; q0 += 4) {   // four head rows' weights in registers at a time; whole groups of 4 (q >= Q: 0 * 0 added)
      if (q0 < Q) {   // (uniform)
        hv4 wk[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) wk[k] = *reinterpret_cast<const hv4 *>(&wl[q0 + k][lane * 4]);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const hv4 d0 = *reinterpret_cast<const hv4 *>(&dl[buf][8 * wave + u][q0]);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const hv2 dv = {d0[k], d0[k]};
            xa[u] = __builtin_elementwise_fma(dv, hv2{wk[k][0], wk[k][1]}, xa[u]);
            xb[u] = __builtin_elementwise_fma(dv, hv2{wk[k][2], wk[k][3]}, xb[u]);
          }
        }
      }
    }
    hv4 x[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) x[u] = hv4{xa[u][0], xa[u][1], xb[u][0], xb[u][1]};
    hv4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int m = r0 + 8 * wave + u;
      unsigned mm = mk[u] << lsh;
      hv4 y;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        y[c] = (mm & 0x80000000u) ? x[u][c] : 0.01f * x[u][c];
        mm <<= 1;
      }
      if (m < M) {
        *reinterpret_cast<hv4 *>(dpre + (long long)m * 256 + lane * 4) = y;
        sum += y;
      }
    }
    *reinterpret_cast<hv4 *>(&cs[wave][lane * 4]) = sum;
    __syncthreads();   // (also: the next chunk's dY has landed, everybody is done with this chunk's)
    if (tid < 256) {
      float t = 0.f;
#pragma unroll
      for (int v = 0; v < 8; ++v) t += cs[v][tid];
      a.colsum[inst][(long long)(r0 >> 6) * 256 + tid] = t;
    }
    __syncthreads();   // (cs is rewritten by the next chunk)
  }
}
hipError_t head_dgrad_masked_launch(const HeadDgradMaskedArgs &a, hipStream_t s) {
  if (a.M <= 0 || a.ninst <= 0) return hipSuccess;
  if (a.ninst > HDM_MAX_INST || a.Q < 1 || a.Q > HDM_MAXQ || (a.M & 31)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_head_dgrad_masked, dim3((unsigned)((a.M + 64 * HDM_CHUNKS - 1) / (64 * HDM_CHUNKS)), (unsigned)a.ninst), dim3(512), 0, s, a);
  return hipGetLastError();
}

// ======================================================================================
// Discrete actor: Gumbel-softmax straight-through sample and log-prob
// (franQ/Agent/models/gumbel_mlp.py:7-54 on torch's ExpRelaxedCategorical.rsample, temperature 1)
// and its backward.  One thread per row, n <= 32 actions.
// ======================================================================================
constexpr int GUMBEL_MAXN = 32;

// A thread's per-action values live in LDS (element j of lane l at p[j * stride], stride = the workgroup's thread count:
// lane-contiguous, conflict-free): as runtime-indexed local arrays they were 272 / 528 bytes of scratch per thread.
struct LaneArr {
  float *p;
  int stride;
  __device__ __forceinline__ float &operator[](int j) const { return p[j * stride]; }
};

__device__ __forceinline__ float row_logsumexp(const LaneArr &x, int n) {
  float mx = -INFINITY;
  for (int j = 0; j < n; ++j) mx = fmaxf(mx, x[j]);
  float s = 0.f;
  for (int j = 0; j < n; ++j) s += (float)dm_exp((double)(x[j] - mx));
  return mx + (float)dm_log((double)s);
}

// forward pieces shared by fwd and bwd: norm = logits - lse, relaxed = softmax(norm + gumbel), hard index
__device__ __forceinline__ int gumbel_forward(const LaneArr &lo, const LaneArr &u, int n, const LaneArr &norm, const LaneArr &relaxed,
                                              const LaneArr &sc) {
  const float tiny = 1.1920928955078125e-07f;  // torch.finfo(float32).eps (clamp_probs)
  const float lse = row_logsumexp(lo, n);
  for (int j = 0; j < n; ++j) {
    norm[j] = lo[j] - lse;
    const float uc = fminf(fmaxf(u[j], tiny), 1.f - tiny);
    const float g = -(float)dm_log((double)(-(float)dm_log((double)uc)));
    sc[j] = (norm[j] + g) / 1.0f;
  }
  const float lse2 = row_logsumexp(sc, n);
  int best = 0;
  float best_v = -INFINITY;
  for (int j = 0; j < n; ++j) {
    const float r = (float)dm_exp((double)(sc[j] - lse2));
    relaxed[j] = r;
    if (j == 0 || r > best_v) { best_v = r; best = j; }   // first maximum, like torch.argmax
  }
  return best;
}

constexpr int GUMBEL_THREADS = 64;

__global__ __launch_bounds__(GUMBEL_THREADS) void k_policy_fwd_gumbel(PolicyFwdArgs a0, PolicyFwdArgs a1, int nprob, int M, int n, const DevState *st,
                                                                     uint64_t seed) {
#pragma clang fp contract(off)
  __shared__ float lane_vals[5 * GUMBEL_MAXN * GUMBEL_THREADS];
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int p = gid / M;
  if (p >= nprob) return;
  const int m = gid - p * M;
  const PolicyFwdArgs &a = p == 0 ? a0 : a1;
  auto arr = [&](int k) { return LaneArr{lane_vals + k * GUMBEL_MAXN * GUMBEL_THREADS + threadIdx.x, GUMBEL_THREADS}; };
  const LaneArr lo = arr(0), u = arr(1), norm = arr(2), relaxed = arr(3), sc = arr(4);
  for (int j = 0; j < n; ++j) {
    lo[j] = a.logits[(long long)m * n + j];
    const float uj = a.noise ? a.noise[(long long)m * n + j] : device_noise(seed, (uint32_t)st->step, a.which, (uint32_t)(m * n + j), false);
    u[j] = uj;
    if (a.noise_out) a.noise_out[(long long)m * n + j] = uj;
  }
  const int best = gumbel_forward(lo, u, n, norm, relaxed, sc);
  const float lse3 = row_logsumexp(norm, n);   // log_softmax of the already normalised logits
  float logp = 0.f;
  for (int j = 0; j < n; ++j) {
    const float hard = j == best ? 1.f : 0.f;
    const float rj = relaxed[j];
    const float stv = (hard - rj) + rj;   // straight-through value, same fp order as torch
    a.action[(long long)m * n + j] = stv;
    if (a.diff) a.diff[(long long)m * n + j] = stv - a.sub[(long long)m * n + j];
    logp += -stv * (norm[j] - lse3);
  }
  a.logp[m] = -logp;
}

__global__ __launch_bounds__(GUMBEL_THREADS) void k_policy_bwd_gumbel(const float *__restrict__ logits, const float *__restrict__ noise,
                                                                     const float *__restrict__ dpi_parts, int nparts, float *__restrict__ dpi_sum,
                                                                     const float *__restrict__ w, const DevState *st, int M, int n,
                                                                     float *__restrict__ dlogits) {
  __shared__ float lane_vals[7 * GUMBEL_MAXN * GUMBEL_THREADS];
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  auto arr = [&](int k) { return LaneArr{lane_vals + k * GUMBEL_MAXN * GUMBEL_THREADS + threadIdx.x, GUMBEL_THREADS}; };
  const LaneArr lo = arr(0), u = arr(1), norm = arr(2), relaxed = arr(3), sc = arr(4), gst = arr(5), dnorm = arr(6);
  for (int j = 0; j < n; ++j) { lo[j] = logits[(long long)m * n + j]; u[j] = noise[(long long)m * n + j]; }
  const int best = gumbel_forward(lo, u, n, norm, relaxed, sc);
  const float lse3 = row_logsumexp(norm, n);
  const float glp = w[m] * st->alpha_cur;   // d loss / d logp
  float dot_st = 0.f, sum_glogsm = 0.f;
  for (int j = 0; j < n; ++j) {
    float g = 0.f;
    for (int c = 0; c < nparts; ++c) g += dpi_parts[((long long)c * M + m) * n + j];
    dpi_sum[(long long)m * n + j] = g;
    const float logsm = norm[j] - lse3;
    const float gj = g + glp * logsm;                 // logp = sum st * logsm
    gst[j] = gj;
    const float rj = relaxed[j];
    dot_st += gj * rj;
    const float hard = j == best ? 1.f : 0.f;
    sum_glogsm += glp * ((hard - rj) + rj);
  }
  float sum_dnorm = 0.f;
  for (int j = 0; j < n; ++j) {
    const float hard = j == best ? 1.f : 0.f;
    const float rj = relaxed[j];
    const float stv = (hard - rj) + rj;
    const float d_scores = rj * (gst[j] - dot_st);                                             // softmax backward
    const float d_logsm = glp * stv - (float)dm_exp((double)(norm[j] - lse3)) * sum_glogsm;       // log_softmax backward
    const float dn = d_scores + d_logsm;
    dnorm[j] = dn;
    sum_dnorm += dn;
  }
  for (int j = 0; j < n; ++j)
    dlogits[(long long)m * n + j] = dnorm[j] - (float)dm_exp((double)norm[j]) * sum_dnorm;        // norm = logits - lse
}

// one-hot of the stored action index (deepQlearning.py:206-210)
__global__ void k_onehot(const float *__restrict__ action, int rows, int n, float *__restrict__ out) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= rows * n) return;
  const int m = e / n, j = e - m * n;
  out[e] = ((int)action[m] == j) ? 1.f : 0.f;
}

// ======================================================================================
// Loss: TQC (sort pooled atoms, drop the top, entropy bonus, pairwise quantile-Huber with
// tau over the pooled index; distributional_soft_actor_critic.py:40-103) or SAC-min
// (soft_actor_critic.py:63-134); n-step lower bound; actor / alpha loss
// (soft_actor_critic.py:136-154); analytic d loss / d q_pred; per-block partial sums.
// G threads per row (G = pow2 >= Nq), 256/G rows per block.
// ======================================================================================
__device__ __forceinline__ float group_sum(float v, float *scratch, int g, int i, int G) {
  // scratch: [rows][G]; all threads of the block call this together
  scratch[g * G + i] = v;
  __syncthreads();
  for (int off = G >> 1; off >= 1; off >>= 1) {
    if (i < off) scratch[g * G + i] += scratch[g * G + i + off];
    __syncthreads();
  }
  const float r = scratch[g * G];
  __syncthreads();
  return r;
}

__device__ __forceinline__ void loss_finish_block(const float *__restrict__ partials, int nblocks, int M, int Nq, DevState *st,
                                                  float *__restrict__ scalars, float *__restrict__ dlog_alpha, double lr, double b1, double b2);

__global__ __launch_bounds__(256) void k_loss(LossArgs a) {
  extern __shared__ float sm[];
  const int G = a.G, RPB = 256 / G;
  float *zs = sm;                 // [RPB][G]
  float *ys = sm + RPB * G;       // [RPB][G]
  float *scratch = sm + 2 * RPB * G;  // [RPB][G]
  float *bsum = sm + 3 * RPB * G;     // [RPB][LOSS_NPART]
  const int tid = threadIdx.x, g = tid / G, i = tid - g * G;
  const int m = blockIdx.x * RPB + g;
  const bool row_ok = m < a.M;
  const int mm = row_ok ? m : a.M - 1;
  const int Nq = a.Nq, Nt = a.Nt;
  const bool act = row_ok && i < Nq;
  const float alpha = a.st->alpha_cur;
  const float log_alpha = *a.log_alpha;

  const float zi = (i < Nq) ? a.z_target[(long long)mm * Nq + i] : INFINITY;
  const float lpn = a.logp_next[mm];
  const float ent = alpha * (-lpn);
  const float rew = a.reward[mm + a.B];
  const float maskg = (a.task_done[mm + a.B] == 0.f ? 1.f : 0.f) * a.gamma;
  const float qi = (i < Nq) ? a.q_pred[(long long)mm * Nq + i] : 0.f;
  const float mc = a.mc_return ? a.mc_return[mm + a.B] : 0.f;
  const float wm = a.w[mm];

  float loss_i = 0.f, grad_i = 0.f, lb_i = 0.f;
  zs[g * G + i] = zi;
  __syncthreads();
  if (a.distributional) {
    int rank = 0;
    for (int j = 0; j < Nq; ++j) {
      const float zj = zs[g * G + j];
      rank += (zj < zi || (zj == zi && j < i)) ? 1 : 0;
    }
    if (i < Nq && rank < Nt) {
      float tq = zi;
      if (a.max_entropy) tq = tq + ent;
      const float y = rew + maskg * tq;
      ys[g * G + rank] = y;
      if (row_ok) a.td_target[(long long)m * Nt + rank] = y;
    }
    __syncthreads();
    if (i < Nq) {
      const float tau = (float)i / (float)Nq + a.half_inv_nq;
      float sl = 0.f, sg = 0.f;
      for (int j = 0; j < Nt; ++j) {
        const float d = ys[g * G + j] - qi;
        const float ad = fabsf(d);
        const float hub = ad > 1.f ? ad - 0.5f : d * d * 0.5f;
        const float wt = fabsf(tau - (d < 0.f ? 1.f : 0.f));
        sl += wt * hub;
        const float dh = ad > 1.f ? (d > 0.f ? 1.f : -1.f) : d;
        sg -= wt * dh;
      }
      const float inv = 1.f / (float)(Nq * Nt);
      loss_i = sl * inv;
      grad_i = sg * inv;
      if (a.lowerbound) {
        const float lb = mc - qi;
        if (lb > 0.f) { lb_i = lb / (float)Nq; grad_i -= 1.f / (float)Nq; }
      }
    }
  } else {
    // SAC: min over all heads of (z + alpha*H), one target per row
    float tq = zi;
    if (a.max_entropy && i < Nq) tq = zi + ent;
    ys[g * G + i] = tq;
    __syncthreads();
    float mn = INFINITY;
    for (int j = 0; j < Nq; ++j) mn = fminf(mn, ys[g * G + j]);
    const float y = rew + maskg * mn;
    if (row_ok && i == 0) a.td_target[m] = y;
    if (i < Nq) {
      const float x = qi - y;
      const float ax = fabsf(x);
      float l = ax < 1.f ? 0.5f * x * x : ax - 0.5f;
      float gq = ax < 1.f ? x : (x > 0.f ? 1.f : -1.f);
      if (a.lowerbound) {
        const float lb = fmaxf(mc - qi, 0.f);
        if (lb != 0.f) { l = lb; gq = -1.f; }
      }
      loss_i = l / (float)Nq;
      grad_i = gq / (float)Nq;
    }
  }
  if (act) {
    a.dz[(long long)m * Nq + i] = wm * grad_i;
    a.dzf[(long long)m * Nq + i] = -wm / (float)Nq;
  }
  const float zf = (i < Nq) ? a.z_frozen[(long long)mm * Nq + i] : 0.f;
  const float qloss = group_sum(loss_i + lb_i, scratch, g, i, G);
  const float zfsum = group_sum(zf, scratch, g, i, G);
  const float qsum = group_sum(i < Nq ? qi : 0.f, scratch, g, i, G);
  const float viol = group_sum((a.lowerbound && i < Nq && (mc - qi) > 0.f) ? 1.f : 0.f, scratch, g, i, G);

  if (i == 0) {
    float part[LOSS_NPART];
#pragma unroll
    for (int k = 0; k < LOSS_NPART; ++k) part[k] = 0.f;
    if (row_ok) {
      const float lp = a.logp[m];
      const float qpi = zfsum / (float)Nq;
      const float pil = -(alpha * (-lp)) - qpi;
      const float all = -(log_alpha * (a.target_entropy - (-lp)));
      a.q_loss[m] = qloss;
      a.pi_loss[m] = pil;
      a.alpha_loss[m] = all;
      part[0] = wm * ((qloss + pil) + all);
      part[1] = qloss; part[2] = pil; part[3] = all;
      part[4] = qsum; part[5] = viol;
      part[6] = -wm * (a.target_entropy + lp);   // d loss / d log_alpha
    }
#pragma unroll
    for (int k = 0; k < LOSS_NPART; ++k) bsum[g * LOSS_NPART + k] = part[k];
  }
  __syncthreads();
  if (tid < LOSS_NPART) {
    float s = 0.f;
    for (int r = 0; r < RPB; ++r) s += bsum[r * LOSS_NPART + tid];
    a.partials[(long long)blockIdx.x * LOSS_NPART + tid] = s;
  }
  if (a.fin) {   // (uniform) fused finish: the last workgroup to arrive sums all partial rows
    __shared__ int is_last;
    __threadfence();                       // this workgroup's row is visible device-wide before it is counted
    __syncthreads();
    const LossFinishArgs f = *a.fin;
    if (tid == 0) is_last = atomicAdd(&f.st->loss_blocks_done, 1u) + 1u == (unsigned)gridDim.x;
    __syncthreads();
    if (is_last) {
      __threadfence();
      loss_finish_block(a.partials, f.nblocks, f.M, f.Nq, f.st, f.scalars, f.dlog_alpha, f.lr, f.b1, f.b2);
      if (tid == 0) f.st->loss_blocks_done = 0u;   // ready for the next step
    }
  }
}

// TQC loss for many atoms (32 < Nq <= 128: BASELINE config 4 has 125): ONE WAVE PER ROW, the row's atoms in registers
// (SLOTS per lane: atom lane + 64 s).  The thread-per-atom kernel above does 125 LDS reads per thread to rank its atom and
// 100 more for the pair terms, with ~28 workgroup barriers around its group sums; here
//   * the pooled atoms are SORTED in registers (bitonic network: 28 compare-exchange stages over 128 elements, partners by
//     lane shuffle / between a lane's own slots) - equal atoms give equal targets, so no stable tie-break is needed for the values
//     torch.sort()[0][:Nt] returns (distributional_soft_actor_critic.py:50-53);
//   * target j of the pair loop is a v_readlane broadcast, shared by the lane's SLOTS atoms; the pair terms are accumulated in the
//     same order j = 0 .. Nt - 1 (the Huber term in a branch-free form with the same fp32 values);
//   * the four per-row sums are wave reductions in the order of the LDS tree above (slot 1 onto slot 0, then lane + off).
template <int SLOTS>
__global__ __launch_bounds__(256) void k_loss_wave(LossArgs a) {
  __shared__ float bsum[4 * LOSS_NPART];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int m = blockIdx.x * 4 + wv;
  const bool row_ok = m < a.M;
  const int mm = row_ok ? m : a.M - 1;
  const int Nq = a.Nq, Nt = a.Nt;
  const float alpha = a.st->alpha_cur;
  const float log_alpha = *a.log_alpha;
  const float lpn = a.logp_next[mm];
  const float ent = alpha * (-lpn);
  const float rew = a.reward[mm + a.B];
  const float maskg = (a.task_done[mm + a.B] == 0.f ? 1.f : 0.f) * a.gamma;
  const float mc = a.mc_return ? a.mc_return[mm + a.B] : 0.f;
  const float wm = a.w[mm];
  float z[SLOTS], q[SLOTS], zf[SLOTS];
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    const int i = lane + 64 * s;
    z[s] = i < Nq ? a.z_target[(long long)mm * Nq + i] : INFINITY;
    q[s] = i < Nq ? a.q_pred[(long long)mm * Nq + i] : 0.f;
    zf[s] = i < Nq ? a.z_frozen[(long long)mm * Nq + i] : 0.f;
  }
  // ---- ascending sort of the 64 SLOTS values: element e = lane + 64 slot
#pragma unroll
  for (int k = 2; k <= 64 * SLOTS; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j >= 1; j >>= 1) {
      if (j == 64) {   // (SLOTS == 2, k == 128) partner in the lane's other slot: ascending
        const float lo = fminf(z[0], z[SLOTS - 1]), hi = fmaxf(z[0], z[SLOTS - 1]);
        z[0] = lo; z[SLOTS - 1] = hi;
      } else {
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
          const int e = lane + 64 * s;
          const float other = __shfl_xor(z[s], j, 64);
          const bool asc = (e & k) == 0, low = (lane & j) == 0;
          z[s] = (asc == low) ? fminf(z[s], other) : fmaxf(z[s], other);
        }
      }
    }
  }
  // ---- targets of the kept atoms (rank r = lane + 64 slot < Nt): entropy bonus after truncation, then the Bellman backup
  float y[SLOTS];
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    const int r = lane + 64 * s;
    float tq = z[s];
    if (a.max_entropy) tq = tq + ent;
    y[s] = rew + maskg * tq;
    if (row_ok && r < Nt) a.td_target[(long long)m * Nt + r] = y[s];
  }
  // ---- pairwise quantile-Huber terms, j = 0 .. Nt - 1
  float tau[SLOTS], omt[SLOTS], sl[SLOTS], sg[SLOTS];
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    tau[s] = (float)(lane + 64 * s) / (float)Nq + a.half_inv_nq;
    omt[s] = fabsf(tau[s] - 1.f);
    sl[s] = 0.f; sg[s] = 0.f;
  }
#pragma unroll
  for (int ys = 0; ys < SLOTS; ++ys) {
    const int jn = min(64, Nt - 64 * ys);   // uniform
    for (int jj = 0; jj < jn; ++jj) {
      const float yj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, y[ys]), jj));
#pragma unroll
      for (int s = 0; s < SLOTS; ++s) {
        // 6.5 vector instructions per pair (the literal formulas are 12.5): with c = clamp(d, -1, 1) the Huber value
        // |d| > 1 ? |d| - 0.5 : 0.5 d^2 is |c| (|d| - 0.5 |c|) - the same fp32 value in both branches (0.5 x is exact) - and
        // its derivative is c itself
        const float d = yj - q[s];
        const float c = __builtin_amdgcn_fmed3f(d, -1.f, 1.f);
        float t, hub;   // |c| and |d| as source modifiers (the compiler's packed-math pairing would materialise them with v_and)
        asm("v_fma_f32 %0, |%1|, -0.5, |%2|" : "=v"(t) : "v"(c), "v"(d));
        asm("v_mul_f32 %0, |%1|, %2" : "=v"(hub) : "v"(c), "v"(t));
        const float wt = d < 0.f ? omt[s] : tau[s];
        sl[s] = fmaf(wt, hub, sl[s]);
        sg[s] = fmaf(-wt, c, sg[s]);
      }
    }
  }
  float tot[SLOTS], viol[SLOTS];
  const float inv = 1.f / (float)(Nq * Nt);
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    const int i = lane + 64 * s;
    float loss_i = 0.f, grad_i = 0.f, lb_i = 0.f;
    viol[s] = 0.f;
    if (i < Nq) {
      loss_i = sl[s] * inv;
      grad_i = sg[s] * inv;
      if (a.lowerbound) {
        const float lb = mc - q[s];
        if (lb > 0.f) { lb_i = lb / (float)Nq; grad_i -= 1.f / (float)Nq; viol[s] = 1.f; }
      }
      if (row_ok) {
        a.dz[(long long)m * Nq + i] = wm * grad_i;
        a.dzf[(long long)m * Nq + i] = -wm / (float)Nq;
      }
    }
    tot[s] = loss_i + lb_i;
  }
  auto row_sum = [&](const float (&x)[SLOTS]) {
    float v = x[0];
    if (SLOTS == 2) v += x[SLOTS - 1];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_down(v, off, 64);
    return v;   // lane 0
  };
  const float qloss = row_sum(tot), zfsum = row_sum(zf), qsum = row_sum(q), nviol = row_sum(viol);
  if (lane == 0) {
    float part[LOSS_NPART];
#pragma unroll
    for (int k = 0; k < LOSS_NPART; ++k) part[k] = 0.f;
    if (row_ok) {
      const float lp = a.logp[m];
      const float qpi = zfsum / (float)Nq;
      const float pil = -(alpha * (-lp)) - qpi;
      const float all = -(log_alpha * (a.target_entropy - (-lp)));
      a.q_loss[m] = qloss;
      a.pi_loss[m] = pil;
      a.alpha_loss[m] = all;
      part[0] = wm * ((qloss + pil) + all);
      part[1] = qloss; part[2] = pil; part[3] = all;
      part[4] = qsum; part[5] = nviol;
      part[6] = -wm * (a.target_entropy + lp);   // d loss / d log_alpha
    }
#pragma unroll
    for (int k = 0; k < LOSS_NPART; ++k) bsum[wv * LOSS_NPART + k] = part[k];
  }
  __syncthreads();
  if (tid < LOSS_NPART) {
    float s = 0.f;
    for (int r = 0; r < 4; ++r) s += bsum[r * LOSS_NPART + tid];
    a.partials[(long long)blockIdx.x * LOSS_NPART + tid] = s;
  }
  if (a.fin) {   // (uniform) fused finish, as in k_loss
    __shared__ int is_last;
    __threadfence();
    __syncthreads();
    const LossFinishArgs f = *a.fin;
    if (tid == 0) is_last = atomicAdd(&f.st->loss_blocks_done, 1u) + 1u == (unsigned)gridDim.x;
    __syncthreads();
    if (is_last) {
      __threadfence();
      loss_finish_block(a.partials, f.nblocks, f.M, f.Nq, f.st, f.scalars, f.dlog_alpha, f.lr, f.b1, f.b2);
      if (tid == 0) f.st->loss_blocks_done = 0u;
    }
  }
}

// wave-per-row form: TQC with 32 < Nq <= 128 (LossArgs::G == 64 then: four rows per workgroup); FDQL_LOSS_WAVE=0: never
bool loss_wave_form(int distributional, int Nq) {
  return distributional && Nq > 32 && Nq <= 128 && plan_switches().loss_wave;
}

hipError_t loss_launch(const LossArgs &a, hipStream_t s) {
  if (a.G == 64 && a.Nq > 32 && a.distributional && loss_wave_form(a.distributional, a.Nq)) {
    const int blocks = (a.M + 3) / 4;
    if (a.Nq <= 64) hipLaunchKernelGGL(k_loss_wave<1>, dim3(blocks), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(k_loss_wave<2>, dim3(blocks), dim3(256), 0, s, a);
    return hipGetLastError();
  }
  const int RPB = 256 / a.G;
  const int blocks = (a.M + RPB - 1) / RPB;
  const size_t lds = (size_t)(3 * RPB * a.G + RPB * LOSS_NPART) * sizeof(float);
  hipLaunchKernelGGL(k_loss, dim3(blocks), dim3(256), lds, s, a);
  return hipGetLastError();
}

// Sums the per-block partials in a fixed order; publishes the scalars and d log_alpha.
__device__ __forceinline__ void loss_finish_block(const float *__restrict__ partials, int nblocks, int M, int Nq,
                                                  DevState *st, float *__restrict__ scalars,
                                                  float *__restrict__ dlog_alpha, double lr, double b1, double b2) {
  // Called by EVERY thread of a workgroup of >= 256 threads (the barriers below are workgroup barriers); the first 256 do
  // the work, in the same order whatever the workgroup's size.
  __shared__ float red[256][LOSS_NPART];
  if (threadIdx.x == 255) tick_adam(st, lr, b1, b2);
  const int tid = threadIdx.x;
  if (tid < 256) {
    float acc[LOSS_NPART];
#pragma unroll
    for (int k = 0; k < LOSS_NPART; ++k) acc[k] = 0.f;
    for (int b = tid; b < nblocks; b += 256)
#pragma unroll
      for (int k = 0; k < LOSS_NPART; ++k) acc[k] += partials[(long long)b * LOSS_NPART + k];
#pragma unroll
    for (int k = 0; k < LOSS_NPART; ++k) red[tid][k] = acc[k];
  }
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if (tid < off)
#pragma unroll
      for (int k = 0; k < LOSS_NPART; ++k) red[tid][k] += red[tid + off][k];
    __syncthreads();
  }
  if (tid == 0) {
    scalars[0] = red[0][0];
    scalars[1] = red[0][1] / (float)M;
    scalars[2] = red[0][2] / (float)M;
    scalars[3] = red[0][3] / (float)M;
    scalars[4] = red[0][4] / ((float)M * (float)Nq);
    scalars[5] = red[0][5] / ((float)M * (float)Nq);
    scalars[6] = st->alpha_cur;
    scalars[7] = (float)st->step;
    *dlog_alpha = red[0][6];
  }
}
__global__ __launch_bounds__(256) void k_loss_finish(const float *__restrict__ partials, int nblocks, int M, int Nq,
                                                     DevState *st, float *__restrict__ scalars,
                                                     float *__restrict__ dlog_alpha, double lr, double b1, double b2) {
  loss_finish_block(partials, nblocks, M, Nq, st, scalars, dlog_alpha, lr, b1, b2);
}

// window-long n-step lower bound on q(t=0): one thread per window walks its T-1 transitions
// (B threads, a few hundred floats each: latency-bound, off the critical path of nothing else)
__global__ __launch_bounds__(256) void k_boot_lowerbound(BootArgs a) {
  __shared__ float red[256];
  float local = 0.f;
  for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < a.B; b += gridDim.x * blockDim.x) {
    float ret = 0.f, valid = 1.f, gp = 1.f;
    for (int t = 0; t < a.T - 1; ++t) {
      const int m = t * a.B + b;
      ret += a.reward[m + a.B] * gp;                        // next_xp["reward"] * gamma^t
      gp *= a.gamma;
      valid *= (a.task_done[m + a.B] == 0.f ? 1.f : 0.f);   // next_xp["mask"].prod(0)
      valid *= a.contig[m];                                 // xp["is_contiguous"].prod(0)
    }
    const float bound = ret + gp * a.td_target[(long long)(a.T - 2) * a.B + b];   // gp == gamma^(T-1); SAC: Nt == 1
    for (int j = 0; j < a.Nq; ++j) {
      const float x = bound - a.q_pred[(long long)b * a.Nq + j];
      if (x > 0.f && valid != 0.f) {
        local += x;
        a.dz[(long long)b * a.Nq + j] -= a.scale;
      }
    }
  }
  red[threadIdx.x] = local;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    for (int k = 0; k < LOSS_NPART; ++k) a.partial_row[k] = 0.f;
    a.partial_row[0] = red[0] * a.scale;
  }
}

hipError_t boot_lowerbound_launch(const BootArgs &a, hipStream_t s) {
  hipLaunchKernelGGL(k_boot_lowerbound, dim3(1), dim3(256), 0, s, a);   // one block: the row is a plain store
  return hipGetLastError();
}

hipError_t loss_finish_launch(const float *partials, int nblocks, int M, int Nq, DevState *st, float *scalars,
                              float *dlog_alpha, double lr, double b1, double b2, hipStream_t s) {
  hipLaunchKernelGGL(k_loss_finish, dim3(1), dim3(256), 0, s, partials, nblocks, M, Nq, st, scalars, dlog_alpha, lr, b1, b2);
  return hipGetLastError();
}

// ======================================================================================
// Slab reduction, Adam + polyak (+ critic_frozen copy)
// ======================================================================================
__global__ void k_reduce_slabs(const float4 *__restrict__ slabs, int nslab, long long n4, long long stride4, float4 *__restrict__ grads) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  float4 s = slabs[i];
  for (int k = 1; k < nslab; ++k) {
    const float4 v = slabs[(long long)k * stride4 + i];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  grads[i] = s;
}

hipError_t reduce_slabs_launch(const float *slabs, int nslab, long long n, float *grads, hipStream_t s) {
  return reduce_slabs_range_launch(slabs, nslab, n, 0, n, grads, s);
}
// the sub-range [first, first + count) of the arena (both multiples of 4): slabs of `stride` floats each
hipError_t reduce_slabs_range_launch(const float *slabs, int nslab, long long stride, long long first, long long count, float *grads,
                                     hipStream_t s) {
  const long long n4 = count / 4;
  if (n4 <= 0) return hipSuccess;
  const int blocks = (int)((n4 + 255) / 256);
  hipLaunchKernelGGL(k_reduce_slabs, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const float4 *>(slabs + first), nslab, n4,
                     stride / 4, reinterpret_cast<float4 *>(grads + first));
  return hipGetLastError();
}

// One thread per 4 parameters (the arena length is a multiple of 4).  a.slabs != null (single-process step): the
// K-split slab sum (k_reduce_slabs' order) is formed here and written to grads on the way.  Thread 0 advances the optimiser step.
__global__ __launch_bounds__(256) void k_adam_polyak(AdamArgs a) {
  const long long i4 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long n4 = a.n / 4;
  const float bc2_sqrt = a.st->bc2_sqrt, neg_step_size = a.st->neg_step_size;
  if (i4 < n4) {
    float4 gv;
    if (a.slabs) {
      const float4 *sl = reinterpret_cast<const float4 *>(a.slabs) + i4;
      gv = sl[0];
      int k = 1;
      for (; k + 4 <= a.nslab; k += 4) {   // four loads in flight, added in slab order
        const float4 v0 = sl[(long long)k * n4], v1 = sl[(long long)(k + 1) * n4], v2 = sl[(long long)(k + 2) * n4],
                     v3 = sl[(long long)(k + 3) * n4];
        gv.x += v0.x; gv.y += v0.y; gv.z += v0.z; gv.w += v0.w;
        gv.x += v1.x; gv.y += v1.y; gv.z += v1.z; gv.w += v1.w;
        gv.x += v2.x; gv.y += v2.y; gv.z += v2.z; gv.w += v2.w;
        gv.x += v3.x; gv.y += v3.y; gv.z += v3.z; gv.w += v3.w;
      }
      for (; k < a.nslab; ++k) {
        const float4 v = sl[(long long)k * n4];
        gv.x += v.x; gv.y += v.y; gv.z += v.z; gv.w += v.w;
      }
      reinterpret_cast<float4 *>(a.grads_out)[i4] = gv;
    } else {
      gv = reinterpret_cast<const float4 *>(a.grads)[i4];
    }
    const float4 pv = reinterpret_cast<const float4 *>(a.params)[i4];
    float4 mv = reinterpret_cast<const float4 *>(a.m)[i4], vv = reinterpret_cast<const float4 *>(a.v)[i4], pnv;
    const float gs[4] = {gv.x, gv.y, gv.z, gv.w}, ps[4] = {pv.x, pv.y, pv.z, pv.w};
    float ms[4] = {mv.x, mv.y, mv.z, mv.w}, vs[4] = {vv.x, vv.y, vv.z, vv.w}, pn[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float g = gs[c] * a.grad_scale;
      ms[c] = ms[c] + a.one_minus_b1 * (g - ms[c]);           // exp_avg.lerp_(grad, 1 - beta1)
      vs[c] = vs[c] * a.b2 + (a.one_minus_b2 * g) * g;        // exp_avg_sq.mul_(b2).addcmul_(g, g, 1 - b2)
      const float denom = sqrtf(vs[c]) / bc2_sqrt + a.eps;
      pn[c] = ps[c] + (neg_step_size * ms[c]) / denom;        // param.addcdiv_(m, denom, -step_size)
    }
    mv = make_float4(ms[0], ms[1], ms[2], ms[3]);
    vv = make_float4(vs[0], vs[1], vs[2], vs[3]);
    pnv = make_float4(pn[0], pn[1], pn[2], pn[3]);
    reinterpret_cast<float4 *>(a.m)[i4] = mv;
    reinterpret_cast<float4 *>(a.v)[i4] = vv;
    reinterpret_cast<float4 *>(a.params)[i4] = pnv;
    const long long i0 = i4 * 4;
    if (i0 + 3 >= a.tgt_begin && i0 < a.tgt_end) {            // common.py:10-19
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const long long i = i0 + c;
        if (i >= a.tgt_begin && i < a.tgt_end) {
          const long long j = i - a.tgt_begin;
          a.targets[j] = a.hard ? pn[c] : a.targets[j] * a.one_minus_tau + pn[c] * a.tau;
        }
      }
    }
    if (a.frozen && i0 + 3 >= a.frozen_begin && i0 < a.frozen_end) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const long long i = i0 + c;
        if (i >= a.frozen_begin && i < a.frozen_end) a.frozen[i - a.frozen_begin] = ps[c];
      }
    }
  }
  // the step counter: nothing in this kernel reads it (the corrections above were derived from step + 1 by
  // k_loss_finish), so one thread can advance it without ordering against the other workgroups
  if (i4 == 0) a.st->step += 1;
}

hipError_t adam_launch(const AdamArgs &a, hipStream_t s) {
  const int blocks = (int)((a.n / 4 + 255) / 256);
  hipLaunchKernelGGL(k_adam_polyak, dim3(blocks), dim3(256), 0, s, a);
  return hipGetLastError();
}

hipError_t prep_launch(const float *task_done, const float *episode_step, int T, int B, int burn_in, int cumprod,
                       float inv_gb, float *w, float *contig, DevState *st, const float *log_alpha, hipStream_t s) {
  hipLaunchKernelGGL(k_prep, dim3((B + 15) / 16), dim3(256), 0, s, task_done, episode_step, T, B, burn_in, cumprod, inv_gb, w,
                     contig, st, log_alpha);
  return hipGetLastError();
}

hipError_t policy_fwd_launch(const PolicyFwdArgs &a0, const PolicyFwdArgs &a1, int nprob, int M, int A,
                             const DevState *st, uint64_t seed, int discrete, hipStream_t s, const PrepArgs *prep) {
  const int total = nprob * M;
  PrepArgs pr = {};
  if (prep && !discrete) pr = *prep;
  else if (prep) {   // (the Gumbel kernel has its own block shape: prep stays a launch of its own)
    hipError_t e = prep_launch(prep->task_done, prep->episode_step, prep->T, prep->B, prep->burn_in, prep->cumprod, prep->inv_global_batch,
                               prep->w, prep->contig, prep->st, prep->log_alpha, s);
    if (e != hipSuccess) return e;
  }
  const int extra = pr.B > 0 ? (pr.B + 15) / 16 : 0;
  if (discrete)
    hipLaunchKernelGGL(k_policy_fwd_gumbel, dim3((total + 63) / 64), dim3(64), 0, s, a0, a1, nprob, M, A, st, seed);
  else if (A <= 8) {
    const int own = (int)(((long long)total * 8 + 255) / 256);
    hipLaunchKernelGGL(k_policy_fwd<8>, dim3((unsigned)(own + extra)), dim3(256), 0, s, a0, a1, nprob, M, A, st, seed, pr, own);
  } else if (A <= 32) {
    const int own = (int)(((long long)total * 32 + 255) / 256);
    hipLaunchKernelGGL(k_policy_fwd<32>, dim3((unsigned)(own + extra)), dim3(256), 0, s, a0, a1, nprob, M, A, st, seed, pr, own);
  } else if (A <= 64) {
    const int own = (int)(((long long)total * 64 + 255) / 256);
    hipLaunchKernelGGL(k_policy_fwd<64>, dim3((unsigned)(own + extra)), dim3(256), 0, s, a0, a1, nprob, M, A, st, seed, pr, own);
  } else
    return hipErrorInvalidValue;   // (act_dim <= 64 is checked at fdql_agent_create)
  return hipGetLastError();
}

hipError_t onehot_launch(const float *action, int rows, int n, float *out, hipStream_t s) {
  hipLaunchKernelGGL(k_onehot, dim3((rows * n + 255) / 256), dim3(256), 0, s, action, rows, n, out);
  return hipGetLastError();
}

bool policy_bwd_dpre_takes(int discrete, int A, int hidden) { return !discrete && 2 * A <= PBD_MAXK && A >= 1 && hidden == PBD_N; }
hipError_t policy_bwd_dpre_launch(const float *logits, const float *noise, const float *action, const float *dpi_parts, int nparts,
                                  float *dpi_sum, const float *w, const DevState *st, int M, int A, float *dlogits, const float *Wh,
                                  int ldw, const float *h, float *dpre, float *colsum, hipStream_t s, const float *loss_partials,
                                  const LossFinishArgs *fin) {
  if (M <= 0) return hipSuccess;
  if (2 * A > PBD_MAXK || (reinterpret_cast<uintptr_t>(h) & 15) || (reinterpret_cast<uintptr_t>(dpre) & 15)) return hipErrorInvalidValue;
  LossFinishRider r = {};
  if (fin) { r.partials = loss_partials; r.f = *fin; }
  PolicyBwdDpreArgs d = {Wh, ldw, h, dpre, colsum};
  const int own = (M + PBD_ROWS - 1) / PBD_ROWS;
  hipLaunchKernelGGL(k_policy_bwd_dpre, dim3((unsigned)(own + (fin ? 1 : 0))), dim3(1024), 0, s, logits, noise, action, dpi_parts, nparts,
                     dpi_sum, w, st, M, A, dlogits, d, r, own);
  return hipGetLastError();
}
hipError_t policy_bwd_launch(const float *logits, const float *noise, const float *action, const float *dpi_parts,
                             int nparts, float *dpi_sum, const float *w, const DevState *st, int M, int A,
                             float *dlogits, int discrete, hipStream_t s, const float *loss_partials, const LossFinishArgs *fin) {
  if (discrete) {
    hipLaunchKernelGGL(k_policy_bwd_gumbel, dim3((M + 63) / 64), dim3(64), 0, s, logits, noise, dpi_parts, nparts, dpi_sum,
                       w, st, M, A, dlogits);
    if (fin) {   // (the Gumbel kernel has its own block shape: the finish stays a launch of its own)
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) return e;
      return loss_finish_launch(loss_partials, fin->nblocks, fin->M, fin->Nq, fin->st, fin->scalars, fin->dlog_alpha, fin->lr, fin->b1, fin->b2, s);
    }
  } else {
    LossFinishRider r = {};
    if (fin) { r.partials = loss_partials; r.f = *fin; }
    const int own = (int)(((long long)M * A + 255) / 256);
    hipLaunchKernelGGL(k_policy_bwd, dim3((unsigned)(own + (fin ? 1 : 0))), dim3(256), 0, s, logits, noise, action, dpi_parts, nparts,
                       dpi_sum, w, st, M, A, dlogits, r, own);
  }
  return hipGetLastError();
}

// ======================================================================================
// Pixel encoder: im2col / col2im around the grouped GEMM (no reference; see include/fdql.h)
// ======================================================================================
// layer 0: NCHW frames, K ordered (c, ky, kx); one thread copies the k contiguous pixels of one (row, c, ky)
__global__ void k_im2col_nchw(const float *__restrict__ in, float scale, long long n_img, ConvGeom g, float *__restrict__ col) {
  const int K = g.C * g.k * g.k, per_row = g.C * g.k;
  const long long total = n_img * g.OH * g.OW * per_row;
  for (long long u = (long long)blockIdx.x * blockDim.x + threadIdx.x; u < total; u += (long long)gridDim.x * blockDim.x) {
    const long long row = u / per_row;
    const int ck = (int)(u - row * per_row), c = ck / g.k, ky = ck - c * g.k;
    const long long img = row / (g.OH * g.OW);
    const int pos = (int)(row - img * g.OH * g.OW), oy = pos / g.OW, ox = pos - oy * g.OW;
    const float *src = in + ((img * g.C + c) * g.H + (oy * g.s + ky)) * g.W + ox * g.s;
    float *dst = col + row * K + ck * g.k;
    if ((g.k & 3) == 0 && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0) {
      for (int kx = 0; kx < g.k; kx += 4) {   // e.g. the 8-pixel runs of an 8x8 stride-4 first layer
        float4 v = *reinterpret_cast<const float4 *>(src + kx);
        v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
        *reinterpret_cast<float4 *>(dst + kx) = v;
      }
    } else {
      for (int kx = 0; kx < g.k; ++kx) dst[kx] = src[kx] * scale;
    }
  }
}
// layer 0, staged: a block owns one output row (img, oy): the C*k input rows it touches are read once,
// coalesced, into LDS and the OW x K block of col - contiguous in memory - is written coalesced from there
constexpr int IM2COL_LDS_FLOATS = 12288;
__global__ __launch_bounds__(256) void k_im2col_nchw_rows(const float *__restrict__ in, float scale, long long n_img,
                                                          ConvGeom g, float *__restrict__ col) {
  extern __shared__ float tile[];   // C * k * W floats
  const int K = g.C * g.k * g.k, kk2 = g.k * g.k, nin = g.C * g.k * g.W;
  const long long total = n_img * g.OH;
  for (long long blk = blockIdx.x; blk < total; blk += gridDim.x) {
    const long long img = blk / g.OH;
    const int oy = (int)(blk - img * g.OH);
    __syncthreads();
    for (int i = threadIdx.x; i < nin; i += 256) {
      const int cr = i / g.W, x = i - cr * g.W, c = cr / g.k, r = cr - c * g.k;
      tile[i] = in[((img * g.C + c) * g.H + oy * g.s + r) * g.W + x] * scale;
    }
    __syncthreads();
    float *dst = col + (blk * g.OW) * K;
    for (int kk = threadIdx.x; kk < K; kk += 256) {   // a thread keeps its (c, ky, kx) and walks the output positions
      const int c = kk / kk2, r2 = kk - c * kk2, ky = r2 / g.k, kx = r2 - ky * g.k;
      const float *src = tile + (c * g.k + ky) * g.W + kx;
      for (int ox = 0; ox < g.OW; ++ox) dst[(long long)ox * K + kk] = src[ox * g.s];
    }
  }
}
// later layers: NHWC feature maps, K ordered (ky, kx, c); one thread copies 4 channels of one (row, ky, kx)
__global__ void k_im2col_nhwc(const float *__restrict__ in, long long n_img, ConvGeom g, float *__restrict__ col) {
  const int K = g.C * g.k * g.k, c4n = g.C >> 2, per_row = g.k * g.k * c4n;
  const long long total = n_img * g.OH * g.OW * per_row;
  for (long long u = (long long)blockIdx.x * blockDim.x + threadIdx.x; u < total; u += (long long)gridDim.x * blockDim.x) {
    const long long row = u / per_row;
    const int r = (int)(u - row * per_row), kk = r / c4n, c4 = r - kk * c4n, ky = kk / g.k, kx = kk - ky * g.k;
    const long long img = row / (g.OH * g.OW);
    const int pos = (int)(row - img * g.OH * g.OW), oy = pos / g.OW, ox = pos - oy * g.OW;
    const float4 v = *reinterpret_cast<const float4 *>(in + ((img * g.H + oy * g.s + ky) * g.W + ox * g.s + kx) * g.C + 4 * c4);
    *reinterpret_cast<float4 *>(col + row * K + kk * g.C + 4 * c4) = v;
  }
}

// d(pre-activation of the previous layer), NHWC: gather over the windows covering a pixel; K ordered (ky, kx, c)
__global__ void k_col2im_mask(const float *__restrict__ dcol, const float *__restrict__ act_prev, long long n_img, ConvGeom g,
                              float *__restrict__ dpre_prev) {
  const int K = g.C * g.k * g.k;
  const long long total = n_img * g.H * g.W * g.C;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(e % g.C);
    const long long p = e / g.C;
    const int x = (int)(p % g.W);
    const long long q = p / g.W;
    const int y = (int)(q % g.H);
    const long long img = q / g.H;
    float acc = 0.f;
    for (int ky = y % g.s; ky < g.k; ky += g.s) {       // windows with oy*s + ky == y
      const int oy = (y - ky) / g.s;
      if (y < ky || oy >= g.OH) continue;
      for (int kx = x % g.s; kx < g.k; kx += g.s) {
        const int ox = (x - kx) / g.s;
        if (x < kx || ox >= g.OW) continue;
        acc += dcol[((img * g.OH + oy) * g.OW + ox) * K + (ky * g.k + kx) * g.C + c];
      }
    }
    dpre_prev[e] = act_prev[e] > 0.f ? acc : 0.01f * acc;
  }
}

hipError_t im2col_launch(const float *in, int nhwc, float scale, long long n_img, const ConvGeom &g, float *col, hipStream_t s) {
  const long long units = n_img * g.OH * g.OW * (nhwc ? g.k * g.k * (g.C >> 2) : g.C * g.k);
  if (units <= 0) return hipSuccess;
  const int blocks = (int)std::min<long long>((units + 255) / 256, 1 << 20);
  if (nhwc) {
    hipLaunchKernelGGL(k_im2col_nhwc, dim3(blocks), dim3(256), 0, s, in, n_img, g, col);
  } else if (g.C * g.k * g.W <= IM2COL_LDS_FLOATS) {
    const int rb = (int)std::min<long long>(n_img * g.OH, 1 << 20);
    hipLaunchKernelGGL(k_im2col_nchw_rows, dim3(rb), dim3(256), (size_t)g.C * g.k * g.W * sizeof(float), s, in, scale, n_img, g, col);
  } else {
    hipLaunchKernelGGL(k_im2col_nchw, dim3(blocks), dim3(256), 0, s, in, scale, n_img, g, col);
  }
  return hipGetLastError();
}
hipError_t col2im_mask_launch(const float *dcol, const float *act_prev, long long n_img, const ConvGeom &g, float *dpre_prev,
                              hipStream_t s) {
  const long long total = n_img * g.H * g.W * g.C;
  if (total <= 0) return hipSuccess;
  const int blocks = (int)std::min<long long>((total + 255) / 256, 1 << 20);
  hipLaunchKernelGGL(k_col2im_mask, dim3(blocks), dim3(256), 0, s, dcol, act_prev, n_img, g, dpre_prev);
  return hipGetLastError();
}

// tall-matrix column sums (conv bias gradients: millions of rows, a few dozen columns)
__global__ __launch_bounds__(256) void k_colsum_tall(const float *__restrict__ X, long long R, int C, int ld,
                                                     float *__restrict__ partial) {
  __shared__ float red[256];
  const int col = threadIdx.x % C, sub = threadIdx.x / C, nsub = 256 / C;   // C <= 256
  const long long r0 = (long long)blockIdx.x * COLSUM_TALL_ROWS, r1 = min(R, r0 + COLSUM_TALL_ROWS);
  float acc = 0.f;
  if (sub < nsub)
  {
    float acc2 = 0.f;
    long long r = r0 + sub;
    for (; r + nsub < r1; r += 2 * nsub) { acc += X[r * ld + col]; acc2 += X[(r + nsub) * ld + col]; }
    if (r < r1) acc += X[r * ld + col];
    acc += acc2;
  }
  red[threadIdx.x] = sub < nsub ? acc : 0.f;
  __syncthreads();
  if (threadIdx.x < C) {
    float t = 0.f;
    for (int j = 0; j < nsub; ++j) t += red[j * C + threadIdx.x];
    partial[(long long)blockIdx.x * C + threadIdx.x] = t;
  }
}

// block = 64 elements x 4 part lanes; each lane sums every 4th partial, LDS folds the 4 lanes (fixed order)
__global__ __launch_bounds__(256) void k_reduce_partials(const float *__restrict__ part, int nparts, long long n,
                                                         float *__restrict__ dst) {
  __shared__ float red[4][64];
  const int el = threadIdx.x & 63, pl = threadIdx.x >> 6;
  const long long e = (long long)blockIdx.x * 64 + el;
  part += (long long)blockIdx.y * nparts * n;   // gridDim.y independent instances, back to back
  dst += (long long)blockIdx.y * n;
  float s0 = 0.f, s1 = 0.f;
  if (e < n) {
    int p = pl;
    for (; p + 4 < nparts; p += 8) { s0 += part[(long long)p * n + e]; s1 += part[(long long)(p + 4) * n + e]; }
    if (p < nparts) s0 += part[(long long)p * n + e];
  }
  red[pl][el] = s0 + s1;
  __syncthreads();
  if (pl == 0 && e < n) dst[e] = (red[0][el] + red[1][el]) + (red[2][el] + red[3][el]);
}

// Many partials (the conv weight gradients' slab per workgroup: 256 of them): 32 quads x 8 part lanes per block, eight 16-byte
// loads in flight per lane (the narrow form above keeps two 4-byte loads in flight: 0.19 ms for the 79 MB of config 5's three
// layers, latency-bound at 0.4 TB/s).  Fixed order: a lane adds its parts in ascending order, LDS folds the lanes in order.
typedef float rp_v4f __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_reduce_partials_wide(const float *__restrict__ part, int nparts, long long n, float *__restrict__ dst) {
  __shared__ rp_v4f red[8][32];
  const int ql = threadIdx.x & 31, pl = threadIdx.x >> 5;
  const long long qd = (long long)blockIdx.x * 32 + ql, nq = n >> 2;
  rp_v4f s = {0.f, 0.f, 0.f, 0.f};
  if (qd < nq) {
    const rp_v4f *src = reinterpret_cast<const rp_v4f *>(part) + qd;
    int p = pl;
    for (; p + 56 < nparts; p += 64) {
      rp_v4f v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[(long long)(p + 8 * u) * nq];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; p < nparts; p += 8) s += src[(long long)p * nq];
  }
  red[pl][ql] = s;
  __syncthreads();
  if (pl == 0 && qd < nq) {
    rp_v4f t = red[0][ql];
#pragma unroll
    for (int k = 1; k < 8; ++k) t += red[k][ql];
    reinterpret_cast<rp_v4f *>(dst)[qd] = t;
  }
}

// see update_kernels.h: the critics' head finish.  grid (ceil(M / 16), groups), one wave = 16 rows per workgroup.
// v_mfma_f32_16x16x4_f32: lane l holds A[row l & 15][k = 4 (l >> 4) + c] and B[k][column l & 15] for step c of a 16-k
// block; D: column l & 15, rows 4 (l >> 4) + reg.
typedef float hf_v4f __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64) void k_head_finish(HeadFinishArgs a) {
  const HeadFinishGroup &G = a.g[blockIdx.y];
  const int lane = threadIdx.x & 63;
  const int L = a.L, A = a.A, Q = a.Q, M = a.M;
  const int i = lane & 15, kq = lane >> 4;
  const int row0 = blockIdx.x * 16;
  if (row0 >= M) return;
  const int set = i / Q, q = i - set * Q;
  const bool nok = set < G.nsets;                      // this lane's head column exists
  // per-lane table entries picked by compares from wave-uniform (scalar) loads of the whole table: indexing the kernel
  // argument with a per-lane index is a vector load from memory - a dependent round trip before the first real load
  const float *wsel = G.Wh[0], *bsel = G.bias[0], *psel0 = G.parts[0][0], *psel1 = G.parts[0][1];
#pragma unroll
  for (int k = 1; k < HEAD_FINISH_MAX_SETS; ++k) {
    const bool m = set == k;
    wsel = m ? G.Wh[k] : wsel; bsel = m ? G.bias[k] : bsel; psel0 = m ? G.parts[k][0] : psel0; psel1 = m ? G.parts[k][1] : psel1;
  }
  const float *wrow = wsel + (long long)q * G.ldw;
  const float *srow = G.s + (long long)min(row0 + i, M - 1) * G.lds;
  // everything the wave needs from memory is requested before the first MFMA: the action blocks, the summed parts and
  // the bias here, the state / weight rows below (a wave is 16 rows: its time is memory round trips, not arithmetic)
  float aval[2][4], wact[4], pval[2][4], bq = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c) wact[c] = (nok && 4 * kq + c < A) ? wrow[L + 4 * kq + c] : 0.f;   // (A <= 16 on this path)
#pragma unroll
  for (int v = 0; v < 2; ++v) {
    const bool vok = v < G.nvar;
    const float *arow = vok ? G.a[v] + (long long)min(row0 + i, M - 1) * G.lda[v] : nullptr;
    const float *pp = (vok && nok) ? (v == 0 ? psel0 : psel1) : nullptr;
#pragma unroll
    for (int c = 0; c < 4; ++c) aval[v][c] = (vok && 4 * kq + c < A) ? arow[4 * kq + c] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = row0 + 4 * kq + r;
      pval[v][r] = (pp && row < M) ? pp[(long long)row * Q + q] : 0.f;
    }
    if (a.sum_planes && pp) {   // (uniform flag) the other planes, in order, eight requests in flight per row
      const long long ps = (long long)M * Q * (a.plane_step > 1 ? a.plane_step : 1);
      for (int p0 = 1; p0 < a.planes; p0 += 8) {
        float x[8][4];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int pl = p0 + u;
          if (pl < a.planes) {   // (uniform: two planes when the layers' launches sum their own)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int row = min(row0 + 4 * kq + r, M - 1);
              x[u][r] = pp[pl * ps + (long long)row * Q + q];
            }
          }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (p0 + u < a.planes) {
#pragma unroll
            for (int r = 0; r < 4; ++r) pval[v][r] += x[u][r];
          }
      }
    }
  }
  if (nok) bq = bsel[q];
  hf_v4f acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};   // even / odd 16-k blocks: two independent MFMA chains
  // (the weight rows need no 16-byte alignment - a head over cat(h1, h0, s, a) has a row pitch of 774 floats, so every odd row
  // starts 8 bytes off: gfx950 serves dwordx4 loads from 4-byte aligned addresses; requiring it sent half the lanes down the
  // element-wise path below, 16 dependent round trips, 27 us for this kernel)
  const bool vec = (L & 15) == 0 && (G.lds & 3) == 0 && (reinterpret_cast<uintptr_t>(G.s) & 15) == 0;   // wave-uniform
  if (vec) {   // whole 16-k blocks: one dwordx4 per operand and block, sixteen blocks in flight
    typedef const __attribute__((address_space(1))) hf_v4f *gq;
    gq sp = (gq)(srow + 4 * kq), wp = (gq)(wrow + 4 * kq);
    for (int k0 = 0; k0 < L; k0 += 256) {
      hf_v4f av[16], bv[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int blk = min((k0 >> 4) + u, (L >> 4) - 1);
        av[u] = sp[4 * blk];
        bv[u] = nok ? wp[4 * blk] : hf_v4f{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < 16; u += 2) {
        if (k0 + 16 * u < L) {   // uniform
#pragma unroll
          for (int c = 0; c < 4; ++c) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][c], bv[u][c], acc, 0, 0, 0);
        }
        if (k0 + 16 * (u + 1) < L) {
#pragma unroll
          for (int c = 0; c < 4; ++c) acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u + 1][c], bv[u + 1][c], acc2, 0, 0, 0);
        }
      }
    }
  } else {
    for (int k0 = 0; k0 < L; k0 += 16) {
      const int kb = k0 + 4 * kq;
      float av[4], bv[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        av[c] = kb + c < L ? srow[kb + c] : 0.f;
        bv[c] = (nok && kb + c < L) ? wrow[kb + c] : 0.f;
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c], bv[c], acc, 0, 0, 0);
    }
  }
  acc += acc2;
  for (int v = 0; v < G.nvar; ++v) {
    hf_v4f accv = acc;
#pragma unroll
    for (int c = 0; c < 4; ++c) accv = __builtin_amdgcn_mfma_f32_16x16x4f32(v == 0 ? aval[0][c] : aval[1][c], wact[c], accv, 0, 0, 0);
    if (nok) {
      float *out = G.out[v];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + 4 * kq + r;
        if (row < M) out[(long long)row * G.ldo[v] + set * Q + q] = (accv[r] + (v == 0 ? pval[0][r] : pval[1][r])) + bq;
      }
    }
  }
}
hipError_t head_finish_launch(const HeadFinishArgs &a, hipStream_t s) {
  if (a.M <= 0 || a.ngroups <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_head_finish, dim3((unsigned)((a.M + 15) / 16), (unsigned)a.ngroups), dim3(64), 0, s, a);
  return hipGetLastError();
}

hipError_t colsum_tall_launch(const float *X, long long R, int C, int ld, float *partial, hipStream_t s) {
  if (R <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_colsum_tall, dim3(colsum_tall_blocks(R)), dim3(256), 0, s, X, R, C, ld, partial);
  return hipGetLastError();
}
hipError_t reduce_partials_batched_launch(const float *part, int ninst, int nparts, long long n, float *dst, hipStream_t s) {
  hipLaunchKernelGGL(k_reduce_partials, dim3((unsigned)((n + 63) / 64), ninst), dim3(256), 0, s, part, nparts, n, dst);
  return hipGetLastError();
}
// Trainer-side summaries of the last update, on demand (never in the step): franQ logs them every log_interval steps
// (deepQlearning.py:231-247, distributional_soft_actor_critic.py:65-67):
//   out[0] = mean over rows of the UNBIASED variance of q_pred over its Nq atoms   (q_pred.var(-1).mean())
//   out[1] = mean of is_contiguous, out[2] / out[3] = max / min over the B windows of (sum_t is_contiguous) / temporal_len
//   out[4 + i] = L2 norm of gradient arena range i (i < nranges: one per parameter tensor the caller lists)
// One workgroup: the inputs are a few hundred KB.
__global__ __launch_bounds__(1024) void k_summaries(const float *__restrict__ q_pred, int M, int Nq, const float *__restrict__ ic, int Tm1, int B,
                                                    int temporal_len, const float *__restrict__ grads, const long long *__restrict__ ranges,
                                                    int nranges, float *__restrict__ out) {
  __shared__ float red[1024];
  __shared__ float red2[1024];
  const int t = threadIdx.x;
  auto block_sum = [&](float v) {
    red[t] = v;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
      if (t < s) red[t] += red[t + s];
      __syncthreads();
    }
    const float r = red[0];
    __syncthreads();
    return r;
  };
  float v = 0.f;
  for (int m = t; m < M; m += 1024) {
    float mu = 0.f;
    for (int i = 0; i < Nq; ++i) mu += q_pred[(long long)m * Nq + i];
    mu /= (float)Nq;
    float s2 = 0.f;
    for (int i = 0; i < Nq; ++i) { const float d = q_pred[(long long)m * Nq + i] - mu; s2 += d * d; }
    v += Nq > 1 ? s2 / (float)(Nq - 1) : 0.f;
  }
  const float qvar = block_sum(v) / (float)M;
  float tot = 0.f, mx = -1e30f, mn = 1e30f;
  for (int b = t; b < B; b += 1024) {
    float s = 0.f;
    for (int k = 0; k < Tm1; ++k) s += ic[(long long)k * B + b];
    tot += s;
    mx = fmaxf(mx, s);
    mn = fminf(mn, s);
  }
  const float total = block_sum(tot);
  red[t] = mx; red2[t] = mn;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if (t < s) { red[t] = fmaxf(red[t], red[t + s]); red2[t] = fminf(red2[t], red2[t + s]); }
    __syncthreads();
  }
  if (t == 0) {
    out[0] = qvar;
    out[1] = total / (float)((long long)Tm1 * B);
    out[2] = red[0] / (float)temporal_len;
    out[3] = red2[0] / (float)temporal_len;
  }
  __syncthreads();
  for (int r = 0; r < nranges; ++r) {
    const long long b0 = ranges[2 * r], n = ranges[2 * r + 1];
    float s = 0.f;
    for (long long i = t; i < n; i += 1024) { const float g = grads[b0 + i]; s += g * g; }
    const float ss = block_sum(s);
    if (t == 0) out[4 + r] = sqrtf(ss);
  }
}
hipError_t summaries_launch(const float *q_pred, int M, int Nq, const float *ic, int Tm1, int B, int temporal_len, const float *grads,
                            const long long *ranges_dev, int nranges, float *out_dev, hipStream_t s) {
  hipLaunchKernelGGL(k_summaries, dim3(1), dim3(1024), 0, s, q_pred, M, Nq, ic, Tm1, B, temporal_len, grads, ranges_dev, nranges, out_dev);
  return hipGetLastError();
}

// dst[rows, cols] = sum over nparts partials of the same shape, + column sums of dst per 32-row block (cs [ceil(rows/32), cols]):
// the sum of the per-network shares of an input gradient, and the bias-gradient partials of the layer that produced its input.
// One 1024-thread workgroup per 32-row block: thread -> (row phase t / 64: rows r0 + ph and r0 + ph + 16, float4 column
// group t % 64 + 64 j); every share of both rows is requested before the first add (the round-3 form - 256 threads, 8 rows
// per thread, a share at a time - was 8 x nparts dependent round trips: 16 us for 256 rows).  cols % 4 == 0.
typedef float rp_v4f __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(1024) void k_sum_parts_colsum(const float *__restrict__ part, int nparts, int rows, int cols,
                                                           float *__restrict__ dst, float *__restrict__ cs) {
  __shared__ rp_v4f red[16][64];
  const int g = threadIdx.x & 63, ph = threadIdx.x >> 6;
  const int r0 = blockIdx.x * 32;
  const long long n = (long long)rows * cols;
  for (int c4 = g; c4 * 4 < cols; c4 += 64) {
    rp_v4f colsum = {0.f, 0.f, 0.f, 0.f};
    const int ra = r0 + ph, rb = r0 + ph + 16;
    const long long ea = (long long)(ra < rows ? ra : rows - 1) * cols + 4 * c4, eb = (long long)(rb < rows ? rb : rows - 1) * cols + 4 * c4;
    rp_v4f va = {0.f, 0.f, 0.f, 0.f}, vb = {0.f, 0.f, 0.f, 0.f};
    for (int p0 = 0; p0 < nparts; p0 += 8) {   // shares in rounds of eight, added in index order
      rp_v4f xa[8], xb[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const long long po = (long long)(p0 + u < nparts ? p0 + u : nparts - 1) * n;
        xa[u] = *reinterpret_cast<const rp_v4f *>(part + po + ea);
        xb[u] = *reinterpret_cast<const rp_v4f *>(part + po + eb);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (p0 + u < nparts) { va += xa[u]; vb += xb[u]; }
    }
    if (ra < rows) { *reinterpret_cast<rp_v4f *>(dst + ea) = va; colsum += va; }
    if (rb < rows) { *reinterpret_cast<rp_v4f *>(dst + eb) = vb; colsum += vb; }
    red[ph][g] = colsum;
    __syncthreads();
    if (ph == 0) {
      rp_v4f t = red[0][g];
#pragma unroll
      for (int j = 1; j < 16; ++j) t += red[j][g];
      *reinterpret_cast<rp_v4f *>(cs + (long long)blockIdx.x * cols + 4 * c4) = t;
    }
    __syncthreads();
  }
}
hipError_t sum_parts_colsum_launch(const float *part, int nparts, int rows, int cols, float *dst, float *cs, hipStream_t s) {
  hipLaunchKernelGGL(k_sum_parts_colsum, dim3((unsigned)((rows + 31) / 32)), dim3(1024), 0, s, part, nparts, rows, cols, dst, cs);
  return hipGetLastError();
}
hipError_t reduce_partials_launch(const float *part, int nparts, long long n, float *dst, hipStream_t s) {
  if (nparts >= 32 && (n & 3) == 0 && ((reinterpret_cast<uintptr_t>(part) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0) {
    hipLaunchKernelGGL(k_reduce_partials_wide, dim3((unsigned)((n / 4 + 31) / 32)), dim3(256), 0, s, part, nparts, n, dst);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(k_reduce_partials, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, s, part, nparts, n, dst);
  return hipGetLastError();
}

// ======================================================================================
// GRU joiner: torch.nn.GRU cell (gate order r, z, n) forward / backward for one time step
//   r = s(gi_r + gh_r), z = s(gi_z + gh_z), n = tanh(gi_n + r * gh_n), h = (1 - z) n + z h_prev
// gi = W_ih x + b_ih and gh = W_hh h_prev + b_hh come from the GEMM kernel; sigmoid / tanh through
// double, rounded once (like the policy kernels: the CPU path's vector math is <= 1 ulp).
// ======================================================================================
__device__ __forceinline__ float sigmoid_f(float x) { return (float)(1.0 / (1.0 + exp(-(double)x))); }

__global__ void k_gru_h0(int mode, const float *__restrict__ src, float *__restrict__ h0, int B, int L) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * L) return;
  h0[i] = mode == 0 ? 0.f : (mode == 1 ? src[i] : src[i % L]);
}

__global__ void k_gru_cell_fwd(const float *__restrict__ gi, float *__restrict__ gh, const float *__restrict__ gh_parts,
                               int nparts, const float *__restrict__ b_hh, const float *__restrict__ hprev,
                               float *__restrict__ h, float *__restrict__ hprev_save, int rows, int L) {
#pragma clang fp contract(off)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * L) return;
  const int b = i / L, l = i - b * L;
  const long long g = (long long)b * 3 * L + l;
  float hr, hz, hn;
  if (nparts > 0) {   // K-split recurrent product: fixed-order sum of the partials, bias once
    const long long ps = (long long)rows * 3 * L;
    hr = b_hh[l]; hz = b_hh[L + l]; hn = b_hh[2 * L + l];
    for (int p = 0; p < nparts; ++p) {
      hr += gh_parts[p * ps + g]; hz += gh_parts[p * ps + g + L]; hn += gh_parts[p * ps + g + 2 * L];
    }
    gh[g] = hr; gh[g + L] = hz; gh[g + 2 * L] = hn;
  } else {
    hr = gh[g]; hz = gh[g + L]; hn = gh[g + 2 * L];
  }
  const float r = sigmoid_f(gi[g] + hr);
  const float z = sigmoid_f(gi[g + L] + hz);
  const float n = (float)dm_tanh((double)(gi[g + 2 * L] + r * hn));
  const float hp = hprev[i];
  h[i] = (1.f - z) * n + z * hp;
  if (hprev_save) hprev_save[i] = hp;
}

__global__ void k_gru_cell_bwd(const float *__restrict__ dstate, const float *__restrict__ carry_a,
                               const float *__restrict__ carry_b, int nparts_b, const float *__restrict__ gi,
                               const float *__restrict__ gh, const float *__restrict__ hprev, float *__restrict__ dgi,
                               float *__restrict__ dgh, float *__restrict__ dh_direct, int rows, int L) {
#pragma clang fp contract(off)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * L) return;
  const int b = i / L, l = i - b * L;
  const long long g = (long long)b * 3 * L + l;
  float dh = dstate ? dstate[i] : 0.f;
  if (carry_a) {
    dh += carry_a[i];
    for (int p = 0; p < nparts_b; ++p) dh += carry_b[(long long)p * rows * L + i];
  }
  const float hn = gh[g + 2 * L];
  const float r = sigmoid_f(gi[g] + gh[g]);
  const float z = sigmoid_f(gi[g + L] + gh[g + L]);
  const float n = (float)dm_tanh((double)(gi[g + 2 * L] + r * hn));
  const float hp = hprev[i];
  const float dn_pre = (dh * (1.f - z)) * (1.f - n * n);
  const float dz_pre = (dh * (hp - n)) * (z * (1.f - z));
  const float dr_pre = (dn_pre * hn) * (r * (1.f - r));
  dgi[g] = dr_pre; dgi[g + L] = dz_pre; dgi[g + 2 * L] = dn_pre;
  dgh[g] = dr_pre; dgh[g + L] = dz_pre; dgh[g + 2 * L] = dn_pre * r;
  dh_direct[i] = dh * z;
}

// block = 32 columns x 8 row groups; fixed-order reduction through LDS
__global__ __launch_bounds__(256) void k_gru_dh0(const float *__restrict__ a, const float *__restrict__ b, int nparts_b,
                                                 int B, int L, float *__restrict__ out) {
  __shared__ float red[8][33];
  const int c = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int l = blockIdx.x * 32 + c;
  float s = 0.f;
  if (l < L) {
    for (int r = rg; r < B; r += 8) {
      float v = a[(long long)r * L + l];
      for (int p = 0; p < nparts_b; ++p) v += b[((long long)p * B + r) * L + l];
      s += v;
    }
  }
  red[rg][c] = s;
  __syncthreads();
  if (rg == 0 && l < L) {
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) t += red[j][c];
    out[l] = t;
  }
}

hipError_t gru_h0_launch(int mode, const float *src, float *h0, int B, int L, hipStream_t s) {
  hipLaunchKernelGGL(k_gru_h0, dim3((B * L + 255) / 256), dim3(256), 0, s, mode, src, h0, B, L);
  return hipGetLastError();
}
hipError_t gru_cell_fwd_launch(const float *gi, float *gh, const float *gh_parts, int nparts, const float *b_hh,
                               const float *hprev, float *h, float *hprev_save, int rows, int L, hipStream_t s) {
  hipLaunchKernelGGL(k_gru_cell_fwd, dim3((rows * L + 255) / 256), dim3(256), 0, s, gi, gh, gh_parts, nparts, b_hh, hprev, h,
                     hprev_save, rows, L);
  return hipGetLastError();
}
hipError_t gru_cell_bwd_launch(const float *dstate, const float *carry_a, const float *carry_b, int nparts_b,
                               const float *gi, const float *gh, const float *hprev, float *dgi, float *dgh,
                               float *dh_direct, int rows, int L, hipStream_t s) {
  hipLaunchKernelGGL(k_gru_cell_bwd, dim3((rows * L + 255) / 256), dim3(256), 0, s, dstate, carry_a, carry_b, nparts_b, gi, gh,
                     hprev, dgi, dgh, dh_direct, rows, L);
  return hipGetLastError();
}
hipError_t gru_dh0_launch(const float *a, const float *b, int nparts_b, int B, int L, float *out, hipStream_t s) {
  hipLaunchKernelGGL(k_gru_dh0, dim3((L + 31) / 32), dim3(256), 0, s, a, b, nparts_b, B, L, out);
  return hipGetLastError();
}

// ======================================================================================
// act(): small-batch inference for the env actors (deepQlearning.py:155-187)
//   One launch per Linear layer of encoder -> joiner -> actor.  The batch is a handful of rows
//   (one per env instance), so a layer is a skinny product bound by the latency of streaming
//   its weights: a wave owns two output columns, its lanes stride over K (coalesced weight
//   rows), ACT_RCH batch rows share every weight load, and N/8 workgroups spread the weight
//   matrix over the chip.  cat(x, h_1..h_n) of the skip head is read as K-segments.
// ======================================================================================
constexpr int ACT_RCH = 8;
constexpr int ACT_KT = 1024;   // K-tile of the input rows staged in LDS (8 x 1024 floats = 32 KB)

constexpr int ACT_PRE_MAXN = 256, ACT_PRE_MAXK = 64;
// WAVES waves per workgroup, two output columns per wave.  POLICY: the launch is one workgroup per 8 rows that holds ALL N <= 32
// outputs (the policy logits), leaves them in LDS and returns them to the caller (k_act_head_policy) instead of storing them.
template <int WAVES, bool POLICY>
__device__ __forceinline__ void act_layer_body(const ActLayerArgs &a, float *xs, float *hpre, float *logits_lds, int logits_pitch) {
  constexpr int THREADS = 64 * WAVES;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n0 = (blockIdx.x * WAVES + wave) * 2;
  const bool live = n0 < a.N;  // wave-uniform; dead waves still help staging and hit the barriers
  const bool has1 = n0 + 1 < a.N;
  const int r0 = blockIdx.y * ACT_RCH;
  const int nr = min(ACT_RCH, a.rows - r0);
  if (a.pre_N > 0) {   // the pre-layer, recomputed by every workgroup: thread -> one of its <= 256 outputs, <= 64 input columns
    int kp = 0;
    for (int s = 0; s < a.pre_nseg; ++s) {
      const ActSeg sg = a.in[s];
      for (int idx = threadIdx.x; idx < ACT_RCH * sg.width; idx += THREADS) {
        const int r = idx / sg.width, k = idx - r * sg.width;
        xs[r * ACT_KT + kp + k] = r < nr ? sg.ptr[(long long)(r0 + r) * sg.ld + k] : 0.f;
      }
      kp += sg.width;
    }
    __syncthreads();
    for (int n = threadIdx.x; n < a.pre_N; n += THREADS) {
      float acc[ACT_RCH];
#pragma unroll
      for (int r = 0; r < ACT_RCH; ++r) acc[r] = 0.f;
      const float *wr = a.pre_W + (long long)n * a.pre_ldw;
      for (int k = 0; k < kp; ++k) {
        const float w = wr[k];
#pragma unroll
        for (int r = 0; r < ACT_RCH; ++r) acc[r] = fmaf(xs[r * ACT_KT + k], w, acc[r]);
      }
      const float bn = a.pre_bias[n];
#pragma unroll
      for (int r = 0; r < ACT_RCH; ++r) {
        const float y = acc[r] + bn;
        hpre[r * ACT_PRE_MAXN + n] = y > 0.f ? y : 0.01f * y;
      }
    }
  }
  float acc0[ACT_RCH], acc1[ACT_RCH];
#pragma unroll
  for (int r = 0; r < ACT_RCH; ++r) acc0[r] = acc1[r] = 0.f;
  const float *w0 = a.W + (long long)(live ? n0 : 0) * a.ldw;
  const float *w1 = has1 ? w0 + a.ldw : w0;
  int ktot = 0;
  for (int s = 0; s < a.nseg; ++s) ktot += a.in[s].width;
  for (int kt0 = 0; kt0 < ktot; kt0 += ACT_KT) {
    const int kt = min(ACT_KT, ktot - kt0);
    __syncthreads();
    // stage rows r0..r0+7 of cat(segments)[kt0 : kt0+kt] (rows past the batch as zeros)
    int soff = 0;
    for (int s = 0; s < a.nseg; ++s) {
      const ActSeg sg = a.in[s];
      const int lo = max(soff, kt0), hi = min(soff + sg.width, kt0 + kt);
      const int w = hi - lo;
      if (w > 0) {
        for (int idx = threadIdx.x; idx < ACT_RCH * w; idx += THREADS) {
          const int r = idx / w, k = idx - r * w;
          float v = 0.f;
          if (r < nr) v = sg.ptr ? sg.ptr[(long long)(r0 + r) * sg.ld + (lo - soff) + k] : hpre[r * ACT_PRE_MAXN + (lo - soff) + k];
          xs[r * ACT_KT + (lo - kt0) + k] = v;
        }
      }
      soff += sg.width;
    }
    __syncthreads();
    if (live) {
#pragma unroll 4
      for (int k = lane; k < kt; k += 64) {
        const float x0 = w0[kt0 + k], x1 = w1[kt0 + k];
#pragma unroll
        for (int r = 0; r < ACT_RCH; ++r) {
          const float v = xs[r * ACT_KT + k];
          acc0[r] = fmaf(v, x0, acc0[r]);
          acc1[r] = fmaf(v, x1, acc1[r]);
        }
      }
    }
  }
  if (!live) return;   // (the caller's barriers come after every wave is back)
#pragma unroll
  for (int r = 0; r < ACT_RCH; ++r) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      acc0[r] += __shfl_xor(acc0[r], off);
      acc1[r] += __shfl_xor(acc1[r], off);
    }
  }
  if (lane == 0) {
    const float b0 = a.bias[n0], b1 = has1 ? a.bias[n0 + 1] : 0.f;
#pragma unroll
    for (int r = 0; r < ACT_RCH; ++r) {
      if (r < nr) {
        float y0 = acc0[r] + b0, y1 = acc1[r] + b1;
        if (a.leaky) {
          y0 = y0 > 0.f ? y0 : 0.01f * y0;
          y1 = y1 > 0.f ? y1 : 0.01f * y1;
        }
        if constexpr (POLICY) {
          logits_lds[r * logits_pitch + n0] = y0;
          if (has1) logits_lds[r * logits_pitch + n0 + 1] = y1;
        } else {
          float *o = a.out + (long long)(r0 + r) * a.ldo + n0;
          o[0] = y0;
          if (has1) o[1] = y1;
        }
      }
    }
  }
}

__global__ __launch_bounds__(256) void k_act_layer(ActLayerArgs a) {
  __shared__ float xs[ACT_RCH * ACT_KT];
  __shared__ float hpre[ACT_RCH * ACT_PRE_MAXN];
  act_layer_body<4, false>(a, xs, hpre, nullptr, 0);
}

// explore / exploit actions, log-prob of the explored one, and the exploit_mask select
// (gaussian_mlp.py:15-39 or gumbel_mlp.py:7-54, then deepQlearning.py:175-180)
// (row m; `lrow`: its logits, in global memory or LDS)
// `vals`: 5 x GUMBEL_MAXN floats of this thread, element (k, j) at vals[(k * GUMBEL_MAXN + j) * stride] (discrete actor only)
// MODE: 1 discrete only, 0 continuous only (k_act_policy: one instantiation each keeps the argument block's live ranges
// inside the SGPR file)
template <int MODE>
__device__ __forceinline__ void act_policy_row(const ActPolicyArgs &a, int m, const float *lrow, float *vals, int stride) {
#pragma clang fp contract(off)
  const bool use_exploit = a.exploit_mask && a.exploit_mask[m] != 0;
  const int A = a.A;
  if (MODE == 1) {
    auto arr = [&](int k) { return LaneArr{vals + k * GUMBEL_MAXN * stride, stride}; };
    const LaneArr lo = arr(0), u = arr(1), norm = arr(2), relaxed = arr(3), sc = arr(4);
    int greedy = 0;
    float greedy_v = -INFINITY;
    for (int j = 0; j < A; ++j) {
      const float l = lrow[j];
      lo[j] = l;
      u[j] = a.noise ? a.noise[(long long)m * A + j]
                     : device_noise(a.seed, (uint32_t)a.counter, 7u, (uint32_t)(m * A + j), false);
      if (j == 0 || l > greedy_v) { greedy_v = l; greedy = j; }
    }
    const int best = gumbel_forward(lo, u, A, norm, relaxed, sc);
    const float lse3 = row_logsumexp(norm, A);
    float logp = 0.f;
    int best_st = 0;
    float best_v = -INFINITY;
    for (int j = 0; j < A; ++j) {
      const float hard = j == best ? 1.f : 0.f;
      const float rj = relaxed[j];
      const float stv = (hard - rj) + rj;
      if (stv > best_v) { best_v = stv; best_st = j; }   // argmax of the straight-through sample
      logp += -stv * (norm[j] - lse3);
    }
    if (a.log_prob) a.log_prob[m] = -logp;
    if (a.explore) a.explore[m] = (float)best_st;
    if (a.exploit) a.exploit[m] = (float)greedy;
    a.action[m] = (float)(use_exploit ? greedy : best_st);
    return;
  }
  const float *lo = lrow;
  float logp = 0.f;
  for (int j = 0; j < A; ++j) {
    const float mean = lo[j];
    const float ls = fminf(fmaxf(lo[A + j], -20.f), 2.f);
    const float sd = (float)dm_exp((double)ls);
    const float eps = a.noise ? a.noise[(long long)m * A + j]
                              : device_noise(a.seed, (uint32_t)a.counter, 7u, (uint32_t)(m * A + j), true);
    const float x = mean + eps * sd;
    const float d = x - mean;
    float lp = -(d * d) / (2.f * (sd * sd)) - (float)dm_log((double)sd) - 0.91893853320467274178f;
    const float ex = (float)dm_tanh((double)x);
    lp -= (float)dm_log((double)((1.f - ex * ex) + 1e-4f));
    logp += lp;
    const float gr = (float)dm_tanh((double)mean);
    if (a.explore) a.explore[(long long)m * A + j] = ex;
    if (a.exploit) a.exploit[(long long)m * A + j] = gr;
    a.action[(long long)m * A + j] = use_exploit ? gr : ex;
  }
  if (a.log_prob) a.log_prob[m] = logp;
}

template <int MODE>
__global__ __launch_bounds__(64) void k_act_policy(ActPolicyArgs a) {
  __shared__ float lane_vals[MODE ? 5 * GUMBEL_MAXN * 64 : 1];
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= a.rows) return;
  act_policy_row<MODE>(a, m, a.logits + (long long)m * a.ld, lane_vals + (MODE ? threadIdx.x : 0), 64);
}

hipError_t act_layer_launch(const ActLayerArgs &a, hipStream_t s) {
  if (a.rows <= 0 || a.N <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_act_layer, dim3((a.N + 7) / 8, (a.rows + ACT_RCH - 1) / ACT_RCH), dim3(256), 0, s, a);
  return hipGetLastError();
}

// Continuous actor, one thread per (row, action) like k_policy_fwd (AG lanes per row, the row's log-prob summed in the order
// j = 0, 1, ... of the sequential loop in act_policy_row): a handful of rows x 6 actions are one short instruction stream per
// lane instead of six in sequence, and nothing of the double-precision math is hoisted into (spilled) SGPRs.
// (m, j): this thread's row and action; lo: the row's logits (global memory or LDS); every lane of the wave calls it.
__device__ __forceinline__ void act_gauss_elem(const ActPolicyArgs &a, int m, int j, bool live, const float *lo) {
#pragma clang fp contract(off)
  const int A = a.A;
  float lp = 0.f;
  if (live) {
    const bool use_exploit = a.exploit_mask && a.exploit_mask[m] != 0;
    const float mean = lo[j];
    const float ls = fminf(fmaxf(lo[A + j], -20.f), 2.f);
    const float sd = (float)dm_exp((double)ls);
    const float eps = a.noise ? a.noise[(long long)m * A + j]
                              : device_noise(a.seed, (uint32_t)a.counter, 7u, (uint32_t)(m * A + j), true);
    const float x = mean + eps * sd;
    const float d = x - mean;
    lp = -(d * d) / (2.f * (sd * sd)) - (float)dm_log((double)sd) - 0.91893853320467274178f;
    const float ex = (float)dm_tanh((double)x);
    lp -= (float)dm_log((double)((1.f - ex * ex) + 1e-4f));
    const float gr = (float)dm_tanh((double)mean);
    if (a.explore) a.explore[(long long)m * A + j] = ex;
    if (a.exploit) a.exploit[(long long)m * A + j] = gr;
    a.action[(long long)m * A + j] = use_exploit ? gr : ex;
  }
  const int base = (threadIdx.x & 63) - j;
  float logp = 0.f;
  for (int k = 0; k < A; ++k) logp += __shfl(lp, base + k, 64);
  if (live && j == 0 && a.log_prob) a.log_prob[m] = logp;
}
template <int AG>
__global__ __launch_bounds__(64) void k_act_policy_gauss(ActPolicyArgs a) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int m = gid / AG, j = gid - m * AG;
  const bool live = m < a.rows && j < a.A;
  act_gauss_elem(a, m, j, live, a.logits + (long long)(live ? m : 0) * a.ld);
}

// The policy's last layer + the policy head in one launch (update_kernels.h): 16 waves x 2 columns hold all N <= 32 logits of
// the workgroup's 8 rows; they stay in LDS and the sample / log-prob / select follows behind one barrier.
constexpr int ACT_HP_PITCH = 36;
__global__ __launch_bounds__(1024) void k_act_head_policy(ActLayerArgs l, ActPolicyArgs p) {
  __shared__ float xs[ACT_RCH * ACT_KT];
  __shared__ float hpre[1];
  __shared__ float lg[ACT_RCH * ACT_HP_PITCH];
  __shared__ float lane_vals[5 * GUMBEL_MAXN * ACT_RCH];
  act_layer_body<16, true>(l, xs, hpre, lg, ACT_HP_PITCH);
  __syncthreads();
  const int r0 = blockIdx.y * ACT_RCH, t = threadIdx.x;
  if (p.discrete) {
    if (t < ACT_RCH && r0 + t < p.rows) act_policy_row<1>(p, r0 + t, lg + t * ACT_HP_PITCH, lane_vals + t, ACT_RCH);
  } else if (t < ACT_RCH * 16) {   // (row, action) = (t / 16, t % 16): A <= 16
    const int r = t >> 4, j = t & 15;
    act_gauss_elem(p, r0 + r, j, r0 + r < p.rows && j < p.A, lg + r * ACT_HP_PITCH);
  }
}
bool act_head_policy_takes(const ActLayerArgs &l, const ActPolicyArgs &p) {
  return l.N <= 32 && l.pre_N == 0 && !l.leaky && (p.discrete ? p.A <= GUMBEL_MAXN && l.N == p.A : p.A <= 16 && l.N == 2 * p.A);
}
hipError_t act_head_policy_launch(const ActLayerArgs &l, const ActPolicyArgs &p, hipStream_t s) {
  if (l.rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_act_head_policy, dim3(1, (l.rows + ACT_RCH - 1) / ACT_RCH), dim3(1024), 0, s, l, p);
  return hipGetLastError();
}

hipError_t act_policy_launch(const ActPolicyArgs &a, hipStream_t s) {
  if (a.rows <= 0) return hipSuccess;
  if (a.discrete) hipLaunchKernelGGL(k_act_policy<1>, dim3((a.rows + 63) / 64), dim3(64), 0, s, a);
  else if (a.A <= 8) hipLaunchKernelGGL(k_act_policy_gauss<8>, dim3((a.rows * 8 + 63) / 64), dim3(64), 0, s, a);
  else if (a.A <= 32) hipLaunchKernelGGL(k_act_policy_gauss<32>, dim3((a.rows * 32 + 63) / 64), dim3(64), 0, s, a);
  else if (a.A <= 64) hipLaunchKernelGGL(k_act_policy_gauss<64>, dim3(a.rows), dim3(64), 0, s, a);
  else return hipErrorInvalidValue;   // (act_dim <= 64 is checked at fdql_agent_create)
  return hipGetLastError();
}

// stand-in for a collective's channel kernels (include/fdql.h, fdql_debug_side_copy)
__global__ __launch_bounds__(256) void k_side_copy(const float4 *__restrict__ src, float4 *__restrict__ dst, long long n4, int passes,
                                                   long long hold_ticks) {
  extern __shared__ float side_lds[];
  const unsigned long long t0 = wall_clock64();       // constant 100 MHz counter
  if (side_lds && threadIdx.x == 0 && hold_ticks < 0) side_lds[0] = 0.f;   // (never: keeps the allocation referenced)
  for (int p = 0; p < passes; ++p)
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
      float4 v = src[i];
      v.x += (float)p;   // (keeps the passes from being folded into one)
      dst[i] = v;
    }
  // a collective's workgroups mostly WAIT (for their peers, for the links) while they hold their CU: stay resident until
  // hold_ticks have passed since the start.  The clock runs on its own, so every wave reaches the exit.
  while ((long long)(wall_clock64() - t0) < hold_ticks) __builtin_amdgcn_s_sleep(64);
}

}  // namespace fdql

extern "C" int fdql_debug_side_copy(const float *src, float *dst, int64_t n, int32_t workgroups, int32_t passes, int32_t hold_us,
                                    int32_t lds_bytes, void *stream) {
  using namespace fdql;
  FDQL_REQUIRE(src && dst && n >= 0 && n % 4 == 0 && workgroups >= 1 && workgroups <= 4096 && passes >= 0, "bad side-copy arguments");
  FDQL_REQUIRE(hold_us >= 0 && hold_us <= 5000 && lds_bytes >= 0 && lds_bytes <= 65536, "side copy: hold_us in [0, 5000], lds_bytes in [0, 64 KiB]");
  FDQL_REQUIRE((reinterpret_cast<uintptr_t>(src) & 15) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0, "side copy needs 16-byte aligned buffers");
  hipLaunchKernelGGL(k_side_copy, dim3(workgroups), dim3(256), lds_bytes, (hipStream_t)stream, (const float4 *)src, (float4 *)dst,
                     (long long)(n / 4), passes, (long long)hold_us * 100);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("side copy: %s", hipGetErrorString(e)); return FDQL_EHIP; }
  return 0;
}
