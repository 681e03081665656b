// Row-block dgrad kernel (rowdgrad.h): single-network input gradients on 64-row blocks with K-strided weights streamed
// straight into MFMA operands.
//
// Why not the tile kernel: joiner.dpre0 / d enc / enc_obs.dpre0 are [12 544 x 256] x [256 x 256] products - 784 tiles of
// 64 x 64, one wave per SIMD, 16 dependent K iterations each: 47-60 TFLOP/s.  Why not the
// weight-stationary kernels: one network = one instance, and 256 workgroups each loading the same 256 KB of weights spend
// half their life in the prologue.  Here a workgroup takes 64 rows x all 256 columns:
//   * the A tiles (one or two segments of 256 k) go global -> LDS once, by LDS-DMA;
//   * weights are K-strided (element (k, n) at W[k*ldw + n]: a dgrad multiplies by the layer's weight matrix as stored).  A
//     lane's 16-byte load W[k][n0 + 4 j .. + 3] (j = lane % 16, k = kb + 8 (lane / 16) + i) is, component c, the B operand
//     (lane (j, kq) = B[k = kq][n = j]) of a v_mfma_f32_16x16x4_f32 whose 16 output columns are n0 + 4 j + c: four
//     accumulators per 16-row tile cover a wave's 64 columns and no transpose is needed anywhere; the matching A operand
//     (lane (row, kq) = x[row][kb + 8 kq + i]) is a component of two ds_read_b128.  8 global loads + 8 LDS reads per
//     128 MFMAs (4096 MFMA cycles) per wave;
//   * a lane ends up with four consecutive columns of a row per (tile, register): the LeakyReLU' gate reads the reference
//     activation as float4, the result leaves as global_store_dwordx4, and the column sums of the block (bias gradients)
//     are 16 in-lane adds + two cross-lane steps per column quad.
// Measured at config 2 (12 544 rows, HIP events around the launch): 27-28 us per 256 x 256 layer and 43 us for the
// two-segment one against 35 / 55 us on the tile kernel (1.303 -> 1.280 ms per update).  With the weight requests left to
// the scheduler (sunk next to their uses, each waited for in turn) it was 32 / 53 us.
#include "rowdgrad.h"

#include <stdlib.h>
#include <string.h>
#include <mutex>

namespace fdql {
namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) float *gcf;
typedef const __attribute__((address_space(1))) v4f *gcf4;
typedef __attribute__((address_space(1))) v4f *gf4;
typedef __attribute__((address_space(3))) void *lds_vp;
typedef const __attribute__((address_space(1))) void *glb_vp;

constexpr int P = RD_K + 4;            // image row pitch: (P / 4) odd
constexpr int RD_RT = RD_BM / 16;      // 16-row MFMA tiles per workgroup
constexpr int IMG = RD_BM * P;

__device__ __forceinline__ int rd_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// SUM: segment 0's rows are the sum of a.nsum shares (RowDgradArgs::sum_*): staged through registers instead of by LDS-DMA
template <int NSEG, bool GATE, bool SUM = false>
__global__ __launch_bounds__(256, 1) void k_rowdgrad(const RowDgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // A images [NSEG][64][P] (SUM: + [4][256] column sums of the sum)
  const int tid = threadIdx.x, lane = tid & 63, wave = rd_uni(tid >> 6);
  const int j = lane & 15, kq = lane >> 4;
  const int blk = blockIdx.x, r0 = blk * RD_BM, n0 = wave * 64;

  // ---- A tiles -> LDS (wave w: rows w, w + 4, ...: one 1 KiB row per instruction)
#pragma unroll
  for (int s = SUM ? 1 : 0; s < NSEG; ++s) {
    const float *src = a.A[s] + (long long)r0 * RD_K;
#pragma unroll
    for (int u = 0; u < RD_BM / 4; ++u) {
      const int row = wave + 4 * u;
      __builtin_amdgcn_global_load_lds((glb_vp)(src + row * RD_K + lane * 4), (lds_vp)(lds + s * IMG + row * P), 16, 0, 0);
    }
  }
  if constexpr (SUM) {   // the shares of a row added in index order (as k_sum_parts_colsum did)
    const int nsum = rd_uni(a.nsum);
    const long long ps = a.sum_stride;
    gcf part = (gcf)a.sum_parts + (long long)r0 * RD_K + lane * 4;
    v4f csum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int u0 = 0; u0 < RD_BM / 4; u0 += 8) {   // eight rows x nsum shares per request round
      v4f v[8][8];
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        if (p < nsum) {   // (uniform)
#pragma unroll
          for (int uu = 0; uu < 8; ++uu) v[uu][p] = *(gcf4)(part + (long long)p * ps + (wave + 4 * (u0 + uu)) * RD_K);
        }
      }
#pragma unroll
      for (int uu = 0; uu < 8; ++uu) {
        const int row = wave + 4 * (u0 + uu);
        v4f t = v[uu][0];
#pragma unroll
        for (int p = 1; p < 8; ++p)
          if (p < nsum) t += v[uu][p];   // (uniform)
        *reinterpret_cast<v4f *>(lds + row * P + lane * 4) = t;
        *(gf4)(a.sum_out + (long long)(r0 + row) * RD_K + lane * 4) = t;
        csum += t;
      }
    }
    *reinterpret_cast<v4f *>(lds + NSEG * IMG + wave * RD_K + lane * 4) = csum;
  }

  v4f acc[RD_RT][4];   // [16-row tile][c]: register r = row 16 rt + 4 kq + r, column n0 + 4 j + c
#pragma unroll
  for (int rt = 0; rt < RD_RT; ++rt)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[rt][c] = v4f{0.f, 0.f, 0.f, 0.f};

  // weight fragments of a 32-k group: load i = row kb + 8 kq + i
  auto load_w = [&](gcf w, int ldw, int kb, v4f (&wv)[8]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 8; ++i) wv[i] = *(gcf4)(w + (long long)(kb + 8 * kq + i) * ldw);
  };
  v4f wv[2][8];
  {
    gcf w0 = (gcf)a.W[0] + n0 + 4 * j;
    load_w(w0, a.ldw[0], 0, wv[0]);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the DMA pieces have landed (and the first weight fragments with them)
  __syncthreads();
  if constexpr (SUM) {   // the block's column sums of the summed rows: the four waves' partial rows, in order
    const float *cw = lds + NSEG * IMG;
    a.sum_colsum[(long long)blk * RD_K + tid] = ((cw[tid] + cw[RD_K + tid]) + cw[2 * RD_K + tid]) + cw[3 * RD_K + tid];
  }

  v4f rq[GATE ? 4 * RD_RT : 1];   // the gate's reference quads: requested under the last group's MFMAs
#pragma unroll
  for (int s = 0; s < NSEG; ++s) {
    gcf w = (gcf)a.W[s] + n0 + 4 * j;
    const int ldw = a.ldw[s];
    const float *img = lds + s * IMG + j * P + 8 * kq;
#pragma unroll
    for (int g = 0; g < RD_K / 32; ++g) {
      const int cur = (s * (RD_K / 32) + g) & 1;
      // next group's weights (of this segment, or the first of the next one)
      if (g + 1 < RD_K / 32) {
        load_w(w, ldw, 32 * (g + 1), wv[cur ^ 1]);
      } else if (s + 1 < NSEG) {
        load_w((gcf)a.W[s + 1] + n0 + 4 * j, a.ldw[s + 1], 0, wv[cur ^ 1]);
      } else if constexpr (GATE) {
#pragma unroll
        for (int rt = 0; rt < RD_RT; ++rt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            rq[4 * rt + r] = *(gcf4)((gcf)a.ref + (long long)(r0 + 16 * rt + 4 * kq + r) * RD_N + n0 + 4 * j);
      }
      // (the empty asm pins the requests ahead of this group's MFMAs: the scheduler otherwise sinks every load next to its
      // use, one group later, and waits for each one in turn)
      asm volatile("" ::: "memory");
      v4f xa[RD_RT], xb[RD_RT];
#pragma unroll
      for (int rt = 0; rt < RD_RT; ++rt) {
        xa[rt] = *reinterpret_cast<const v4f *>(img + 16 * rt * P + 32 * g);
        xb[rt] = *reinterpret_cast<const v4f *>(img + 16 * rt * P + 32 * g + 4);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int rt = 0; rt < RD_RT; ++rt) {
          const float av = i < 4 ? xa[rt][i] : xb[rt][i - 4];
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[rt][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, wv[cur][i][c], acc[rt][c], 0, 0, 0);
        }
    }
  }

  // ---- epilogue: gate, store, column sums
  v4f cs = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int rt = 0; rt < RD_RT; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v4f x = {acc[rt][0][r], acc[rt][1][r], acc[rt][2][r], acc[rt][3][r]};
      if constexpr (GATE) {
        const v4f h = rq[4 * rt + r];
#pragma unroll
        for (int c = 0; c < 4; ++c) x[c] = h[c] > 0.f ? x[c] : 0.01f * x[c];
      }
      *(gf4)(a.C + (long long)(r0 + 16 * rt + 4 * kq + r) * RD_N + n0 + 4 * j) = x;
      cs += x;
    }
  if (a.colsum) {   // rows 4 kq + .. of every tile are in this lane; the other three quarters sit 16, 32, 48 lanes away
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float v = cs[c];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      cs[c] = v;
    }
    if (kq == 0) *(gf4)(a.colsum + (long long)blk * RD_N + n0 + 4 * j) = cs;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// k_rowdgrad_chain<RT> (rowdgrad.h, RowChainArgs): k_rowdgrad's K loop three times over the same 16 RT rows, the layers' inputs in
// two LDS images: image 0 = x0 (the summed shares), image 1 = x1; x2 then replaces x0 in image 0.  The first weight fragments of
// the next layer are requested before a layer's epilogue.  RT = 4 (64-row blocks) is the form of round 4; RT = 2 / 1 (round 6)
// are one rank's share of a data-parallel batch and temporal_len 2: the same four weight matrices streamed per workgroup for
// fewer rows, so twice / four times the workgroups fill the chip where 64-row blocks are a fraction of a dispatch round, and two
// workgroups share a CU (the images are 33 / 66 KB) so that one's weight requests fly under the other's MFMAs.
template <int RT>
__global__ __launch_bounds__(256, RT <= 2 ? 2 : 1) void k_rowdgrad_chain(const RowChainArgs a) {
  constexpr int BM = 16 * RT, IMGB = BM * P, RW = BM / 4, RB = RT <= 2 ? 4 : 8;   // rows per wave / per request round of the sum (RB x 8 quads in flight: 128 registers under the two-workgroup budget)
  extern __shared__ __attribute__((aligned(16))) float lds[];   // images [2][BM][P], then [4][256] column sums of x0
  const int tid = threadIdx.x, lane = tid & 63, wave = rd_uni(tid >> 6);
  const int j = lane & 15, kq = lane >> 4;
  const int blk = blockIdx.x, r0 = blk * BM, n0 = wave * 64;
  float *const img0 = lds, *const img1 = lds + IMGB, *const csw = lds + 2 * IMGB;

  v4f acc[RT][4];
  v4f wv[2][8];
  auto load_w = [&](gcf w, int ldw, int kb, v4f (&d)[8]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 8; ++i) d[i] = *(gcf4)(w + (long long)(kb + 8 * kq + i) * ldw);
  };
  load_w((gcf)a.W1 + n0 + 4 * j, a.ldw1, 0, wv[0]);   // layer 1's first fragments: under the staging of the shares
  {   // ---- x0 = sum of the shares (as k_rowdgrad<.., SUM>)
    const int nsum = rd_uni(a.nsum);
    const long long ps = a.sum_stride;
    gcf part = (gcf)a.sum_parts + (long long)r0 * RD_K + lane * 4;
    v4f csum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int u0 = 0; u0 < RW; u0 += RB) {
      v4f v[RB][8];
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        if (p < nsum) {   // (uniform)
#pragma unroll
          for (int uu = 0; uu < RB; ++uu) v[uu][p] = *(gcf4)(part + (long long)p * ps + (wave + 4 * (u0 + uu)) * RD_K);
        }
      }
#pragma unroll
      for (int uu = 0; uu < RB; ++uu) {
        const int row = wave + 4 * (u0 + uu);
        v4f t = v[uu][0];
#pragma unroll
        for (int p = 1; p < 8; ++p)
          if (p < nsum) t += v[uu][p];
        *reinterpret_cast<v4f *>(img0 + row * P + lane * 4) = t;
        *(gf4)(a.x0 + (long long)(r0 + row) * RD_K + lane * 4) = t;
        csum += t;
      }
    }
    *reinterpret_cast<v4f *>(csw + wave * RD_K + lane * 4) = csum;
  }
  __syncthreads();
  a.cs0[(long long)blk * RD_K + tid] = ((csw[tid] + csw[RD_K + tid]) + csw[2 * RD_K + tid]) + csw[3 * RD_K + tid];

  // one 256-k segment: acc += image x W (K-strided weights streamed as in k_rowdgrad); the segment's first fragments are in
  // wv[cur0]; `next`: the first fragments of what follows (next segment / next layer) are requested under the last group
  auto segment = [&](const float *image, gcf w, int ldw, int cur0, gcf wnext, int ldwnext, bool has_next) __attribute__((always_inline)) {
    const float *img = image + j * P + 8 * kq;
#pragma unroll
    for (int g = 0; g < RD_K / 32; ++g) {
      const int cur = (cur0 + g) & 1;
      if (g + 1 < RD_K / 32) load_w(w, ldw, 32 * (g + 1), wv[cur ^ 1]);
      else if (has_next) load_w(wnext, ldwnext, 0, wv[cur ^ 1]);
      asm volatile("" ::: "memory");
      v4f xa[RT], xb[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        xa[rt] = *reinterpret_cast<const v4f *>(img + 16 * rt * P + 32 * g);
        xb[rt] = *reinterpret_cast<const v4f *>(img + 16 * rt * P + 32 * g + 4);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const float av = i < 4 ? xa[rt][i] : xb[rt][i - 4];
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[rt][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, wv[cur][i][c], acc[rt][c], 0, 0, 0);
        }
    }
  };
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[rt][c] = v4f{0.f, 0.f, 0.f, 0.f};
  };
  // gate (ref != null), store, column sums, and the tile into an LDS image for the next layer (dst != null)
  auto epilogue = [&](const float *ref, float *C, float *colsum, float *dst) __attribute__((always_inline)) {
    v4f cs = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      v4f h[4];
      if (ref) {   // (uniform) the tile's four reference quads together
#pragma unroll
        for (int r = 0; r < 4; ++r) h[r] = *(gcf4)((gcf)ref + (long long)(r0 + 16 * rt + 4 * kq + r) * RD_N + n0 + 4 * j);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v4f x = {acc[rt][0][r], acc[rt][1][r], acc[rt][2][r], acc[rt][3][r]};
        if (ref) {
#pragma unroll
          for (int c = 0; c < 4; ++c) x[c] = h[r][c] > 0.f ? x[c] : 0.01f * x[c];
        }
        const int row = 16 * rt + 4 * kq + r;
        *(gf4)(C + (long long)(r0 + row) * RD_N + n0 + 4 * j) = x;
        if (dst) *reinterpret_cast<v4f *>(dst + row * P + n0 + 4 * j) = x;
        cs += x;
      }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float v = cs[c];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      cs[c] = v;
    }
    if (kq == 0) *(gf4)(colsum + (long long)blk * RD_N + n0 + 4 * j) = cs;
  };

  // ---- layer 1: x1 = gate1(x0 W1) -> image 1                                   (wv[0] holds W1's first fragments)
  zero_acc();
  segment(img0, (gcf)a.W1 + n0 + 4 * j, a.ldw1, 0, (gcf)a.W2a + n0 + 4 * j, a.ldw2a, true);
  epilogue(a.ref1, a.x1, a.cs1, img1);
  __syncthreads();
  // ---- layer 2: x2 = x0 W2a + x1 W2b -> image 0 (once every wave is done reading x0)   (8 groups per segment: wv[0] again)
  zero_acc();
  segment(img0, (gcf)a.W2a + n0 + 4 * j, a.ldw2a, 0, (gcf)a.W2b + n0 + 4 * j, a.ldw2b, true);
  segment(img1, (gcf)a.W2b + n0 + 4 * j, a.ldw2b, 0, (gcf)a.W3 + n0 + 4 * j, a.ldw3, true);
  __syncthreads();
  epilogue(nullptr, a.x2, a.cs2, img0);
  __syncthreads();
  // ---- layer 3: x3 = gate3(x2 W3)
  zero_acc();
  segment(img0, (gcf)a.W3 + n0 + 4 * j, a.ldw3, 0, nullptr, 0, false);
  epilogue(a.ref3, a.x3, a.cs3, nullptr);
}

// ---------------------------------------------------------------------------------------------------------------------
// k_rowdot (rowdgrad.h): wave = 16-row tiles of one problem, walked with a stride; v_mfma_f32_16x16x4_f32 with the rows as the
// A operand (lane (row, kq): 16 bytes X[row][16 g + 4 kq ..]: sixteen requests = the tile's 16 KB in flight, two tiles deep)
// and the network's few weight columns as the B operand, held in registers for the wave's life (lane (a, kq):
// W[16 g + 4 kq + c][a]).  Four accumulators (one per c) keep the MFMA chains independent.
__global__ __launch_bounds__(256) void k_rowdot(const RowDotArgs a) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, kq = lane >> 4;
  const int prob = blockIdx.x / a.wgs_per_prob, wg = blockIdx.x - prob * a.wgs_per_prob;
  const int M = a.M, A = a.A, Q = a.Q;
  gcf X = (gcf)a.X[prob], W = (gcf)a.W[prob], D = (gcf)a.D[prob], V = (gcf)a.V[prob];
  const bool aok = j < A;
  float wm[16][4];   // main weights of this lane: [g][c] = W[16 g + 4 kq + c][a = j]
#pragma unroll
  for (int g = 0; g < 16; ++g)
#pragma unroll
    for (int c = 0; c < 4; ++c) wm[g][c] = aok ? W[(long long)(16 * g + 4 * kq + c) * a.ldw + j] : 0.f;
  float wv[8];       // narrow segment: step s multiplies D[row][4 s + kq] by V[4 s + kq][a]
#pragma unroll
  for (int s = 0; s < 8; ++s) wv[s] = (D && aok && 4 * s + kq < Q) ? V[(long long)(4 * s + kq) * a.ldv + j] : 0.f;
  const int nsteps = D ? (Q + 3) >> 2 : 0;
  const int ntiles = (M + 15) >> 4, stride = a.wgs_per_prob * 4;
  auto load = [&](int t, v4f (&x)[16], float (&d)[8]) __attribute__((always_inline)) {
    const int row = min(16 * t + j, M - 1);   // (rows past the batch repeat its last row; never stored)
    gcf xr = X + (long long)row * RD_K + 4 * kq;
#pragma unroll
    for (int g = 0; g < 16; ++g) x[g] = *(gcf4)(xr + 16 * g);
#pragma unroll
    for (int s = 0; s < 8; ++s) d[s] = (s < nsteps) ? D[(long long)row * a.lddy + min(4 * s + kq, Q - 1)] : 0.f;
  };
  auto tile = [&](int t, const v4f (&x)[16], const float (&d)[8]) __attribute__((always_inline)) {
    v4f acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < 16; ++g)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[g][c], wm[g][c], acc[c], 0, 0, 0);
#pragma unroll
    for (int s = 0; s < 8; ++s)
      if (s < nsteps) acc[s & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(4 * s + kq < Q ? d[s] : 0.f, wv[s], acc[s & 3], 0, 0, 0);
    const v4f sum = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    // D[row][a]: lane (a = j, kq) holds rows 4 kq + r
    if (aok) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * t + 4 * kq + r;
        if (row < M) ((__attribute__((address_space(1))) float *)a.out[prob])[(long long)row * a.ldo + j] = sum[r];
      }
    }
  };
  int t = wg * 4 + wave;
  if (t >= ntiles) return;
  v4f xa[16], xb[16];
  float da[8], db[8];
  load(t, xa, da);
  asm volatile("" ::: "memory");
#pragma unroll 1
  for (; t < ntiles; t += 2 * stride) {
    const bool more = t + stride < ntiles;
    load(more ? t + stride : t, xb, db);
    asm volatile("" ::: "memory");
    tile(t, xa, da);
    asm volatile("" ::: "memory");
    if (more) {
      load(t + 2 * stride < ntiles ? t + 2 * stride : t, xa, da);
      asm volatile("" ::: "memory");
      tile(t + stride, xb, db);
      asm volatile("" ::: "memory");
    }
  }
}

}  // namespace

bool rowdot_from_problems(const GemmProblem *probs, int nprob, RowDotArgs &args) {
  if (!plan_switches().rowdgrad) return false;
  if (nprob < 1 || nprob > RDOT_MAX_PROB) return false;
  memset(&args, 0, sizeof(args));
  auto aligned = [](const void *q, uintptr_t n) { return (reinterpret_cast<uintptr_t>(q) & (n - 1)) == 0; };
  const GemmProblem &p0 = probs[0];
  if (p0.N < 1 || p0.N > 16 || p0.M < 16) return false;
  args.M = p0.M; args.A = p0.N; args.nprob = nprob; args.ldo = p0.ldc;
  for (int i = 0; i < nprob; ++i) {
    const GemmProblem &p = probs[i];
    if (p.M != p0.M || p.N != p0.N || p.ldc != p0.ldc || p.nseg < 1 || p.nseg > 2 || p.ksplit != 1 || p.bias || p.colsum || p.C2 || p.hf_w ||
        p.fz_h || p.epi != EPI_NONE || p.nseg != p0.nseg)
      return false;
    const GemmSeg *mainseg = nullptr, *nar = nullptr;
    for (int s = 0; s < p.nseg; ++s) {
      const GemmSeg &sg = p.seg[s];
      if (!sg.a_kc || sg.b_kc) return false;
      if (sg.K == RD_K && !mainseg) mainseg = &sg;
      else if (sg.K >= 1 && sg.K <= 32 && !nar) nar = &sg;
      else return false;
    }
    if (!mainseg || mainseg->lda != RD_K || !aligned(mainseg->A, 16)) return false;
    if (i == 0) { args.ldw = mainseg->ldb; args.Q = nar ? nar->K : 0; args.lddy = nar ? nar->lda : 0; args.ldv = nar ? nar->ldb : 0; }
    if (mainseg->ldb != args.ldw || (nar && (nar->K != args.Q || nar->lda != args.lddy || nar->ldb != args.ldv))) return false;
    args.X[i] = mainseg->A; args.W[i] = mainseg->B; args.D[i] = nar ? nar->A : nullptr; args.V[i] = nar ? nar->B : nullptr; args.out[i] = p.C;
  }
  const int ntiles = (args.M + 15) / 16;
  int wgs = (ntiles + 4 * 4 - 1) / (4 * 4);   // ~4 tiles per wave (measured at config 2: 12 / 4 / 2 tiles per wave = 0.038 / 0.022 / 0.026 ms; tile kernel 0.030)
  if (wgs < 1) wgs = 1;
  args.wgs_per_prob = wgs;
  return true;
}

hipError_t rowdot_launch(const RowDotArgs &a, hipStream_t s) {
  hipLaunchKernelGGL(k_rowdot, dim3(a.nprob * a.wgs_per_prob), dim3(256), 0, s, a);
  return hipGetLastError();
}

bool rowdgrad_from_problem(const GemmProblem &p, RowDgradArgs &args, int bm) {
  if (!plan_switches().rowdgrad) return false;   // FDQL_ROWDGRAD=0
  if ((bm != 16 && bm != 32 && bm != RD_BM) || p.M < bm || p.M % bm || p.N != RD_N || p.nseg < 1 || p.nseg > RD_MAX_SEG || p.ksplit != 1 || p.bias || p.C2 || p.hf_w || p.fz_h ||
      p.ldc != RD_N || (p.epi != EPI_NONE && p.epi != EPI_LRELU_GRAD))
    return false;
  auto aligned = [](const void *q, uintptr_t n) { return (reinterpret_cast<uintptr_t>(q) & (n - 1)) == 0; };
  memset(&args, 0, sizeof(args));
  args.M = p.M; args.nseg = p.nseg; args.gate = p.epi == EPI_LRELU_GRAD;
  for (int s = 0; s < p.nseg; ++s) {
    const GemmSeg &sg = p.seg[s];
    if (sg.K != RD_K || !sg.a_kc || sg.b_kc || sg.lda != RD_K || !aligned(sg.A, 16) || !aligned(sg.B, 4)) return false;
    args.A[s] = sg.A; args.W[s] = sg.B; args.ldw[s] = sg.ldb;
  }
  if (args.gate && (!p.ref || p.ldref != RD_N || !aligned(p.ref, 16))) return false;
  if (!aligned(p.C, 16) || (p.colsum && !aligned(p.colsum, 16))) return false;
  args.ref = p.ref; args.C = p.C; args.colsum = p.colsum;
  return true;
}

// The launch that sums `nsum` shares [M, 256] (stride floats apart) into segment 0's array can be folded into this launch when
// segment 0 is that array: at most 8 shares, 16-byte aligned.
bool rowdgrad_fold_sum(RowDgradArgs &args, const float *parts, int nsum, long long stride, float *sum_out, float *sum_colsum) {
  auto aligned = [](const void *q, uintptr_t n) { return (reinterpret_cast<uintptr_t>(q) & (n - 1)) == 0; };
  if (nsum < 2 || nsum > 8 || args.A[0] != sum_out || !aligned(parts, 16) || (stride & 3) || !aligned(sum_out, 16) || !sum_colsum || args.nseg != 1) return false;
  args.sum_parts = parts; args.nsum = nsum; args.sum_stride = stride; args.sum_out = sum_out; args.sum_colsum = sum_colsum;
  return true;
}

bool rowchain_from_launches(const RowDgradArgs &l1, const RowDgradArgs &l2, const RowDgradArgs &l3, RowChainArgs &c, int bm) {
  if (!plan_switches().rowdgrad_chain) return false;   // FDQL_NO_ROWDGRAD_CHAIN
  if ((bm != 16 && bm != 32 && bm != RD_BM) || l1.M % bm) return false;
  if (l1.nsum < 2 || l1.nseg != 1 || !l1.gate || !l1.colsum || !l1.sum_colsum) return false;
  if (l2.nsum || l2.nseg != 2 || l2.gate || !l2.colsum || l2.A[0] != l1.sum_out || l2.A[1] != l1.C) return false;
  if (l3.nsum || l3.nseg != 1 || !l3.gate || !l3.colsum || l3.A[0] != l2.C) return false;
  if (l1.M != l2.M || l1.M != l3.M) return false;
  memset(&c, 0, sizeof(c));
  c.M = l1.M; c.bm = bm;
  c.sum_parts = l1.sum_parts; c.nsum = l1.nsum; c.sum_stride = l1.sum_stride; c.x0 = l1.sum_out; c.cs0 = l1.sum_colsum;
  c.W1 = l1.W[0]; c.ldw1 = l1.ldw[0]; c.ref1 = l1.ref; c.x1 = l1.C; c.cs1 = l1.colsum;
  c.W2a = l2.W[0]; c.ldw2a = l2.ldw[0]; c.W2b = l2.W[1]; c.ldw2b = l2.ldw[1]; c.x2 = l2.C; c.cs2 = l2.colsum;
  c.W3 = l3.W[0]; c.ldw3 = l3.ldw[0]; c.ref3 = l3.ref; c.x3 = l3.C; c.cs3 = l3.colsum;
  return true;
}

hipError_t rowchain_launch(const RowChainArgs &a, hipStream_t s) {
  static bool attr[64];
  static std::mutex mu;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
  const int bm = a.bm ? a.bm : RD_BM;
  if ((bm != 16 && bm != 32 && bm != 64) || a.M % bm) return hipErrorInvalidValue;   // (a grid that does not cover M exactly must never start)
  auto lds_of = [](int rows) { return (size_t)(2 * rows * P + 4 * RD_K) * 4; };
  {
    std::lock_guard<std::mutex> lk(mu);
    if (!attr[dev]) {
      e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rowdgrad_chain<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_of(64));
      if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rowdgrad_chain<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_of(32));
      if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rowdgrad_chain<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_of(16));
      if (e != hipSuccess) return e;
      attr[dev] = true;
    }
  }
  const dim3 grid(a.M / bm), block(256);
  if (bm == 64) hipLaunchKernelGGL(k_rowdgrad_chain<4>, grid, block, lds_of(64), s, a);
  else if (bm == 32) hipLaunchKernelGGL(k_rowdgrad_chain<2>, grid, block, lds_of(32), s, a);
  else hipLaunchKernelGGL(k_rowdgrad_chain<1>, grid, block, lds_of(16), s, a);
  return hipGetLastError();
}

hipError_t rowdgrad_launch(const RowDgradArgs &a, hipStream_t s) {
  static bool attr[64];
  static std::mutex mu;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
  {
    std::lock_guard<std::mutex> lk(mu);
    if (!attr[dev]) {   // the opt-in to > 64 KiB of dynamic LDS belongs to the (device, function) pair
      const void *fns[6] = {reinterpret_cast<const void *>(&k_rowdgrad<1, false>), reinterpret_cast<const void *>(&k_rowdgrad<1, true>),
                            reinterpret_cast<const void *>(&k_rowdgrad<2, false>), reinterpret_cast<const void *>(&k_rowdgrad<2, true>),
                            reinterpret_cast<const void *>(&k_rowdgrad<1, false, true>), reinterpret_cast<const void *>(&k_rowdgrad<1, true, true>)};
      for (int i = 0; i < 6 && e == hipSuccess; ++i)
        e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, ((i & 3) < 2 ? 1 : 2) * IMG * 4 + (i >= 4 ? 4 * RD_K * 4 : 0));
      if (e != hipSuccess) return e;
      attr[dev] = true;
    }
  }
  if (a.M % RD_BM) return hipErrorInvalidValue;   // (smaller blocks exist only inside the chain launch)
  const dim3 grid(a.M / RD_BM), block(256);
  const size_t lds_bytes = (size_t)a.nseg * IMG * 4 + (a.nsum > 0 ? 4 * RD_K * 4 : 0);
  if (a.nsum > 0 && a.nseg == 1) {
    if (a.gate) hipLaunchKernelGGL((k_rowdgrad<1, true, true>), grid, block, lds_bytes, s, a);
    else hipLaunchKernelGGL((k_rowdgrad<1, false, true>), grid, block, lds_bytes, s, a);
    return hipGetLastError();
  }
  if (a.nseg == 1 && !a.gate) hipLaunchKernelGGL((k_rowdgrad<1, false>), grid, block, lds_bytes, s, a);
  else if (a.nseg == 1) hipLaunchKernelGGL((k_rowdgrad<1, true>), grid, block, lds_bytes, s, a);
  else if (!a.gate) hipLaunchKernelGGL((k_rowdgrad<2, false>), grid, block, lds_bytes, s, a);
  else hipLaunchKernelGGL((k_rowdgrad<2, true>), grid, block, lds_bytes, s, a);
  return hipGetLastError();
}

}  // namespace fdql
