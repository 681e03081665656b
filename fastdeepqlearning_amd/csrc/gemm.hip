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
// Tile: 128x128 per 256-thread workgroup, 4 waves as 2x2, each wave 2x2 MFMA tiles of
// 32x32 (64 accumulator VGPRs).  K advances in chunks of 16 through a double-buffered LDS
// image laid out [k][row] with pitch 132 floats, so every fragment read is a
// conflict-free ds_read_b32 of 32 consecutive rows for one k.  Global loads of chunk c+1
// are issued before the MFMAs of chunk c and written to the other LDS buffer after them
// (one barrier per chunk).
#include "common.h"

#include <cstdarg>
#include <cstdio>

namespace fdql {

constexpr int BM = 128, BN = 128, BK = 16, PITCH = 132, GEMM_THREADS = 256;
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
__device__ __forceinline__ bool aligned8(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 7) == 0; }

// Load this thread's share (2 x 4 floats) of a [128 rows x 16 k] operand chunk.
// kc: element (r,k) at P[r*ld + k]; thread -> rows (tid>>2)+{0,64}, k = (tid&3)*4 + j.
__device__ __forceinline__ void load_chunk_kc(const float *__restrict__ P, int ld, int R, int kend, int r0, int k0,
                                              int tid, float (&v)[8]) {
  const int kq = k0 + (tid & 3) * 4;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int r = r0 + (tid >> 2) + 64 * h;
    const float *p = P + (long long)r * ld + kq;
    if (r < R && kq + 3 < kend) {
      if (aligned16(p)) {
        const float4 x = *reinterpret_cast<const float4 *>(p);
        v[4 * h + 0] = x.x; v[4 * h + 1] = x.y; v[4 * h + 2] = x.z; v[4 * h + 3] = x.w;
      } else if (aligned8(p)) {
        const float2 x = *reinterpret_cast<const float2 *>(p);
        const float2 y = *reinterpret_cast<const float2 *>(p + 2);
        v[4 * h + 0] = x.x; v[4 * h + 1] = x.y; v[4 * h + 2] = y.x; v[4 * h + 3] = y.y;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[4 * h + j] = p[j];
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[4 * h + j] = (r < R && kq + j < kend) ? p[j] : 0.f;
    }
  }
}

// ks: element (r,k) at P[k*ld + r]; thread -> k = (tid>>5)+{0,8}, rows (tid&31)*4 + j.
__device__ __forceinline__ void load_chunk_ks(const float *__restrict__ P, int ld, int R, int kend, int r0, int k0,
                                              int tid, float (&v)[8]) {
  const int rq = r0 + (tid & 31) * 4;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int k = k0 + (tid >> 5) + 8 * h;
    const float *p = P + (long long)k * ld + rq;
    if (k < kend && rq + 3 < R) {
      if (aligned16(p)) {
        const float4 x = *reinterpret_cast<const float4 *>(p);
        v[4 * h + 0] = x.x; v[4 * h + 1] = x.y; v[4 * h + 2] = x.z; v[4 * h + 3] = x.w;
      } else if (aligned8(p)) {
        const float2 x = *reinterpret_cast<const float2 *>(p);
        const float2 y = *reinterpret_cast<const float2 *>(p + 2);
        v[4 * h + 0] = x.x; v[4 * h + 1] = x.y; v[4 * h + 2] = y.x; v[4 * h + 3] = y.y;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[4 * h + j] = p[j];
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[4 * h + j] = (k < kend && rq + j < R) ? p[j] : 0.f;
    }
  }
}

__device__ __forceinline__ void store_chunk_kc(float *__restrict__ lds, int tid, const float (&v)[8]) {
  const int kq = (tid & 3) * 4;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int r = (tid >> 2) + 64 * h;
#pragma unroll
    for (int j = 0; j < 4; ++j) lds[(kq + j) * PITCH + r] = v[4 * h + j];
  }
}

__device__ __forceinline__ void store_chunk_ks(float *__restrict__ lds, int tid, const float (&v)[8]) {
  const int rq = (tid & 31) * 4;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int k = (tid >> 5) + 8 * h;
    *reinterpret_cast<float4 *>(&lds[k * PITCH + rq]) = make_float4(v[4 * h], v[4 * h + 1], v[4 * h + 2], v[4 * h + 3]);
  }
}

__global__ __launch_bounds__(GEMM_THREADS, 2) void k_gemm_grouped(const GemmProblem *__restrict__ probs, int nprob) {
  __shared__ __attribute__((aligned(16))) float lds[2][2][BK * PITCH];

  const int tid = threadIdx.x;
  const int bid = blockIdx.x;
  int pi = 0;
  for (int i = 1; i < nprob; ++i)
    if (bid >= probs[i].tile_start) pi = i;
  const GemmProblem &P = probs[pi];

  const int M = P.M, N = P.N, nseg = P.nseg, ksplit = P.ksplit;
  const int local = bid - P.tile_start;
  const int tiles_mn = P.tiles_m * P.tiles_n;
  const int split = local / tiles_mn;
  const int rem = local - split * tiles_mn;
  const int tile_m = rem / P.tiles_n;
  const int tile_n = rem - tile_m * P.tiles_n;
  const int r0 = tile_m * BM, c0 = tile_n * BN;

  // K range of segment 0 when the problem is K-split (wgrad); whole segments otherwise.
  int kb0 = 0, ke0 = P.seg[0].K;
  if (ksplit > 1) {
    const int per = ((P.seg[0].K + ksplit - 1) / ksplit + BK - 1) / BK * BK;
    kb0 = split * per;
    ke0 = min(P.seg[0].K, kb0 + per);
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;

  // first non-empty chunk
  int s = 0, k = kb0, ke = ke0;
  while (s < nseg && k >= ke) {
    ++s;
    if (s < nseg) { k = 0; ke = P.seg[s].K; }
  }
  bool have = (s < nseg) && (ksplit == 1 || s == 0);

  float va[8], vb[8];
  int cur_akc = 1, cur_bkc = 1;
  if (have) {
    const GemmSeg &S = P.seg[s];
    cur_akc = S.a_kc; cur_bkc = S.b_kc;
    if (cur_akc) load_chunk_kc(S.A, S.lda, M, ke, r0, k, tid, va); else load_chunk_ks(S.A, S.lda, M, ke, r0, k, tid, va);
    if (cur_bkc) load_chunk_kc(S.B, S.ldb, N, ke, c0, k, tid, vb); else load_chunk_ks(S.B, S.ldb, N, ke, c0, k, tid, vb);
    if (cur_akc) store_chunk_kc(lds[0][0], tid, va); else store_chunk_ks(lds[0][0], tid, va);
    if (cur_bkc) store_chunk_kc(lds[0][1], tid, vb); else store_chunk_ks(lds[0][1], tid, vb);
  }
  __syncthreads();

  int cur = 0;
  while (have) {
    // locate the next chunk
    int ns = s, nk = k + BK, nke = ke;
    if (nk >= ke) {
      ns = s + 1;
      nk = 0;
      while (ns < nseg && P.seg[ns].K <= 0) ++ns;
      if (ns < nseg) nke = P.seg[ns].K;
    }
    const bool has_next = (ns < nseg) && (ksplit == 1 || ns == 0);
    int n_akc = 1, n_bkc = 1;
    if (has_next) {
      const GemmSeg &S = P.seg[ns];
      n_akc = S.a_kc; n_bkc = S.b_kc;
      if (n_akc) load_chunk_kc(S.A, S.lda, M, nke, r0, nk, tid, va); else load_chunk_ks(S.A, S.lda, M, nke, r0, nk, tid, va);
      if (n_bkc) load_chunk_kc(S.B, S.ldb, N, nke, c0, nk, tid, vb); else load_chunk_ks(S.B, S.ldb, N, nke, c0, nk, tid, vb);
    }

    const float *la = lds[cur][0] + wm * 64 + li;
    const float *lb = lds[cur][1] + wn * 64 + li;
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      const int row = (2 * kk + lh) * PITCH;
      const float a0 = la[row], a1 = la[row + 32];
      const float b0 = lb[row], b1 = lb[row + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }

    if (!has_next) break;
    if (n_akc) store_chunk_kc(lds[cur ^ 1][0], tid, va); else store_chunk_ks(lds[cur ^ 1][0], tid, va);
    if (n_bkc) store_chunk_kc(lds[cur ^ 1][1], tid, vb); else store_chunk_ks(lds[cur ^ 1][1], tid, vb);
    __syncthreads();
    cur ^= 1;
    s = ns; k = nk; ke = nke;
  }

  // epilogue: D[i][j] of a 32x32 tile sits at col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  float *C = P.C + (long long)split * P.split_stride;
  const int ldc = P.ldc, epi = P.epi;
  const float *bias = P.bias;
  const float *ref = P.ref;
  const int ldref = P.ldref;
#pragma unroll
  for (int tm = 0; tm < 2; ++tm) {
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      const int col = c0 + wn * 64 + tn * 32 + li;
      if (col >= N) continue;
      const float bv = bias ? bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = r0 + wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row >= M) continue;
        float x = acc[tm][tn][r] + bv;
        if (epi == EPI_LRELU) {
          x = x > 0.f ? x : 0.01f * x;
        } else if (epi == EPI_LRELU_GRAD) {
          const float a = ref[(long long)row * ldref + col];
          x = a > 0.f ? x : 0.01f * x;
        }
        C[(long long)row * ldc + col] = x;
      }
    }
  }
}

int gemm_finalize(GemmProblem *probs, int nprob) {
  int total = 0;
  for (int i = 0; i < nprob; ++i) {
    GemmProblem &p = probs[i];
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.N + BN - 1) / BN;
    if (p.ksplit < 1) p.ksplit = 1;
    p.tile_start = total;
    total += p.tiles_m * p.tiles_n * p.ksplit;
  }
  return total;
}

double gemm_flops(const GemmProblem &p) {
  double k = 0;
  for (int s = 0; s < p.nseg; ++s) k += p.seg[s].K;
  return 2.0 * p.M * (double)p.N * k;
}

hipError_t gemm_launch(const GemmProblem *probs_dev, int nprob, int total_blocks, hipStream_t stream) {
  if (total_blocks <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_gemm_grouped, dim3(total_blocks), dim3(GEMM_THREADS), 0, stream, probs_dev, nprob);
  return hipGetLastError();
}

}  // namespace fdql
