// Small-batch fp32 MFMA GEMM (shape GEMM_SMALL of the grouped-GEMM problem tables, common.h).
//
// temporal_len 2 / small batches (franQ/Agent/deepQlearning.py:105-127 at T = 2: 512 rows) are latency-bound: a layer is a
// 512 x 256 x 256 problem, 32 tiles of the 64x64 tile kernel on 256 CUs, and a launch lasts as long as ONE workgroup's
// serial K loop (one wave per SIMD walking 8-16 chunks, two barriers and an LDS round trip each: 9 us + 0.5 us per 16 k,
// profiles/r03_temporal_len_2_stages.txt).  Here the K range is what is parallel:
//   * a 1024-thread workgroup owns a 64 x 32 output tile; its 16 waves are 2 row tiles x 8 K-SLICES: wave (rt, sl) multiplies
//     rows [32 rt, 32 rt + 32) by the tile's 32 columns over its contiguous eighth of the k range (all K-segments of the
//     problem laid end to end: torch.cat never materialises);
//   * no LDS staging and no barrier in front of the MFMAs: every operand fragment goes global -> registers directly in MFMA
//     layout (lane (i, h) holds 4 consecutive k of its row / column: one 16-byte load per 8 k when the operand is K-contiguous
//     and aligned, four coalesced dword loads when it is K-strided), ALL of a wave's loads are in flight before its first MFMA;
//   * the 8 partial tiles of a row tile meet in LDS (64 KiB) and are summed in slice order by the threads that then apply
//     the epilogue (bias, LeakyReLU / LeakyReLU' gate / add-reference, per-64-row column sums) and store the tile.
// The sum over k runs in a fixed order (within a slice in k order, slices 0..7): results are reproducible, not bitwise equal
// to the tile kernels' (different association), like every other kernel pair of this library.
#include "common.h"

namespace fdql {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) float *gcf;
typedef const __attribute__((address_space(1))) v4f *gcf4;
typedef __attribute__((address_space(1))) float *gf;

constexpr int SG_BM = 64, SG_BN = 32, SG_SLICES = 8, SG_THREADS = 1024;
constexpr int SG_PASS = 4;   // 8-k groups a wave loads before it multiplies (32 k per slice and pass: K <= 256 is one pass; a bigger
                             // pass doubles the code of a kernel whose cold instruction fetch is part of its latency)

// Operand fragments are loaded in passes of SG_PASS 8-k groups.  The load phase of a pass is STRAIGHT-LINE code - no
// branch, no use of a loaded value - so that all of its requests are in flight together: with a branch per group (16-byte
// load or four dwords?) or a select behind each load the compiler waits in every group and the round trips (1-2 us each)
// run one after the other (measured: 2.5 us per group, 64 us for a 25-group slice).  Hence:
//   * one uniform decision per pass and operand - VEC (every group of the pass is K-contiguous, 16-byte aligned and whole:
//     one dwordx4 per group) or not (four dword loads per group at base + c * kstride: K-contiguous or K-strided alike);
//   * groups beyond the slice's end re-read its last group (clamped) and are zeroed by the mask afterwards;
//   * the segment of a group comes from a ballot over per-lane segment ends, its descriptor from v_readlane - no loops.
struct SgSeg {   // per-lane copy of one K-segment's descriptor (lane s <-> segment s)
  unsigned long long A, B;
  int lda, ldb, K, kc, gend;   // kc: bit 0 a_kc, bit 1 b_kc; gend: 8-k groups up to and including this segment
};
__device__ __forceinline__ unsigned long long sg_bcast64(unsigned long long v, int l) {
  return ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(v >> 32), l) << 32) | (unsigned)__builtin_amdgcn_readlane((int)v, l);
}
// uniform facts about group g: its segment's descriptor and its first k
struct SgGroup { unsigned long long A, B; int lda, ldb, K, kc, k0; };
__device__ __forceinline__ SgGroup sg_group(const SgSeg &d, int nseg, int lane, int g) {
  const int seg = __builtin_popcountll(__ballot(lane < nseg - 1 && d.gend <= g));   // segments that end at or before g
  const int gbase = seg > 0 ? __builtin_amdgcn_readlane(d.gend, seg > 0 ? seg - 1 : 0) : 0;
  SgGroup G;
  G.A = sg_bcast64(d.A, seg); G.B = sg_bcast64(d.B, seg);
  G.lda = __builtin_amdgcn_readlane(d.lda, seg); G.ldb = __builtin_amdgcn_readlane(d.ldb, seg);
  G.K = __builtin_amdgcn_readlane(d.K, seg); G.kc = __builtin_amdgcn_readlane(d.kc, seg);
  G.k0 = (g - gbase) << 3;
  return G;
}
__device__ __forceinline__ bool sg_vec_ok(unsigned long long p, int ld, bool kc, int k0, int K) {
  return kc && k0 + 8 <= K && (ld & 3) == 0 && (p & 15) == 0;
}
// lane (idx, h) of operand P: values k = kb + c (kb = k0 + 4 h) of row / column rc (clamped by the caller)
template <bool VEC>
__device__ __forceinline__ void sg_issue(unsigned long long p, int ld, bool kc, int rc, int kb, int K, v4f &o) {
  gcf P = (gcf)p;
  if constexpr (VEC) {
    o = *(gcf4)(P + (long long)rc * ld + kb);
  } else {
    const long long rs = kc ? ld : 1, ks = kc ? 1 : ld;   // uniform strides: element (r, k) at r * rs + k * ks
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int k = kb + c, kcl = k < K ? k : K - 1;
      o[c] = P[(long long)rc * rs + (long long)kcl * ks];
    }
  }
}
__device__ __forceinline__ void sg_mask(v4f &o, int rem) {   // keep the first `rem` values (rem <= 0: none)
#pragma unroll
  for (int c = 0; c < 4; ++c) o[c] = c < rem ? o[c] : 0.f;
}

template <bool AVEC, bool BVEC>
__device__ __forceinline__ void sg_pass(const SgSeg &d, int nseg, int lane, int h, int g0, int g_hi, int arc, int brc, bool a_ok,
                                        bool b_ok, f32x16 &acc) {
  v4f av[SG_PASS], bv[SG_PASS];
  int rem[SG_PASS];
#pragma unroll
  for (int u = 0; u < SG_PASS; ++u) {
    const int g = g0 + u < g_hi ? g0 + u : g_hi - 1;
    const SgGroup G = sg_group(d, nseg, lane, g);
    const int kb = G.k0 + 4 * h;
    sg_issue<AVEC>(G.A, G.lda, G.kc & 1, arc, kb, G.K, av[u]);
    sg_issue<BVEC>(G.B, G.ldb, (G.kc & 2) != 0, brc, kb, G.K, bv[u]);
    rem[u] = g0 + u < g_hi ? G.K - kb : 0;
  }
#pragma unroll
  for (int u = 0; u < SG_PASS; ++u) {
    sg_mask(av[u], a_ok ? rem[u] : 0);
    sg_mask(bv[u], b_ok ? rem[u] : 0);
#pragma unroll
    for (int c = 0; c < 4; ++c) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][c], bv[u][c], acc, 0, 0, 0);
  }
}

__global__ __launch_bounds__(SG_THREADS) void k_gemm_small(const GemmProblem *__restrict__ probs, int nprob) {
  __shared__ float part[2 * SG_SLICES][32][SG_BN];   // [wave = rt * 8 + slice][row][col]: 64 KiB
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int idx = lane & 31, h = lane >> 5;
  const int rt = wave >> 3, sl = wave & 7;
  const int pi = find_problem<GemmProblem, &GemmProblem::tile_start>(probs, nprob, (int)blockIdx.x, lane);
  const GemmProblem &P = probs[pi];
  const int local = (int)blockIdx.x - P.tile_start;
  const int tm = local / P.tiles_n, tn = local - tm * P.tiles_n;
  const int M = P.M, N = P.N, nseg = P.nseg;
  const int r0 = tm * SG_BM + rt * 32, c0 = tn * SG_BN;
  // The segment table in ONE load round: lane s holds segment s's descriptor (nseg <= GEMM_MAX_SEG = 24 < 64 lanes)
  typedef const __attribute__((address_space(1))) GemmSeg *gseg;
  gseg segs = (gseg)P.seg;
  const int sl_i = lane < nseg ? lane : 0;
  SgSeg d;
  d.A = (unsigned long long)segs[sl_i].A; d.B = (unsigned long long)segs[sl_i].B;
  d.lda = segs[sl_i].lda; d.ldb = segs[sl_i].ldb; d.K = lane < nseg ? segs[sl_i].K : 0;
  d.kc = (segs[sl_i].a_kc ? 1 : 0) | (segs[sl_i].b_kc ? 2 : 0);
  const int dgroups = (d.K + 7) >> 3;
  // inclusive prefix of the group counts (a segment's last group may be partial); gtot: all groups of the problem
  int gtot = 0;
  d.gend = 0;
  for (int s = 0; s < nseg; ++s) {
    gtot += __builtin_amdgcn_readlane(dgroups, s);
    d.gend = lane == s ? gtot : d.gend;
  }
  // this wave's groups: [g_lo, g_hi), a contiguous eighth of the k range
  const int per = (gtot + SG_SLICES - 1) / SG_SLICES;
  const int g_lo = sl * per, g_hi = g_lo + per < gtot ? g_lo + per : gtot;

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const int ar = r0 + idx, bc = c0 + idx;
  const bool a_ok = ar < M, b_ok = bc < N;
  const int arc = a_ok ? ar : 0, brc = b_ok ? bc : 0;   // clamped: every load is unconditional

#pragma unroll 1
  for (int g0 = g_lo; g0 < g_hi; g0 += SG_PASS) {
    // one uniform decision per pass and operand (a pass that straddles segments of different kinds takes the dword form)
    bool avec = true, bvec = true;
    for (int u = 0; u < SG_PASS; ++u) {
      const int g = g0 + u < g_hi ? g0 + u : g_hi - 1;
      const SgGroup G = sg_group(d, nseg, lane, g);
      avec = avec && sg_vec_ok(G.A, G.lda, G.kc & 1, G.k0, G.K);
      bvec = bvec && sg_vec_ok(G.B, G.ldb, (G.kc & 2) != 0, G.k0, G.K);
    }
    if (avec && bvec) sg_pass<true, true>(d, nseg, lane, h, g0, g_hi, arc, brc, a_ok, b_ok, acc);
    else if (avec) sg_pass<true, false>(d, nseg, lane, h, g0, g_hi, arc, brc, a_ok, b_ok, acc);
    else if (bvec) sg_pass<false, true>(d, nseg, lane, h, g0, g_hi, arc, brc, a_ok, b_ok, acc);
    else sg_pass<false, false>(d, nseg, lane, h, g0, g_hi, arc, brc, a_ok, b_ok, acc);
  }
  // ---- reduction over the slices + epilogue: thread -> column tid % 32, rows tid / 32 and tid / 32 + 32 of the tile.
  // The epilogue's operands (bias, reference values) are requested BEFORE the partial tiles meet: their round trip runs
  // under the MFMAs' tail and the barrier instead of after them.
  const int col = tid & 31, rl = tid >> 5;
  const int gcol = c0 + col;
  const int epi = P.epi;
  gcf bias = (gcf)P.bias, ref = (gcf)P.ref;
  gf C = (gf)P.C;
  const float bvl = (bias && gcol < N) ? bias[gcol] : 0.f;
  const bool want_ref = epi == EPI_LRELU_GRAD || epi == EPI_ADD_REF;
  float rv[2] = {0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int row = tm * SG_BM + t * 32 + rl;
    if (want_ref && row < M && gcol < N) rv[t] = ref[(long long)row * P.ldref + gcol];
  }
  // D layout: lane (j = idx: column, h), register r: row i = (r & 3) + 8 (r >> 2) + 4 h
#pragma unroll
  for (int r = 0; r < 16; ++r) part[wave][(r & 3) + 8 * (r >> 2) + 4 * h][idx] = acc[r];
  __syncthreads();

  float colpart = 0.f;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int row = tm * SG_BM + t * 32 + rl;
    float x = 0.f;
#pragma unroll
    for (int s = 0; s < SG_SLICES; ++s) x += part[t * SG_SLICES + s][rl][col];
    if (row < M && gcol < N) {
      x += bvl;
      if (epi == EPI_LRELU) x = x > 0.f ? x : 0.01f * x;
      else if (epi == EPI_LRELU_GRAD) x = rv[t] > 0.f ? x : 0.01f * x;
      else if (epi == EPI_ADD_REF) x += rv[t];
      C[(long long)row * P.ldc + gcol] = x;
      colpart += x;
    }
  }
  if (P.colsum) {   // column sums of the stored values per 64-row block (bias gradients): rows in a fixed order
    __syncthreads();
    float *red = &part[0][0][0];   // [32 row pairs][32 columns]
    red[rl * SG_BN + col] = colpart;
    __syncthreads();
    if (tid < SG_BN && c0 + tid < N && tm * SG_BM < M) {
      float t = 0.f;
#pragma unroll 8
      for (int j = 0; j < 32; ++j) t += red[j * SG_BN + tid];
      ((gf)P.colsum)[(long long)tm * N + c0 + tid] = t;
    }
  }
}

}  // namespace

bool gemm_small_takes(const GemmProblem &p) {
  if (p.ksplit > 1 || p.C2 || p.hf_w || p.fz_h) return false;
  if (p.emit_seg >= 0 && p.emit_seg != p.nseg - 1) return false;
  if (p.nseg < 1 || p.M < 1 || p.N < 1) return false;
  for (int s = 0; s < p.nseg; ++s)
    if (p.seg[s].K < 1) return false;
  if ((p.epi == EPI_LRELU_GRAD || p.epi == EPI_ADD_REF) && !p.ref) return false;
  return true;
}

hipError_t gemm_small_launch(const GemmProblem *probs_dev, int nprob, int total_blocks, hipStream_t stream) {
  if (total_blocks <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_gemm_small, dim3(total_blocks), dim3(SG_THREADS), 0, stream, probs_dev, nprob);
  return hipGetLastError();
}

}  // namespace fdql
