// Small-block forward chain (fwdchain.h): encoder -> joiner -> online actor + target actor on 16- / 32-row blocks.
//
// Why a second forward chain kernel (measured, profiles/r05_stage_times_config2_per_rank.txt): k_chain keeps a 64- or 32-row block
// per CU and requests a layer's weight fragments one 32-k group ahead of their MFMAs - at 12 800 rows the 196 workgroups are MFMA
// work, but one rank's share of a data-parallel batch (6 400 / 3 200 / 1 600 rows) is 100-200 blocks of eight DEPENDENT layers whose
// time is the latency of their weight requests: 0.102 ms at 3 200 rows, 0.108 at 6 400 (26 / 48 TFLOP/s).  Here
//   * a workgroup owns 16 RT rows (RT = 1, 2), the accumulators of a layer are 16 RT registers, and the weight ring holds EIGHT
//     16-k groups (128 registers): a lane's 16-byte request W[n0 + 16 ct + j][16 g + 4 kq ..] is, component s, the B operand of a
//     v_mfma_f32_16x16x4_f32 whose k's are 16 g + 4 kq + s (K-contiguous torch Linear weights need no transpose: the A operand - the
//     row's 16 bytes x[row][16 g + 4 kq ..] from the LDS image - pairs the same k's);
//   * the ring runs on across segment and layer boundaries (weights do not depend on activations): the first groups of the next
//     layer are in flight under the epilogue and the barrier of this one;
//   * two workgroups share a CU (images: 2 x [16 RT][260] + the observation image), so one's epilogue / barrier sits under the
//     other's MFMAs.
// Layer outputs go to the next layer's LDS image (ds_write_b32: the MFMA leaves a lane with rows 4 kq + r of column j) and, after
// the barrier, from the image to memory as whole rows (global_store_dwordx4).
#include "fwdchain.h"

#include <mutex>

namespace fdql {
namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) float *gcf;
typedef const __attribute__((address_space(1))) v4f *gcf4;
typedef __attribute__((address_space(1))) v4f *gf4;
typedef __attribute__((address_space(1))) float *gf;

constexpr int FP = F3_W + 4;   // row pitch of the wide images ((FP / 4) odd: the 16 rows of a fragment read hit distinct banks)
constexpr int NDEPTH = 4;      // ... of the narrow head's (one request per group)

__device__ __forceinline__ int f3_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// VOBS: the observation segment through the 16-byte request ring as well (wide observations whose weight rows are multiples of
// four floats - config 4: 376 columns = 24 groups; the dword path below is one request round per 16-k group, 0.05 ms of them)
// RT = 1 runs EIGHT waves per workgroup (32 output columns each, two waves per SIMD): a 16-byte weight request of 16 rows costs
// its wave ~100 cycles of issue (profiles/r06_fwd3_cycle_stamps.txt), one per four MFMAs at 16 rows - with a second wave on the SIMD
// that time is the other wave's MFMAs.  RT = 2 gets the same from its two workgroups per CU.
template <int RT, bool VOBS>
__global__ __launch_bounds__(RT == 1 ? 512 : 256, RT == 1 ? 1 : 2) void k_fwd3(const Fwd3Args a) {
  constexpr int BM = 16 * RT;
  constexpr int NW = RT == 1 ? 8 : 4;      // waves per workgroup
  constexpr int CT = 16 / NW;              // 16-column tiles per wave
  constexpr int NT = 64 * NW;              // threads
  // 16-k weight groups in flight per wave (16 % DEPTH == 0: a segment's groups keep their ring slots): a group is 16 RT MFMAs
  // = 0.21 RT us of matrix-pipe time against ~1 us of request latency
  constexpr int DEPTH = RT == 1 ? 8 : 4;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = f3_uni(tid >> 6);
  const int j = lane & 15, kq = lane >> 4;
  const int r0 = blockIdx.x * BM, n0 = wave * (16 * CT);
  const int K0 = a.K0, G0 = (K0 + 15) >> 4, PX = 16 * G0 + 4;   // (VOBS: G0 % DEPTH == 0, checked by the launcher)
  // RT = 1 runs one workgroup per CU (the plan picks 16-row blocks only up to one per CU): every activation keeps an LDS image
  // of its own and goes to memory at the END of the kernel - a store in the middle makes the next use of a requested weight
  // wait for vmcnt(0), i.e. for the store's round trip (loads and stores retire out of order with each other), five times per
  // block with nothing to cover it.  RT = 2 (two workgroups per CU, 79 KB each) ping-pongs two images and stores as it goes.
  constexpr bool DEFER = RT == 1;
  constexpr int NIMG = DEFER ? 6 : 2;
  float *const I_eh = lds, *const I_e = lds + BM * FP, *const I_jh = DEFER ? lds + 2 * BM * FP : I_eh, *const I_s = DEFER ? lds + 3 * BM * FP : I_e,
               *const I_ah = DEFER ? lds + 4 * BM * FP : I_eh, *const I_th = DEFER ? lds + 5 * BM * FP : I_eh;
  float *const HS = lds + NIMG * BM * FP, *const X0 = HS + NW * BM * 16;   // HS: the narrow head's partial tiles, one per wave
  const bool has_act = r0 < a.M, has_tgt = r0 >= a.B;   // (uniform: a block never straddles a time step, B % BM == 0)

  v4f ring[DEPTH][CT];
  v4f acc[RT][CT];
  // a wide segment's requests: uniform base (W + koff, column tile ct, group g: scalar arithmetic) + the lane's offset
  // (row n0 + j of pitch ldw, column 4 kq): one 32-bit register per pitch instead of a 64-bit pointer per segment
  struct WSeg { const float *W; int ldw, off; };
  auto wseg = [&](const float *W, int ldw, int koff) __attribute__((always_inline)) { return WSeg{W + koff, ldw, (n0 + j) * ldw + 4 * kq}; };
  auto req = [&](const WSeg &w, int g, v4f (&d)[CT]) __attribute__((always_inline)) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) d[ct] = *(gcf4)((gcf)(w.W + (16 * ct * w.ldw + 16 * g)) + w.off);
  };
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = v4f{0.f, 0.f, 0.f, 0.f};
  };
  // one 256-k segment: acc += image x W^T.  On entry the ring holds the segment's groups 0 .. DEPTH - 1; a group's slot is
  // refilled with group g + DEPTH of this segment or group g + DEPTH - 16 of the next one (wn) right behind its MFMAs.
  auto wide = [&](const float *IM, const WSeg &w, const WSeg &wn, bool has_next) __attribute__((always_inline)) {
    const float *xp = IM + j * FP + 4 * kq;
    v4f xc[RT], xn[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) xc[rt] = *reinterpret_cast<const v4f *>(xp + 16 * rt * FP);
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      if (g + 1 < 16) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) xn[rt] = *reinterpret_cast<const v4f *>(xp + 16 * rt * FP + 16 * (g + 1));
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
            acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(xc[rt][s], ring[g % DEPTH][ct][s], acc[rt][ct], 0, 0, 0);
      if (g + DEPTH < 16) req(w, g + DEPTH, ring[g % DEPTH]);
      else if (has_next) req(wn, g + DEPTH - 16, ring[g % DEPTH]);
      // (the empty asm pins the requests here: left to the scheduler they sink next to their uses, eight groups later)
      asm volatile("" ::: "memory");
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) xc[rt] = xn[rt];
    }
  };
  // the ragged observation segment (K0 columns, zero-padded image X0, any row pitch / alignment of W): dword requests, one
  // 16-k group ahead; wg holds group 0 on entry
  auto rag_load = [&](gcf W, int ldw, int g, float (&w)[4 * CT]) __attribute__((always_inline)) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int k = 16 * g + 4 * kq + s;
        w[4 * ct + s] = k < K0 ? W[(long long)(n0 + 16 * ct + j) * ldw + k] : 0.f;
      }
  };
  auto rag_mma = [&](int g, const float (&w)[4 * CT]) __attribute__((always_inline)) {
    v4f x[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) x[rt] = *reinterpret_cast<const v4f *>(X0 + (16 * rt + j) * PX + 16 * g + 4 * kq);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[rt][s], w[4 * ct + s], acc[rt][ct], 0, 0, 0);
  };
  // VOBS: the same segment as G0 (a multiple of DEPTH) ring groups; columns >= K0 of X0 are zero, so what the last group reads
  // past a weight row's end does not count
  auto wide_obs = [&](const WSeg &w, const WSeg &wn) __attribute__((always_inline)) {
    const float *xp = X0 + j * PX + 4 * kq;
#pragma unroll 1
    for (int gb = 0; gb < G0; gb += DEPTH) {
#pragma unroll
      for (int q = 0; q < DEPTH; ++q) {
        const int g = gb + q;
        v4f x[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) x[rt] = *reinterpret_cast<const v4f *>(xp + 16 * rt * PX + 16 * g);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[rt][s], ring[q][ct][s], acc[rt][ct], 0, 0, 0);
        if (g + DEPTH < G0) req(w, g + DEPTH, ring[q]);
        else req(wn, g + DEPTH - G0, ring[q]);
        asm volatile("" ::: "memory");
      }
    }
  };
  auto ragged = [&](gcf W, int ldw, float (&w0)[4 * CT]) __attribute__((always_inline)) {
    float w1[4 * CT];
#pragma unroll 1
    for (int g = 0; g < G0; g += 2) {
      if (g + 1 < G0) rag_load(W, ldw, g + 1, w1);
      asm volatile("" ::: "memory");
      rag_mma(g, w0);
      if (g + 1 < G0) {
        if (g + 2 < G0) rag_load(W, ldw, g + 2, w0);
        asm volatile("" ::: "memory");
        rag_mma(g + 1, w1);
      }
    }
  };
  // bias (+ LeakyReLU) and the tile into an LDS image: lane (j, kq) holds rows 16 rt + 4 kq + r of column n0 + 16 ct + j
  float bq[CT];   // the layer's bias, requested before its K loop
  auto bias_req = [&](const float *bias) __attribute__((always_inline)) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) bq[ct] = ((gcf)bias)[n0 + 16 * ct + j];
  };
  auto to_image = [&](bool lrelu, float *IM) __attribute__((always_inline)) {
    float b[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) b[ct] = bq[ct];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc[rt][ct][r] + b[ct];
          if (lrelu) v = v > 0.f ? v : 0.01f * v;
          IM[(16 * rt + 4 * kq + r) * FP + n0 + 16 * ct + j] = v;
        }
  };
  // a finished image -> memory, whole rows (wave w: rows w, w + NW, ...); rows [lo, hi) of the batch only, stored at row - shift
  auto store_image = [&](const float *IM, float *out, int lo, int hi, int shift) __attribute__((always_inline)) {
    if (!out) return;
#pragma unroll
    for (int u = 0; u < BM / NW; ++u) {
      const int row = wave + NW * u, gr = r0 + row;
      if (gr >= lo && gr < hi) *(gf4)(out + (long long)(gr - shift) * F3_W + lane * 4) = *reinterpret_cast<const v4f *>(IM + row * FP + lane * 4);
    }
  };
  // narrow head over cat(S = IS, H = IH).  P <= 16 (one column tile): the K range is dealt to the four waves - wave w takes the
  // 128 k's [128 w, 128 w + 128) (waves 0, 1 read S, waves 2, 3 read H), its eight 16-byte requests are issued by narrow_req()
  // before the hidden layer's epilogue, the four partial tiles meet in LDS and are added in wave order.  Wider heads (config 4: 34
  // outputs): wave w owns column tile w over the whole K range (waves past the head's width idle).
  constexpr int KW = 512 / NW, NG = KW / 16;   // the wave's share of the head's 512 k's, in 16-k groups
  v4f hw[NG];
  auto narrow_req = [&](const Fwd3Mlp &m) __attribute__((always_inline)) {
    if (a.P > 16) return;
    gcf w = (gcf)m.Wh + (long long)(j < a.P ? j : a.P - 1) * (2 * F3_W) + KW * wave + 4 * kq;
#pragma unroll
    for (int q = 0; q < NG; ++q) hw[q] = *(gcf4)(w + 16 * q);
  };
  auto narrow = [&](const Fwd3Mlp &m, const float *IS, const float *IH, float *out, int shift) __attribute__((always_inline)) {
    const int P = a.P;
    if (P <= 16) {
      const float *xp = (wave < NW / 2 ? IS : IH) + j * FP + KW * (wave & (NW / 2 - 1)) + 4 * kq;
      v4f hs[4][RT];
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) hs[s][rt] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        v4f x[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) x[rt] = *reinterpret_cast<const v4f *>(xp + 16 * rt * FP + 16 * g);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) hs[s][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[rt][s], hw[g][s], hs[s][rt], 0, 0, 0);
      }
      float *sc = HS + wave * (BM * 16);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const v4f t = (hs[0][rt] + hs[1][rt]) + (hs[2][rt] + hs[3][rt]);
#pragma unroll
        for (int r = 0; r < 4; ++r) sc[(16 * rt + 4 * kq + r) * 16 + j] = t[r];
      }
      __syncthreads();
      for (int idx = tid; idx < BM * 16; idx += NT) {
        const int row = idx >> 4, col = idx & 15;
        if (col < P) {
          float v = HS[idx];
#pragma unroll
          for (int w8 = 1; w8 < NW; ++w8) v += HS[w8 * BM * 16 + idx];   // (wave order)
          ((gf)out)[(long long)(r0 + row - shift) * P + col] = v + ((gcf)m.bh)[col];
        }
      }
      return;
    }
    if (16 * wave >= P) return;
    const int n = 16 * wave + j;
    const bool ok = n < P;
    gcf w = (gcf)m.Wh + (long long)(ok ? n : P - 1) * (2 * F3_W) + 4 * kq;
    v4f wr[NDEPTH];
#pragma unroll
    for (int q = 0; q < NDEPTH; ++q) wr[q] = *(gcf4)(w + 16 * q);
    v4f hs[4][RT];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) hs[s][rt] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < 32; ++g) {
      const float *xp = (g < 16 ? IS : IH) + j * FP + 16 * (g & 15) + 4 * kq;
      v4f x[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) x[rt] = *reinterpret_cast<const v4f *>(xp + 16 * rt * FP);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) hs[s][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[rt][s], wr[g % NDEPTH][s], hs[s][rt], 0, 0, 0);
      if (g + NDEPTH < 32) wr[g % NDEPTH] = *(gcf4)(w + 16 * (g + NDEPTH));
      asm volatile("" ::: "memory");
    }
    if (ok) {
      const float b = ((gcf)m.bh)[n];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const v4f t = (hs[0][rt] + hs[1][rt]) + (hs[2][rt] + hs[3][rt]);
#pragma unroll
        for (int r = 0; r < 4; ++r) ((gf)out)[(long long)(r0 + 16 * rt + 4 * kq + r - shift) * P + n] = t[r] + b;
      }
    }
  };

  // ---- requests that depend on nothing: the first observation group(s) of layer 1, the ring for the first wide segment
  float wg[4 * CT];
  const int ld_eh = K0 + F3_W;
  const WSeg w_eh = wseg(a.enc.Wh, ld_eh, K0), w_j0 = wseg(a.joi.W0, F3_W, 0), w_jh0 = wseg(a.joi.Wh, 2 * F3_W, 0),
             w_jh1 = wseg(a.joi.Wh, 2 * F3_W, F3_W), w_a0 = wseg(a.act.W0, F3_W, 0), w_t0 = wseg(a.act_t.W0, F3_W, 0), w_none = {nullptr, 0, 0};
  const WSeg w_x0 = wseg(a.enc.W0, K0, 0), w_xh = wseg(a.enc.Wh, ld_eh, 0);
  if constexpr (!VOBS) rag_load((gcf)a.enc.W0, K0, 0, wg);
  bias_req(a.enc.b0);
#pragma unroll
  for (int q = 0; q < DEPTH; ++q) req(VOBS ? w_x0 : w_eh, q, ring[q]);
  asm volatile("" ::: "memory");
  // ---- observation rows -> X0 (zero-padded to 16 G0 columns)
  {
    const int KP = 16 * G0;
    for (int idx = tid; idx < BM * KP; idx += NT) {
      const int row = idx / KP, col = idx - row * KP;
      float v = 0.f;
      int c0 = 0;
#pragma unroll
      for (int sgm = 0; sgm < F3_MAX_IN; ++sgm) {
        if (sgm < a.nin) {
          const int wdt = a.in_w[sgm];
          if (col >= c0 && col < c0 + wdt) v = ((gcf)a.in[sgm])[(long long)(r0 + row) * a.in_ld[sgm] + (col - c0)];
          c0 += wdt;
        }
      }
      X0[row * PX + col] = v;
    }
  }
  __syncthreads();

  // ---- encoder, hidden layer: h_e = LeakyReLU(W0 x + b0)
  zero_acc();
  if constexpr (VOBS) {
    wide_obs(w_x0, w_xh);
  } else {
    ragged((gcf)a.enc.W0, K0, wg);
    rag_load((gcf)a.enc.Wh, ld_eh, 0, wg);   // the head's observation part: first group under the epilogue
  }
  to_image(true, I_eh);
  bias_req(a.enc.bh);
  // ---- encoder head: e = Wh cat(x, h_e) + bh (its observation part needs no barrier)
  zero_acc();
  if constexpr (VOBS) wide_obs(w_xh, w_eh);
  else ragged((gcf)a.enc.Wh, ld_eh, wg);
  __syncthreads();
  if constexpr (!DEFER) store_image(I_eh, a.enc_h, 0, a.N, 0);
  wide(I_eh, w_eh, w_j0, true);
  to_image(false, I_e);
  bias_req(a.joi.b0);
  __syncthreads();
  if constexpr (!DEFER) store_image(I_e, a.enc_out, 0, a.N, 0);
  // ---- joiner, hidden layer: h_j = LeakyReLU(W0 e + b0)
  zero_acc();
  wide(I_e, w_j0, w_jh0, true);
  to_image(true, I_jh);
  bias_req(a.joi.bh);
  __syncthreads();
  if constexpr (!DEFER) store_image(I_jh, a.joi_h, 0, a.N, 0);
  // ---- joiner head: state = Wh cat(e, h_j) + bh (two images: over e, once every wave has read it)
  zero_acc();
  wide(I_e, w_jh0, w_jh1, true);
  wide(I_jh, w_jh1, has_act ? w_a0 : w_t0, true);
  if constexpr (!DEFER) __syncthreads();
  to_image(false, I_s);
  __syncthreads();
  if constexpr (!DEFER) store_image(I_s, a.state, 0, a.N, 0);
  // ---- online actor on rows [0, M): h_a, logits = Wh cat(state, h_a) + bh
  if (has_act) {
    bias_req(a.act.b0);
    zero_acc();
    wide(I_s, w_a0, w_t0, has_tgt);
    narrow_req(a.act);
    to_image(true, I_ah);
    __syncthreads();
    if constexpr (!DEFER) store_image(I_ah, a.act_h, 0, a.M, 0);
    narrow(a.act, I_s, I_ah, a.act_out, 0);
  }
  // ---- target actor on rows [B, N), stored at row - B
  if (has_tgt) {
    bias_req(a.act_t.b0);
    zero_acc();
    wide(I_s, w_t0, w_none, false);
    narrow_req(a.act_t);
    if (!DEFER && has_act) __syncthreads();   // two images: the online head's waves are done with h_a
    to_image(true, I_th);
    __syncthreads();
    narrow(a.act_t, I_s, I_th, a.act_t_out, a.B);
  }
  if constexpr (DEFER) {
    store_image(I_eh, a.enc_h, 0, a.N, 0);
    store_image(I_e, a.enc_out, 0, a.N, 0);
    store_image(I_jh, a.joi_h, 0, a.N, 0);
    store_image(I_s, a.state, 0, a.N, 0);
    if (has_act) store_image(I_ah, a.act_h, 0, a.M, 0);
  }
}

}  // namespace

bool fwd3_takes(const Fwd3Args &a) {
  auto al = [](const void *p, uintptr_t n) { return p && (reinterpret_cast<uintptr_t>(p) & (n - 1)) == 0; };
  if ((a.bm != 16 && a.bm != 32) || a.N <= 0 || a.N % a.bm || a.B % a.bm || a.M != a.N - a.B || a.M <= 0) return false;
  if (a.nin < 1 || a.nin > F3_MAX_IN || a.K0 < 1 || a.K0 > F3_MAX_K0 || a.P < 1 || a.P > F3_MAX_P) return false;
  int k = 0;
  for (int i = 0; i < a.nin; ++i) { if (!al(a.in[i], 4) || a.in_w[i] < 1 || a.in_ld[i] < a.in_w[i]) return false; k += a.in_w[i]; }
  if (k != a.K0) return false;
  for (const Fwd3Mlp *m : {&a.enc, &a.joi, &a.act, &a.act_t})
    if (!al(m->W0, 4) || !al(m->b0, 4) || !al(m->Wh, 4) || !al(m->bh, 4)) return false;
  // 16-byte requests: weight rows of the wide segments (pitch 256 / 512 / K0 + 256 floats: any dword-aligned base works on
  // gfx950), whole-row stores of the activations (16-byte aligned bases)
  for (const float *p : {a.enc_h, a.enc_out, a.joi_h, a.state, a.act_h})
    if (p && (reinterpret_cast<uintptr_t>(p) & 15)) return false;
  if (!a.state || !a.act_out || !a.act_t_out || !al(a.act_out, 4) || !al(a.act_t_out, 4)) return false;
  return true;
}

hipError_t fwd3_launch(const Fwd3Args &a, hipStream_t s) {
  if (!fwd3_takes(a)) return hipErrorInvalidValue;   // (a grid that does not cover the rows exactly must never start)
  static bool attr[64];
  static std::mutex mu;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
  auto lds_of = [&](int bm, int k0) { return (size_t)((bm == 16 ? 6 : 2) * bm * FP + (bm == 16 ? 8 : 4) * bm * 16 + bm * (((k0 + 15) & ~15) + 4)) * 4; };
  {
    std::lock_guard<std::mutex> lk(mu);
    if (!attr[dev]) {
      const void *fns[4] = {reinterpret_cast<const void *>(&k_fwd3<1, false>), reinterpret_cast<const void *>(&k_fwd3<1, true>),
                            reinterpret_cast<const void *>(&k_fwd3<2, false>), reinterpret_cast<const void *>(&k_fwd3<2, true>)};
      for (int i = 0; i < 4 && e == hipSuccess; ++i)
        e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_of(i < 2 ? 16 : 32, F3_MAX_K0));
      if (e != hipSuccess) return e;
      attr[dev] = true;
    }
  }
  const dim3 grid(a.N / a.bm), block(a.bm == 16 ? 512 : 256);
  const int depth = a.bm == 16 ? 8 : 4, g0 = (a.K0 + 15) / 16;
  const bool vobs = a.K0 >= 128 && a.K0 % 4 == 0 && g0 % depth == 0;   // (row pitches K0 and K0 + 256 are then multiples of four floats)
  if (a.bm == 16) {
    if (vobs) hipLaunchKernelGGL((k_fwd3<1, true>), grid, block, lds_of(16, a.K0), s, a);
    else hipLaunchKernelGGL((k_fwd3<1, false>), grid, block, lds_of(16, a.K0), s, a);
  } else {
    if (vobs) hipLaunchKernelGGL((k_fwd3<2, true>), grid, block, lds_of(32, a.K0), s, a);
    else hipLaunchKernelGGL((k_fwd3<2, false>), grid, block, lds_of(32, a.K0), s, a);
  }
  return hipGetLastError();
}

}  // namespace fdql
