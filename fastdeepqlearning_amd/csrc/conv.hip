// Implicit-GEMM convolutions of the pixel encoder (conv.h): persistent workgroups, image groups resident in LDS (LDS-DMA,
// double-buffered), stationary weights / stationary weight-gradient blocks, v_mfma_f32_16x16x4_f32 throughout.
//
// v_mfma_f32_16x16x4_f32 (32 cycles per SIMD, exact fp32): lane l supplies A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15]
// and ends with D[i = 4 (l >> 4) + r][j = l & 15] in register r.  Everywhere below the stationary / narrow side is the A operand
// (output channels) and the rows (output positions) are the B operand, so that a lane finishes with FOUR CONSECUTIVE CHANNELS
// of one position: a 16-byte NHWC store per tile and lane, no transpose.  One ds_read_b128 (four consecutive channels of the
// lane's position) feeds four MFMA steps - the weights are laid out so that step c of a 16-k group pairs k = 16 g + 4 (l >> 4) + c on
// both sides; the fp32 sum therefore runs over a fixed permutation of k.
#include "conv.h"

#include <algorithm>
#include <type_traits>

namespace fdql {
namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void *lds_vp;
typedef const __attribute__((address_space(1))) void *glb_vp;

constexpr int LDS_LIMIT = 160 * 1024;

template <int C_, int H_, int W_, int KS_, int S_, int CO_>
struct Geo {
  static constexpr int C = C_, H = H_, W = W_, KS = KS_, S = S_, CO = CO_;
  static constexpr int OH = (H - KS) / S + 1, OW = (W - KS) / S + 1, POS = OH * OW, K = KS * KS * C, IMGF = H * W * C;
  static_assert((H - KS) % S == 0 && (W - KS) % S == 0, "the windows must tile the map exactly");
};

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ v4f mfma4(float a, float b, v4f c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float lrelu(float v) { return fmaxf(v, 0.01f * v); }

// first byte of image i of a source (uniform: i and the source are)
__device__ __forceinline__ const char *image_ptr(const ConvSrc &s, long long i, long long imgbytes) {
  return (const char *)s.base + (s.slots ? (long long)s.slots[i] : i) * imgbytes;
}

// `bytes` (a multiple of 16) global -> LDS by LDS-DMA: 1 KiB per wave instruction, chunk c by wave (first + c) % NW.
// src / dst / bytes / first are wave-uniform.  Returns the chunk count (the caller rotates `first` by it).
template <int NW>
__device__ __forceinline__ int dma_bytes(const char *src, char *dst, int bytes, int first, int wave, int lane) {
  const int nch = (bytes + 1023) >> 10;
  int c = wave - (first % NW);
  if (c < 0) c += NW;
  for (; c < nch; c += NW) {
    const int off = c << 10;
    if (off + lane * 16 < bytes)
      __builtin_amdgcn_global_load_lds((glb_vp)(src + off + lane * 16), (lds_vp)(dst + off), 16, 0, 0);
  }
  return nch;
}
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <int NT>
__device__ __forceinline__ void lds_zero(float *lds, int floats, int tid) {
  for (int i = tid * 4; i < floats; i += NT * 4) *reinterpret_cast<v4f *>(lds + i) = v4f{0.f, 0.f, 0.f, 0.f};
}

// ================================================================================================ forward, float32 NHWC input
// Waves = (CO / 16 channel groups) x row groups.  A wave keeps the K weights of its 16 output channels in K / 4 registers and
// walks 16-row tiles of the image group, two at a time (independent accumulators: a dependent 16x16x4 needs 40 cycles, an
// issue slot is 32).
template <typename GE, int G, int NW>
struct FwdCfg {
  static constexpr int ROWS = G * GE::POS, NTILE = (ROWS + 15) / 16;
  static constexpr int NCG = GE::CO / 16, NRG = NW / NCG;
  static constexpr int BUF = G * GE::IMGF;                       // floats per image-group buffer
  static constexpr int LDS_FLOATS = 2 * BUF + NTILE * 16;
  static_assert(GE::CO % 16 == 0 && NW % NCG == 0 && GE::C % 16 == 0, "16 output channels per wave, 16 input channels per k-group");
  static_assert(LDS_FLOATS * 4 <= LDS_LIMIT, "image group does not fit the LDS");
};

// byte offset of 16-k group j4 inside a patch: k = (ky, kx, c), 16 consecutive channels of one tap
template <typename GE>
__host__ __device__ constexpr int koff_nhwc(int j4) {
  const int k0 = 16 * j4, cell = k0 / GE::C, c0 = k0 % GE::C;
  return (((cell / GE::KS) * GE::W + (cell % GE::KS)) * GE::C + c0) * 4;
}

template <typename GE, int G, int NW>
__global__ __launch_bounds__(NW * 64, 1) void k_conv_fwd(const ConvFwdArgs a) {
  using CF = FwdCfg<GE, G, NW>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int NT = NW * 64, K = GE::K, NJ4 = K / 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6), l16 = lane & 15, q = lane >> 4;
  const int cg = wave % CF::NCG, rg = wave / CF::NCG;
  int *rowoff = reinterpret_cast<int *>(lds + 2 * CF::BUF);
  lds_zero<NT>(lds, 2 * CF::BUF, tid);
  for (int r = tid; r < CF::NTILE * 16; r += NT) {
    int off = 0;
    if (r < CF::ROWS) {
      const int gi = r / GE::POS, p = r - gi * GE::POS, oy = p / GE::OW, ox = p - oy * GE::OW;
      off = (gi * GE::IMGF + (GE::S * oy * GE::W + GE::S * ox) * GE::C) * 4;
    }
    rowoff[r] = off;
  }
  // stationary weights: step j = 4 j4 + c of the K loop pairs k = 16 j4 + 4 q + c
  float w[K / 4];
  {
    const float *Wr = a.W + (long long)(16 * cg + l16) * K + 4 * q;
#pragma unroll
    for (int j4 = 0; j4 < NJ4; ++j4) {
      const v4f t = *reinterpret_cast<const v4f *>(Wr + 16 * j4);
      w[4 * j4] = t.x; w[4 * j4 + 1] = t.y; w[4 * j4 + 2] = t.z; w[4 * j4 + 3] = t.w;
    }
  }
  const v4f bias = *reinterpret_cast<const v4f *>(a.bias + 16 * cg + 4 * q);
  const long long ngroups = (a.nimg + G - 1) / G;
  const long long total_rows = a.nimg * GE::POS;
  const float *in = (const float *)a.in.base;
  __syncthreads();
  auto load_group = [&](long long grp, int buf) {
    int first = 0;
    for (int gi = 0; gi < G; ++gi) {
      const long long img = grp * G + gi;
      if (img < a.nimg)
        first += dma_bytes<NW>((const char *)(in + img * GE::IMGF), (char *)(lds + buf * CF::BUF + gi * GE::IMGF), GE::IMGF * 4, first, wave, lane);
    }
  };
  long long grp = blockIdx.x;
  if (grp < ngroups) load_group(grp, 0);
  for (int it = 0; grp < ngroups; grp += gridDim.x, ++it) {
    const int cur = it & 1;
    dma_wait_all();
    __syncthreads();
    if (grp + gridDim.x < ngroups) load_group(grp + gridDim.x, cur ^ 1);
    const char *bufb = reinterpret_cast<const char *>(lds + cur * CF::BUF) + q * 16;
    const long long row0 = grp * CF::ROWS;
    auto tiles = [&](auto twoc, int t0, int t1) __attribute__((always_inline)) {
      constexpr bool TWO = decltype(twoc)::value;
      const char *p0 = bufb + rowoff[16 * t0 + l16];
      const char *p1 = bufb + rowoff[16 * t1 + l16];
      v4f acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      v4f x0 = *reinterpret_cast<const v4f *>(p0 + koff_nhwc<GE>(0)), x1 = x0;
      if constexpr (TWO) x1 = *reinterpret_cast<const v4f *>(p1 + koff_nhwc<GE>(0));
#pragma unroll
      for (int j4 = 0; j4 < NJ4; ++j4) {
        // The group's first MFMAs, THEN the next group's reads, then the rest: hipcc waits for every outstanding LDS read at the
        // first use of a fragment (s_waitcnt lgkmcnt(0)), so reads issued ahead of that point would be waited for at once; issued
        // behind it they have six MFMAs to land.  The barriers keep the scheduler from sinking the reads to their use.
        acc0 = mfma4(w[4 * j4], x0[0], acc0);
        if constexpr (TWO) acc1 = mfma4(w[4 * j4], x1[0], acc1);
        __builtin_amdgcn_sched_barrier(0);
        v4f n0 = x0, n1 = x1;
        if (j4 + 1 < NJ4) {
          n0 = *reinterpret_cast<const v4f *>(p0 + koff_nhwc<GE>(j4 + 1));
          if constexpr (TWO) n1 = *reinterpret_cast<const v4f *>(p1 + koff_nhwc<GE>(j4 + 1));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 1; c < 4; ++c) {
          acc0 = mfma4(w[4 * j4 + c], x0[c], acc0);
          if constexpr (TWO) acc1 = mfma4(w[4 * j4 + c], x1[c], acc1);
        }
        x0 = n0; x1 = n1;
      }
      auto store = [&](int t, const v4f &acc) __attribute__((always_inline)) {
        const int r = 16 * t + l16;
        if (r < CF::ROWS && row0 + r < total_rows) {
          v4f y;
#pragma unroll
          for (int c = 0; c < 4; ++c) y[c] = lrelu(acc[c] + bias[c]);
          *reinterpret_cast<v4f *>(a.out + (row0 + r) * GE::CO + 16 * cg + 4 * q) = y;
        }
      };
      store(t0, acc0);
      if constexpr (TWO) store(t1, acc1);
    };
    for (int t0 = rg; t0 < CF::NTILE; t0 += 2 * CF::NRG) {
      const int t1 = t0 + CF::NRG;
      if (t1 < CF::NTILE) tiles(std::true_type{}, t0, t1);   // (uniform)
      else tiles(std::false_type{}, t0, t0);
    }
  }
}

// ================================================================================================ forward, uint8 NCHW frames
// K = (c, ky, kx) with 8-wide kernels: a 16-k group is two (c, ky) rows of 8 pixels; lane quad q reads the dword
// (row q / 2, pixels 4 (q % 2) .. + 3) and widens a byte per MFMA step.  Every wave holds ALL CO / 16 channel tiles (one
// v_cvt_f32_ubyte per CO / 16 MFMAs) and the waves split the row tiles.  1 / 255 is folded into the stationary weights.
template <typename GE, int G, int NW>
struct Fwd8Cfg {
  static constexpr int IMGB = GE::C * GE::H * GE::W;             // bytes per frame stack
  static constexpr int ROWS = G * GE::POS, NTILE = (ROWS + 15) / 16, NCO = GE::CO / 16;
  static constexpr int BUF = G * IMGB;                           // bytes
  static constexpr int LDS_BYTES = 2 * BUF + NTILE * 16 * 4;
  static_assert(GE::KS == 8 && GE::S % 4 == 0 && GE::W % 4 == 0 && IMGB % 16 == 0 && GE::CO % 16 == 0, "dword-aligned 8-pixel rows");
  static_assert(LDS_BYTES <= LDS_LIMIT, "image group does not fit the LDS");
};
template <typename GE>
__host__ __device__ constexpr int koff_u8(int g) {   // (c, ky) row 2 g of the patch
  return ((2 * g) / GE::KS) * GE::H * GE::W + ((2 * g) % GE::KS) * GE::W;
}

template <typename GE, int G, int NW>
__global__ __launch_bounds__(NW * 64, 1) void k_conv_fwd_u8(const ConvFwdArgs a) {
  using CF = Fwd8Cfg<GE, G, NW>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int NT = NW * 64, K = GE::K, NG = K / 16, NCO = CF::NCO;
  char *ldsb = reinterpret_cast<char *>(lds);
  const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6), l16 = lane & 15, q = lane >> 4;
  int *rowoff = reinterpret_cast<int *>(ldsb + 2 * CF::BUF);
  lds_zero<NT>(lds, 2 * CF::BUF / 4, tid);
  for (int r = tid; r < CF::NTILE * 16; r += NT) {
    int off = 0;
    if (r < CF::ROWS) {
      const int gi = r / GE::POS, p = r - gi * GE::POS, oy = p / GE::OW, ox = p - oy * GE::OW;
      off = gi * CF::IMGB + GE::S * oy * GE::W + GE::S * ox;
    }
    rowoff[r] = off;
  }
  float w[NCO][K / 4];
#pragma unroll
  for (int t = 0; t < NCO; ++t) {
    const float *Wr = a.W + (long long)(16 * t + l16) * K + 4 * q;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const v4f v = *reinterpret_cast<const v4f *>(Wr + 16 * g);
#pragma unroll
      for (int c = 0; c < 4; ++c) w[t][4 * g + c] = v[c] * (1.0f / 255.0f);
    }
  }
  v4f bias[NCO];
#pragma unroll
  for (int t = 0; t < NCO; ++t) bias[t] = *reinterpret_cast<const v4f *>(a.bias + 16 * t + 4 * q);
  const long long ngroups = (a.nimg + G - 1) / G;
  const long long total_rows = a.nimg * GE::POS;
  __syncthreads();
  auto load_group = [&](long long grp, int buf) {
    int first = 0;
    for (int gi = 0; gi < G; ++gi) {
      const long long img = grp * G + gi;
      if (img < a.nimg) first += dma_bytes<NW>(image_ptr(a.in, img, CF::IMGB), ldsb + buf * CF::BUF + gi * CF::IMGB, CF::IMGB, first, wave, lane);
    }
  };
  long long grp = blockIdx.x;
  if (grp < ngroups) load_group(grp, 0);
  const int lane_off = (q >> 1) * GE::W + 4 * (q & 1);
  for (int it = 0; grp < ngroups; grp += gridDim.x, ++it) {
    const int cur = it & 1;
    dma_wait_all();
    __syncthreads();
    if (grp + gridDim.x < ngroups) load_group(grp + gridDim.x, cur ^ 1);
    const char *bufb = ldsb + cur * CF::BUF + lane_off;
    const long long row0 = grp * CF::ROWS;
    auto tiles = [&](auto twoc, int t0, int t1) __attribute__((always_inline)) {
      constexpr bool TWO = decltype(twoc)::value;
      const char *p0 = bufb + rowoff[16 * t0 + l16];
      const char *p1 = bufb + rowoff[16 * t1 + l16];
      v4f acc0[NCO], acc1[NCO];
#pragma unroll
      for (int t = 0; t < NCO; ++t) acc0[t] = acc1[t] = v4f{0.f, 0.f, 0.f, 0.f};
      unsigned d0 = *reinterpret_cast<const unsigned *>(p0 + koff_u8<GE>(0)), d1 = d0;
      if constexpr (TWO) d1 = *reinterpret_cast<const unsigned *>(p1 + koff_u8<GE>(0));
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        float x0[4], x1[4];   // widened ahead of the MFMAs that take them (a conversion right in front of its MFMA costs wait states)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          x0[c] = (float)((d0 >> (8 * c)) & 255u);
          x1[c] = TWO ? (float)((d1 >> (8 * c)) & 255u) : 0.f;
        }
        __builtin_amdgcn_sched_barrier(0);   // (the next group's reads behind this group's wait: see k_conv_fwd)
        unsigned n0 = d0, n1 = d1;
        if (g + 1 < NG) {
          n0 = *reinterpret_cast<const unsigned *>(p0 + koff_u8<GE>(g + 1));
          if constexpr (TWO) n1 = *reinterpret_cast<const unsigned *>(p1 + koff_u8<GE>(g + 1));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int t = 0; t < NCO; ++t) {
            acc0[t] = mfma4(w[t][4 * g + c], x0[c], acc0[t]);
            if constexpr (TWO) acc1[t] = mfma4(w[t][4 * g + c], x1[c], acc1[t]);
          }
        d0 = n0; d1 = n1;
      }
      auto store = [&](int tt, const v4f (&acc)[NCO]) __attribute__((always_inline)) {
        const int r = 16 * tt + l16;
        if (r < CF::ROWS && row0 + r < total_rows) {
#pragma unroll
          for (int t = 0; t < NCO; ++t) {
            v4f y;
#pragma unroll
            for (int c = 0; c < 4; ++c) y[c] = lrelu(acc[t][c] + bias[t][c]);
            *reinterpret_cast<v4f *>(a.out + (row0 + r) * GE::CO + 16 * t + 4 * q) = y;
          }
        }
      };
      store(t0, acc0);
      if constexpr (TWO) store(t1, acc1);
    };
    for (int t0 = wave; t0 < CF::NTILE; t0 += 2 * NW) {
      const int t1 = t0 + NW;
      if (t1 < CF::NTILE) tiles(std::true_type{}, t0, t1);   // (uniform)
      else tiles(std::false_type{}, t0, t0);
    }
  }
}

// ================================================================================================ data gradient (gather form)
// dprev[y, x, c] = gate * sum over the taps (ky, kx) with y = S oy + ky, x = S ox + kx of dpre[oy, ox, :] . W[:, (ky, kx, c)].
// Pixels of one PARITY CLASS (y % S, x % S) share their tap set: class (py, px) is a stride-1 convolution with TA x TA taps
// (TA = KS / S) over the layer's dpre image padded by TA - 1 zero positions on every side - the padded image is what sits in
// LDS (rows DMA'd into place, the borders zeroed once), so out-of-range taps read zeros and no lane ever branches.
// A "combination" = (class, 16 input channels); a wave keeps the TA^2 CO weights of its combinations in registers.
template <typename GE, int G, int NW>
struct DgCfg {
  static constexpr int TA = GE::KS / GE::S;
  static constexpr int PH = GE::OH + 2 * (TA - 1), PW = GE::OW + 2 * (TA - 1), PIMG = PH * PW * GE::CO;
  static constexpr int HY = GE::OH - 1 + TA, HX = GE::OW - 1 + TA, CPOS = HY * HX;   // class grid (H = S HY exactly)
  static constexpr int ROWS = G * CPOS, NTILE = (ROWS + 15) / 16;
  static constexpr int KP = TA * TA * GE::CO;
  // combinations per wave (same class): as many as keep the stationary weights within ~150 registers; waves beyond the
  // combination groups split the row tiles (row groups)
  static constexpr int NCH = GE::C / 16, NCOMBO = GE::S * GE::S * NCH;
  static constexpr int NCW = (NCOMBO >= NW) ? NCOMBO / NW : 1;
  static constexpr int NCGRP = NCOMBO / NCW, NRG = NW / NCGRP;
  static constexpr int BUF = G * PIMG;
  static constexpr int LDS_FLOATS = 2 * BUF + 2 * NTILE * 16;
  static_assert(GE::KS % GE::S == 0 && GE::C % 16 == 0 && GE::CO % 16 == 0, "uniform tap sets, 16-channel groups");
  static_assert(NCOMBO % NCW == 0 && NW % NCGRP == 0 && NCH % NCW == 0, "a wave's combinations share one class");
  static_assert(GE::H == GE::S * HY && GE::W == GE::S * HX, "class grids tile the map");
  static_assert(LDS_FLOATS * 4 <= LDS_LIMIT, "padded image group does not fit the LDS");
};
template <typename GE, int G, int NW>
__host__ __device__ constexpr int koff_dg(int j4) {
  using CF = DgCfg<GE, G, NW>;
  const int k0 = 16 * j4, tap = k0 / GE::CO, co0 = k0 % GE::CO, ta = tap / CF::TA, tb = tap % CF::TA;
  return (((CF::TA - 1 - ta) * CF::PW + (CF::TA - 1 - tb)) * GE::CO + co0) * 4;
}

template <typename GE, int G, int NW>
__global__ __launch_bounds__(NW * 64, 1) void k_conv_dgrad(const ConvDgradArgs a, const float *__restrict__ act_prev, float *__restrict__ dprev) {
  // (act_prev / dprev = a.act_prev / a.dprev as restrict kernel parameters: without the promise that they are distinct a gate
  // request placed behind a store waits for that store to complete - s_waitcnt vmcnt(0) after every tile pair)
  using CF = DgCfg<GE, G, NW>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int NT = NW * 64, KP = CF::KP, NJ4 = KP / 16, NCW = CF::NCW, TA = CF::TA;
  const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6), l16 = lane & 15, q = lane >> 4;
  const int rg = wave / CF::NCGRP;
  const int combo0 = (wave % CF::NCGRP) * NCW, cls = combo0 / CF::NCH, chg0 = combo0 % CF::NCH, py = cls / GE::S, px = cls % GE::S;
  int *rowoff = reinterpret_cast<int *>(lds + 2 * CF::BUF), *outoff = rowoff + CF::NTILE * 16;
  lds_zero<NT>(lds, 2 * CF::BUF, tid);
  for (int r = tid; r < CF::NTILE * 16; r += NT) {
    int off = 0, oo = -1;
    if (r < CF::ROWS) {
      const int gi = r / CF::CPOS, p = r - gi * CF::CPOS, Y = p / CF::HX, X = p - Y * CF::HX;
      off = (gi * CF::PIMG + (Y * CF::PW + X) * GE::CO) * 4;
      oo = gi * GE::IMGF + (GE::S * Y * GE::W + GE::S * X) * GE::C;
    }
    rowoff[r] = off;
    outoff[r] = oo;
  }
  // stationary weights of combination i (channels 16 (chg0 + i) + l16): step j = 4 j4 + c pairs k' = 16 j4 + 4 q + c = (tap, co)
  float w[NCW][KP / 4];
#pragma unroll
  for (int i = 0; i < NCW; ++i) {
    const int ch = 16 * (chg0 + i) + l16;
#pragma unroll
    for (int j = 0; j < KP / 4; ++j) {
      const int kp = 16 * (j >> 2) + 4 * q + (j & 3);
      const int tap = kp / GE::CO, co = kp - tap * GE::CO, ta = tap / TA, tb = tap - ta * TA;
      w[i][j] = a.W[(long long)co * GE::K + ((py + GE::S * ta) * GE::KS + (px + GE::S * tb)) * GE::C + ch];
    }
  }
  const long long ngroups = (a.nimg + G - 1) / G;
  __syncthreads();
  // the OH rows of an image's dpre, each OW x CO floats, into the padded image
  auto load_group = [&](long long grp, int buf) {
    int first = 0;
    for (int gi = 0; gi < G; ++gi) {
      const long long img = grp * G + gi;
      if (img >= a.nimg) continue;
      for (int oy = 0; oy < GE::OH; ++oy)
        first += dma_bytes<NW>((const char *)(a.dpre + ((img * GE::OH + oy) * GE::OW) * GE::CO),
                               (char *)(lds + buf * CF::BUF + gi * CF::PIMG + ((oy + TA - 1) * CF::PW + (TA - 1)) * GE::CO), GE::OW * GE::CO * 4,
                               first, wave, lane);
    }
  };
  long long grp = blockIdx.x;
  if (grp < ngroups) load_group(grp, 0);
  const int cls_off = (py * GE::W + px) * GE::C + 16 * chg0 + 4 * q;
  for (int it = 0; grp < ngroups; grp += gridDim.x, ++it) {
    const int cur = it & 1;
    dma_wait_all();
    __syncthreads();
    if (grp + gridDim.x < ngroups) load_group(grp + gridDim.x, cur ^ 1);
    // a partial last group: the images it does not have must read as zero gradients - they do not contribute (their
    // results are not stored), but stale values would be multiplied for nothing; nothing to do
    const char *bufb = reinterpret_cast<const char *>(lds + cur * CF::BUF) + q * 16;
    const long long img0 = grp * G;
    const int valid = (int)((a.nimg - img0 < G ? a.nimg - img0 : G) * CF::CPOS);   // rows of this group that exist
    auto tiles = [&](auto twoc, int t0, int t1) __attribute__((always_inline)) {
      constexpr bool TWO = decltype(twoc)::value;
      const int r0 = 16 * t0 + l16, r1 = 16 * t1 + l16;
      const char *p0 = bufb + rowoff[r0], *p1 = bufb + rowoff[r1];
      const int o0 = outoff[r0], o1 = outoff[r1];
      const bool ok0 = r0 < valid, ok1 = TWO && r1 < valid;
      const long long g0 = img0 * GE::IMGF + (ok0 ? o0 : 0) + cls_off, g1 = img0 * GE::IMGF + (ok1 ? o1 : 0) + cls_off;
      v4f gate0[NCW], gate1[NCW];   // the gate references, requested before the K loop
#pragma unroll
      for (int i = 0; i < NCW; ++i) {
        gate0[i] = ok0 ? *reinterpret_cast<const v4f *>(act_prev + g0 + 16 * i) : v4f{0.f, 0.f, 0.f, 0.f};
        if constexpr (TWO) gate1[i] = ok1 ? *reinterpret_cast<const v4f *>(act_prev + g1 + 16 * i) : v4f{0.f, 0.f, 0.f, 0.f};
      }
      v4f acc0[NCW], acc1[NCW];
#pragma unroll
      for (int i = 0; i < NCW; ++i) acc0[i] = acc1[i] = v4f{0.f, 0.f, 0.f, 0.f};
      v4f x0 = *reinterpret_cast<const v4f *>(p0 + koff_dg<GE, G, NW>(0)), x1 = x0;
      if constexpr (TWO) x1 = *reinterpret_cast<const v4f *>(p1 + koff_dg<GE, G, NW>(0));
#pragma unroll
      for (int j4 = 0; j4 < NJ4; ++j4) {
        // (first MFMAs, then the next group's reads, then the rest: as in k_conv_fwd)
#pragma unroll
        for (int i = 0; i < NCW; ++i) {
          acc0[i] = mfma4(w[i][4 * j4], x0[0], acc0[i]);
          if constexpr (TWO) acc1[i] = mfma4(w[i][4 * j4], x1[0], acc1[i]);
        }
        __builtin_amdgcn_sched_barrier(0);
        v4f n0 = x0, n1 = x1;
        if (j4 + 1 < NJ4) {
          n0 = *reinterpret_cast<const v4f *>(p0 + koff_dg<GE, G, NW>(j4 + 1));
          if constexpr (TWO) n1 = *reinterpret_cast<const v4f *>(p1 + koff_dg<GE, G, NW>(j4 + 1));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 1; c < 4; ++c)
#pragma unroll
          for (int i = 0; i < NCW; ++i) {
            acc0[i] = mfma4(w[i][4 * j4 + c], x0[c], acc0[i]);
            if constexpr (TWO) acc1[i] = mfma4(w[i][4 * j4 + c], x1[c], acc1[i]);
          }
        x0 = n0; x1 = n1;
      }
      // every gated value is formed BEFORE the first store: with loads and stores both outstanding hipcc waits vmcnt(0) for
      // any load result, i.e. for the stores issued so far to complete
      v4f y0[NCW], y1[NCW];
#pragma unroll
      for (int i = 0; i < NCW; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          y0[i][c] = gate0[i][c] > 0.f ? acc0[i][c] : 0.01f * acc0[i][c];
          y1[i][c] = TWO ? (gate1[i][c] > 0.f ? acc1[i][c] : 0.01f * acc1[i][c]) : 0.f;
        }
#pragma unroll
      for (int i = 0; i < NCW; ++i) asm volatile("" : "+v"(y0[i]), "+v"(y1[i]));   // (materialised here: not sunk into the store branches)
#pragma unroll
      for (int i = 0; i < NCW; ++i) {
        if (ok0) *reinterpret_cast<v4f *>(dprev + g0 + 16 * i) = y0[i];
        if constexpr (TWO) {
          if (ok1) *reinterpret_cast<v4f *>(dprev + g1 + 16 * i) = y1[i];
        }
      }
    };
    for (int t0 = rg; t0 < CF::NTILE; t0 += 2 * CF::NRG) {
      if (t0 + CF::NRG < CF::NTILE) tiles(std::true_type{}, t0, t0 + CF::NRG);   // (uniform)
      else tiles(std::false_type{}, t0, t0);
    }
  }
}

// ================================================================================================ weight gradient, NHWC input
// Output-stationary: dW[co, k] = sum over rows of dpre[row, co] patch[row, k].  One MFMA takes 4 rows: A = dpre^T (16 channels x
// 4 rows), B = patch (4 rows x 16 k).  Waves = KS (kernel row ky: the KS C floats of a patch row are one contiguous run of
// NR = KS C / 64 64-float pieces) x CGW (halves of the output channels) x RG (row groups: each writes a slab of its own).
// A lane's 16-byte reads give four CONSECUTIVE channels (A: four channel tiles) / k (B: four k tiles of a 64-run), so a step
// is 1 + NR LDS reads for 4 NR NCT MFMAs; the tiles are stored as float4 over the four k tiles.
template <typename GE_, int G_, int CGW_, int RG_>
struct WgCfg {
  using GE = GE_;
  static constexpr int G = G_, CGW = CGW_, RG = RG_;
  static constexpr int KG = GE::KS, NW = KG * CGW * RG, NT = NW * 64;
  static constexpr int NR = GE::KS * GE::C / 64;              // 64-runs per wave
  static constexpr int NCT = GE::CO / 16 / CGW;               // channel tiles per wave
  static constexpr int ROWS = G * GE::POS, NSTEP = (ROWS + 3) / 4, ROWSP = NSTEP * 4;
  static constexpr int IN_F = G * GE::IMGF, DP_F = ROWSP * GE::CO;
  static constexpr int BUF = IN_F + DP_F;
  static constexpr int LDS_FLOATS = 2 * BUF + ROWSP;
  static_assert((GE::KS * GE::C) % 64 == 0 && (NCT == 4 || NCT == 2) && GE::CO % (16 * CGW) == 0, "64-float runs per kernel row");
  static_assert(LDS_FLOATS * 4 <= LDS_LIMIT, "image group does not fit the LDS");
  static_assert(NT <= 1024, "workgroup size");
};

template <typename CF>
__global__ __launch_bounds__(CF::NT, 1) void k_conv_wgrad(const ConvWgradArgs a) {
  using GE = typename CF::GE;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int G = CF::G, CGW = CF::CGW, RG = CF::RG;
  constexpr int NT = CF::NT, NW = CF::NW, NR = CF::NR, NCT = CF::NCT, K = GE::K, CO = GE::CO;
  const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6), l16 = lane & 15, q = lane >> 4;
  const int kg = wave % CF::KG, cgw = (wave / CF::KG) % CGW, rg = wave / (CF::KG * CGW);
  int *rowoff = reinterpret_cast<int *>(lds + 2 * CF::BUF);
  lds_zero<NT>(lds, 2 * CF::BUF, tid);
  for (int r = tid; r < CF::ROWSP; r += NT) {
    int off = 0;
    if (r < CF::ROWS) {
      const int gi = r / GE::POS, p = r - gi * GE::POS, oy = p / GE::OW, ox = p - oy * GE::OW;
      off = (gi * GE::IMGF + (GE::S * oy * GE::W + GE::S * ox) * GE::C) * 4;
    }
    rowoff[r] = off;
  }
  v4f acc[NCT][NR][4];
#pragma unroll
  for (int t = 0; t < NCT; ++t)
#pragma unroll
    for (int rr = 0; rr < NR; ++rr)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[t][rr][j] = v4f{0.f, 0.f, 0.f, 0.f};
  float bsum[NCT];
#pragma unroll
  for (int t = 0; t < NCT; ++t) bsum[t] = 0.f;
  const long long ngroups = (a.nimg + G - 1) / G;
  const float *in = (const float *)a.in.base;
  __syncthreads();
  auto load_group = [&](long long grp, int buf) {
    int first = 0;
    for (int gi = 0; gi < G; ++gi) {
      const long long img = grp * G + gi;
      if (img >= a.nimg) continue;
      first += dma_bytes<NW>((const char *)(in + img * GE::IMGF), (char *)(lds + buf * CF::BUF + gi * GE::IMGF), GE::IMGF * 4, first, wave, lane);
      first += dma_bytes<NW>((const char *)(a.dpre + img * GE::POS * CO), (char *)(lds + buf * CF::BUF + CF::IN_F + gi * GE::POS * CO), GE::POS * CO * 4,
                             first, wave, lane);
    }
  };
  long long grp = blockIdx.x;
  if (grp < ngroups) load_group(grp, 0);
  const int b_lane = kg * GE::W * GE::C * 4 + l16 * 16;                 // this wave's kernel row, the lane's four k of a 64-run
  const int a_lane = (cgw * 16 * NCT + NCT * l16) * 4;                  // the lane's NCT consecutive output channels
  for (int it = 0; grp < ngroups; grp += gridDim.x, ++it) {
    const int cur = it & 1;
    dma_wait_all();
    __syncthreads();
    const long long img0 = grp * G;
    const int have = (int)(a.nimg - img0 < G ? a.nimg - img0 : G);
    if (have < G) {   // (uniform; the last group only) images it does not have contribute nothing: zero their dpre rows
      float *dp = lds + cur * CF::BUF + CF::IN_F;
      for (int i = have * GE::POS * CO + tid; i < G * GE::POS * CO; i += NT) dp[i] = 0.f;
      __syncthreads();
    }
    if (grp + gridDim.x < ngroups) load_group(grp + gridDim.x, cur ^ 1);
    const char *inb = reinterpret_cast<const char *>(lds + cur * CF::BUF) + b_lane;
    const char *dpb = reinterpret_cast<const char *>(lds + cur * CF::BUF + CF::IN_F) + a_lane;
    // steps rg, rg + RG, ...: the operands of the next step are requested before this step's MFMAs, its patch offset one step
    // earlier still (the patch read depends on it)
    float av[NCT], an[NCT];
    v4f bv[NR], bn[NR];
    auto read_ops = [&](int st, int ro, float (&ao)[NCT], v4f (&bo)[NR]) __attribute__((always_inline)) {
      const int row = 4 * st + q;
      if constexpr (NCT == 4) {
        const v4f t = *reinterpret_cast<const v4f *>(dpb + row * CO * 4);
        ao[0] = t.x; ao[1] = t.y; ao[2] = t.z; ao[3] = t.w;
      } else {
        const v2f t = *reinterpret_cast<const v2f *>(dpb + row * CO * 4);
        ao[0] = t.x; ao[1] = t.y;
      }
#pragma unroll
      for (int rr = 0; rr < NR; ++rr) bo[rr] = *reinterpret_cast<const v4f *>(inb + ro + rr * 256);
    };
    int st = rg;
    int ro_n = 0;
    if (st < CF::NSTEP) {
      read_ops(st, rowoff[4 * st + q], av, bv);
      if (st + RG < CF::NSTEP) ro_n = rowoff[4 * (st + RG) + q];
    }
    for (; st < CF::NSTEP; st += RG) {
      const bool more = st + RG < CF::NSTEP;   // (uniform)
      int ro_nn = 0;
      // (this step's first MFMAs - where hipcc waits for every outstanding LDS read - then the next step's reads, then the rest)
#pragma unroll
      for (int t = 0; t < NCT; ++t) acc[t][0][0] = mfma4(av[t], bv[0][0], acc[t][0][0]);
      __builtin_amdgcn_sched_barrier(0);
      if (more) {
        read_ops(st + RG, ro_n, an, bn);
        if (st + 2 * RG < CF::NSTEP) ro_nn = rowoff[4 * (st + 2 * RG) + q];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int rr = 0; rr < NR; ++rr)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int t = 0; t < NCT; ++t)
            if (rr + j > 0) acc[t][rr][j] = mfma4(av[t], bv[rr][j], acc[t][rr][j]);
#pragma unroll
      for (int t = 0; t < NCT; ++t) bsum[t] += av[t];
      if (more) {
#pragma unroll
        for (int t = 0; t < NCT; ++t) av[t] = an[t];
#pragma unroll
        for (int rr = 0; rr < NR; ++rr) bv[rr] = bn[rr];
        ro_n = ro_nn;
      }
    }
  }
  // ---- this wave's block of slab (workgroup, row group): dW[co][k] with co = 16 NCT cgw + NCT (4 q + r) + t, k = kg KS C + 64 rr + 4 l16 + j
  float *slab = a.wpart + ((long long)blockIdx.x * RG + rg) * ((long long)CO * K + CO);
#pragma unroll
  for (int t = 0; t < NCT; ++t)
#pragma unroll
    for (int rr = 0; rr < NR; ++rr)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = cgw * 16 * NCT + NCT * (4 * q + r) + t;
        const v4f v = {acc[t][rr][0][r], acc[t][rr][1][r], acc[t][rr][2][r], acc[t][rr][3][r]};
        *reinterpret_cast<v4f *>(slab + (long long)co * K + kg * (GE::KS * GE::C) + 64 * rr + 4 * l16) = v;
      }
  if (kg == 0) {   // (uniform) bias gradient: the lane's channels summed over its rows; the four row quads of a wave added here
#pragma unroll
    for (int t = 0; t < NCT; ++t) {
      float s = bsum[t];
      s += __shfl_xor(s, 16);
      s += __shfl_xor(s, 32);
      if (q == 0) slab[(long long)CO * K + cgw * 16 * NCT + NCT * l16 + t] = s;
    }
  }
}

// ================================================================================================ weight gradient, uint8 frames
// K = (c, ky, kx), 8-wide kernels: the 64 k of one input channel are the 16 dwords (ky = l16 / 2, pixels 4 (l16 % 2) .. + 3) of
// its 8 x 8 window - a dword read gives four k tiles (byte b -> k = 64 c + 4 l16 + b).  Every wave holds the WHOLE dW
// (CO / 16 x 4 C tiles) and the waves split the rows; one slab per wave.  dpre . (pixel / 255): the 1 / 255 is applied once, at the end.
template <typename GE, int NW>
struct Wg8Cfg {
  static constexpr int IMGB = GE::C * GE::H * GE::W;
  static constexpr int NCT = GE::CO / 16;
  static constexpr int NSTEP = (GE::POS + 3) / 4, ROWSP = NSTEP * 4;
  static constexpr int DP_F = ROWSP * GE::CO;
  static constexpr int BUF = IMGB + DP_F * 4;                    // bytes (one image per group)
  static constexpr int LDS_BYTES = 2 * BUF + ROWSP * 4;
  static_assert(GE::KS == 8 && GE::S % 4 == 0 && GE::W % 4 == 0 && IMGB % 16 == 0 && NCT == 2, "dword-aligned 8-pixel rows, 32 output channels");
  static_assert(LDS_BYTES <= LDS_LIMIT, "image does not fit the LDS");
};

template <typename GE, int NW>
__global__ __launch_bounds__(NW * 64, 1) void k_conv_wgrad_u8(const ConvWgradArgs a) {
  using CF = Wg8Cfg<GE, NW>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int NT = NW * 64, NCT = CF::NCT, C = GE::C, K = GE::K, CO = GE::CO;
  char *ldsb = reinterpret_cast<char *>(lds);
  const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6), l16 = lane & 15, q = lane >> 4;
  int *rowoff = reinterpret_cast<int *>(ldsb + 2 * CF::BUF);
  lds_zero<NT>(lds, 2 * CF::BUF / 4, tid);
  for (int r = tid; r < CF::ROWSP; r += NT) {
    int off = 0;
    if (r < GE::POS) {
      const int oy = r / GE::OW, ox = r - oy * GE::OW;
      off = GE::S * oy * GE::W + GE::S * ox;
    }
    rowoff[r] = off;
  }
  v4f acc[NCT][C][4];
#pragma unroll
  for (int t = 0; t < NCT; ++t)
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[t][c][b] = v4f{0.f, 0.f, 0.f, 0.f};
  float bsum[NCT];
#pragma unroll
  for (int t = 0; t < NCT; ++t) bsum[t] = 0.f;
  __syncthreads();
  auto load_group = [&](long long img, int buf) {
    int first = dma_bytes<NW>(image_ptr(a.in, img, CF::IMGB), ldsb + buf * CF::BUF, CF::IMGB, 0, wave, lane);
    dma_bytes<NW>((const char *)(a.dpre + img * GE::POS * CO), ldsb + buf * CF::BUF + CF::IMGB, GE::POS * CO * 4, first, wave, lane);
  };
  long long img = blockIdx.x;
  if (img < a.nimg) load_group(img, 0);
  const int b_lane = (l16 >> 1) * GE::W + 4 * (l16 & 1);
  for (int it = 0; img < a.nimg; img += gridDim.x, ++it) {
    const int cur = it & 1;
    dma_wait_all();
    __syncthreads();
    if (img + gridDim.x < a.nimg) load_group(img + gridDim.x, cur ^ 1);
    const char *inb = ldsb + cur * CF::BUF + b_lane;
    const char *dpb = ldsb + cur * CF::BUF + CF::IMGB + 2 * l16 * 4;
    v2f av, an = {0.f, 0.f};
    unsigned d[C], dn[C];
    auto read_ops = [&](int st, int ro, v2f &ao, unsigned (&dd)[C]) __attribute__((always_inline)) {
      ao = *reinterpret_cast<const v2f *>(dpb + (4 * st + q) * CO * 4);
#pragma unroll
      for (int c = 0; c < C; ++c) dd[c] = *reinterpret_cast<const unsigned *>(inb + ro + c * GE::H * GE::W);
    };
    int st = wave;
    int ro_n = 0;
    if (st < CF::NSTEP) {
      read_ops(st, rowoff[4 * st + q], av, d);
      if (st + NW < CF::NSTEP) ro_n = rowoff[4 * (st + NW) + q];
    }
    for (; st < CF::NSTEP; st += NW) {
      const bool more = st + NW < CF::NSTEP;   // (uniform)
      int ro_nn = 0;
      float x[C][4];
#pragma unroll
      for (int c = 0; c < C; ++c)
#pragma unroll
        for (int b = 0; b < 4; ++b) x[c][b] = (float)((d[c] >> (8 * b)) & 255u);
      __builtin_amdgcn_sched_barrier(0);   // (the next step's reads behind this step's wait)
      if (more) {
        read_ops(st + NW, ro_n, an, dn);
        if (st + 2 * NW < CF::NSTEP) ro_nn = rowoff[4 * (st + 2 * NW) + q];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < C; ++c)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          acc[0][c][b] = mfma4(av.x, x[c][b], acc[0][c][b]);
          acc[1][c][b] = mfma4(av.y, x[c][b], acc[1][c][b]);
        }
      bsum[0] += av.x;
      bsum[1] += av.y;
      if (more) {
        av = an;
#pragma unroll
        for (int c = 0; c < C; ++c) d[c] = dn[c];
        ro_n = ro_nn;
      }
    }
  }
  // ---- slab (workgroup, wave): dW[co = 2 (4 q + r) + t][k = 64 c + 4 l16 + b]
  float *slab = a.wpart + ((long long)blockIdx.x * NW + wave) * ((long long)CO * K + CO);
#pragma unroll
  for (int t = 0; t < NCT; ++t)
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = 2 * (4 * q + r) + t;
        const float s = 1.0f / 255.0f;
        const v4f v = {acc[t][c][0][r] * s, acc[t][c][1][r] * s, acc[t][c][2][r] * s, acc[t][c][3][r] * s};
        *reinterpret_cast<v4f *>(slab + (long long)co * K + 64 * c + 4 * l16) = v;
      }
#pragma unroll
  for (int t = 0; t < NCT; ++t) {
    float s = bsum[t];
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    if (q == 0) slab[(long long)CO * K + 2 * l16 + t] = s;
  }
}

// ================================================================================================ instantiations and launchers
// BASELINE config 5's stack (the Atari encoder on 4 x 84 x 84 frame stacks): 32 x 8 / 4, 64 x 4 / 2, 64 x 3 / 1
using L0 = Geo<4, 84, 84, 8, 4, 32>;
using L1 = Geo<32, 20, 20, 4, 2, 64>;
using L2 = Geo<64, 9, 9, 3, 1, 64>;

template <typename GE>
bool geo_is(const ConvGeom &g, int cout) {
  return g.C == GE::C && g.H == GE::H && g.W == GE::W && g.k == GE::KS && g.s == GE::S && cout == GE::CO;
}

int num_cus() {
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) ncu = p.multiProcessorCount;
    if (ncu <= 0) ncu = 256;
  }
  return ncu;
}

template <typename KernelT, typename... ArgsT>
hipError_t launch_persistent(KernelT kernel, long long units, int threads, int lds_bytes, hipStream_t s, ArgsT... args) {
  if (units <= 0) return hipSuccess;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  if (e != hipSuccess) return e;
  const int blocks = (int)std::min<long long>(units, cu_budget(num_cus()));
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), lds_bytes, s, args...);
  return hipGetLastError();
}

// images per group (forward / data gradient / weight gradient): rows per group close to a multiple of 16 (4), double-buffered in 160 KB
constexpr int G_F0 = 2, G_F1 = 1, G_F2 = 3, G_D1 = 2, G_D2 = 2, G_W1 = 1, G_W2 = 2;
#ifndef CONV_NW_F0
#define CONV_NW_F0 4
#endif
#ifndef CONV_NW_F1
#define CONV_NW_F1 8
#endif
#ifndef CONV_NW_F2
#define CONV_NW_F2 8
#endif
#ifndef CONV_NW_D1
#define CONV_NW_D1 8
#endif
#ifndef CONV_NW_D2
#define CONV_NW_D2 8
#endif
#ifndef CONV_NW_W0
#define CONV_NW_W0 8
#endif
constexpr int NW_F0 = CONV_NW_F0, NW_F1 = CONV_NW_F1, NW_F2 = CONV_NW_F2, NW_D1 = CONV_NW_D1, NW_D2 = CONV_NW_D2, NW_W0 = CONV_NW_W0;
using W1Cfg = WgCfg<L1, G_W1, 1, 2>;   // 4 kernel rows x 2 row groups = 8 waves
using W2Cfg = WgCfg<L2, G_W2, 2, 2>;   // 3 kernel rows x 2 channel halves x 2 row groups = 12 waves

}  // namespace

bool conv_fwd_takes(const ConvGeom &g, int cout, bool u8) {
  if (!plan_switches().implicit_conv) return false;
  return u8 ? geo_is<L0>(g, cout) : (geo_is<L1>(g, cout) || geo_is<L2>(g, cout));
}
bool conv_dgrad_takes(const ConvGeom &g, int cout) {
  if (!plan_switches().implicit_conv) return false;
  return geo_is<L1>(g, cout) || geo_is<L2>(g, cout);
}
bool conv_wgrad_takes(const ConvGeom &g, int cout, bool u8) { return conv_fwd_takes(g, cout, u8); }

int conv_wgrad_slabs(const ConvGeom &g, int cout, bool u8, long long nimg) {
  if (!conv_wgrad_takes(g, cout, u8)) return 0;
  const long long ncu = cu_budget(num_cus());
  if (u8) return (int)std::min<long long>(nimg, ncu) * NW_W0;
  if (geo_is<L1>(g, cout)) return (int)std::min<long long>((nimg + G_W1 - 1) / G_W1, ncu) * W1Cfg::RG;
  return (int)std::min<long long>((nimg + G_W2 - 1) / G_W2, ncu) * W2Cfg::RG;
}

hipError_t conv_fwd_launch(const ConvFwdArgs &a, hipStream_t s) {
  if (a.in.u8 && geo_is<L0>(a.g, a.cout))
    return launch_persistent(k_conv_fwd_u8<L0, G_F0, NW_F0>, (a.nimg + G_F0 - 1) / G_F0, NW_F0 * 64, Fwd8Cfg<L0, G_F0, NW_F0>::LDS_BYTES, s, a);
  if (!a.in.u8 && geo_is<L1>(a.g, a.cout))
    return launch_persistent(k_conv_fwd<L1, G_F1, NW_F1>, (a.nimg + G_F1 - 1) / G_F1, NW_F1 * 64, FwdCfg<L1, G_F1, NW_F1>::LDS_FLOATS * 4, s, a);
  if (!a.in.u8 && geo_is<L2>(a.g, a.cout))
    return launch_persistent(k_conv_fwd<L2, G_F2, NW_F2>, (a.nimg + G_F2 - 1) / G_F2, NW_F2 * 64, FwdCfg<L2, G_F2, NW_F2>::LDS_FLOATS * 4, s, a);
  return hipErrorInvalidValue;
}

hipError_t conv_dgrad_launch(const ConvDgradArgs &a, hipStream_t s) {
  if (geo_is<L1>(a.g, a.cout))
    return launch_persistent(k_conv_dgrad<L1, G_D1, NW_D1>, (a.nimg + G_D1 - 1) / G_D1, NW_D1 * 64, DgCfg<L1, G_D1, NW_D1>::LDS_FLOATS * 4, s, a, a.act_prev, a.dprev);
  if (geo_is<L2>(a.g, a.cout))
    return launch_persistent(k_conv_dgrad<L2, G_D2, NW_D2>, (a.nimg + G_D2 - 1) / G_D2, NW_D2 * 64, DgCfg<L2, G_D2, NW_D2>::LDS_FLOATS * 4, s, a, a.act_prev, a.dprev);
  return hipErrorInvalidValue;
}

hipError_t conv_wgrad_launch(const ConvWgradArgs &a, hipStream_t s) {
  if (a.in.u8 && geo_is<L0>(a.g, a.cout))
    return launch_persistent(k_conv_wgrad_u8<L0, NW_W0>, a.nimg, NW_W0 * 64, Wg8Cfg<L0, NW_W0>::LDS_BYTES, s, a);
  if (!a.in.u8 && geo_is<L1>(a.g, a.cout))
    return launch_persistent(k_conv_wgrad<W1Cfg>, (a.nimg + G_W1 - 1) / G_W1, W1Cfg::NT, W1Cfg::LDS_FLOATS * 4, s, a);
  if (!a.in.u8 && geo_is<L2>(a.g, a.cout))
    return launch_persistent(k_conv_wgrad<W2Cfg>, (a.nimg + G_W2 - 1) / G_W2, W2Cfg::NT, W2Cfg::LDS_FLOATS * 4, s, a);
  return hipErrorInvalidValue;
}

}  // namespace fdql
