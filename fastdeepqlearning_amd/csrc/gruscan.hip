// GRU joiner scans as ONE persistent launch each (franQ/Agent/components/encoder.py:40-42, 78-94: nn.GRU over the T axis).
//
// The step-by-step form (agent.hip: T x (recurrent GEMM + gate kernel), ~200 dependent launches of ~10 us) is latency-bound;
// the recurrence itself is independent across the batch.  Here a workgroup owns R = 4 RT batch rows for the WHOLE scan - no
// synchronisation between workgroups - and per time step multiplies its rows by the recurrent matrix
//     forward :  gh[R, 3L] = h_{t-1}[R, L] W_hh^T              (K = L,  N = 3L; then the gate math, h_t stays in LDS)
//     backward:  dh_{t-1}[R, L] += dgh_t[R, 3L] W_hh           (K = 3L, N = L;  the gate backward in front of it)
// as v_mfma_f32_4x4x1_16b_f32 with the 4 rows as the B operand (the same for all 16 blocks) and 64 weight rows as the A
// operand: wave w owns hidden units [64 w, 64 w + 64) - for all three gates in the forward scan - and a lane ends with 4
// consecutive units of one row per accumulator: the gate math, the state and every store are float4 per lane, in registers.
// The weights are STREAMED: every step needs all of W (768 KB at L = 256), K-contiguous per output unit (the backward scan
// reads a transposed copy), so a lane's A operand for 4 consecutive k is one 16-byte load from its own row.  With one wave per
// SIMD nothing hides that latency but the wave itself: each wave keeps a REGISTER ring of 8 batches x 3 slots (a slot = the 64
// rows' 4 k of one accumulator = 1 KiB per wave, 96 VGPRs in all) in flight - a batch is refilled for 8 batches ahead right
// after its MFMAs - and the stream never drains: the addresses repeat every step, so the ring runs on across step boundaries.
// (An LDS-DMA ring of the same depth ran at 55 GB/s per CU - 14 us per step: a wave's DMA cadence is ~100 cycles per KiB.)
// Per step and workgroup: 3 L^2 MACs x R rows on one CU against 3 L^2 x 4 bytes of weights through its L2 port - about
// 2.6 us of MFMA (R = 4) / 5.2 us (R = 8) against ~5 us of weight stream at L = 256: both forms take ~5-6 us per step, 64 / 32
// workgroups (B = 256); the step-by-step form took ~20 us.
#include <mutex>

#include "update_kernels.h"

namespace fdql {
namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) v4f *gcf4;
typedef __attribute__((address_space(1))) v4f *gf4;

constexpr int GS_DEPTH = 8;   // batches of 3 slots (1 KiB per wave each) in flight per wave: a REGISTER ring of 24 float4 per lane (16: no faster)

// Gate non-linearities in float32 (libm-accurate expf / tanhf, <= 1-2 ulp): the step-by-step kernels evaluate them through
// double, but here ONE wave per SIMD does the gate math of its rows between two K loops - 12 double-precision transcendentals
// per lane and step were ~4 us of every 8 us step.  The difference (a few 1e-7 per step) is inside the GRU cases' tolerance.
__device__ __forceinline__ float gs_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float gs_tanh(float x) { return tanhf(x); }

// Weight stream of one wave: slot q of a step = (k4 = q / NG, accumulator g = q % NG): the 4 k starting at 4 k4 of rows
// row0(g) + lane.  wbase: the wave's first weight row; rows of accumulator g start g * gstride rows further; row pitch K.
// The weights are PRE-PACKED in slot order (k_gru_pack, once per update: 2 x 768 KB): the wave's stream is one contiguous run
// of 1 KiB slots, so every request reads 1 KiB of consecutive memory.  (Unpacked - a lane's 16 bytes from its own weight row,
// 64 cache lines per request - the stream ran at 30 GB/s per CU: 25 us per step.)

// ---------------------------------------------------------------------------------------------------------------------
// forward: NG = 3 accumulators per row tile (r, z, n pre-activations of the wave's 64 units)
// ---------------------------------------------------------------------------------------------------------------------
template <int RT>
__global__ __launch_bounds__(256, 1) void k_gru_scan_fwd(GruScanArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int L = a.L, B = a.B, T = a.T, L3 = 3 * L;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int R = 4 * RT;
  const int HP = L + 4;                                   // pitch of the state tile (floats)
  float *hs = lds;                                        // [2][R][HP]: h_{t-1} tile, double-buffered over t
  const int r0 = blockIdx.x * R;
  const int j = lane & 3, bq = lane >> 2;                 // this lane: row j of every row tile, units u0 .. u0 + 3
  const int u0 = 64 * wave + 4 * bq;
  // bias of this lane's units, per gate
  v4f bias[3];
#pragma unroll
  for (int g = 0; g < 3; ++g) bias[g] = *(gcf4)(a.bhh + g * L + u0);
  // start state -> LDS image 0 (rows past the batch: row B - 1 again, their results are never stored)
  for (int e = tid; e < R * (L / 4); e += blockDim.x) {
    const int r = e / (L / 4), c4 = e - r * (L / 4);
    const int row = r0 + r < B ? r0 + r : B - 1;
    *reinterpret_cast<v4f *>(hs + r * HP + 4 * c4) = *(gcf4)(a.h0 + (long long)row * L + 4 * c4);
  }
  const int Q = 3 * (L / 4);                              // slots per step: batch k4 = its three gates' slots
  gcf4 stream = (gcf4)(a.W + (long long)wave * Q * 256) + lane;   // packed forward stream (k_gru_pack): slot q at stream[64 q]
  v4f wr[GS_DEPTH][3];                                    // the ring: batch k4 lives in wr[k4 % 8]
  int qnext = 0;                                          // first slot of the next batch to request
#pragma unroll
  for (int u = 0; u < GS_DEPTH; ++u) {
#pragma unroll
    for (int g = 0; g < 3; ++g) wr[u][g] = stream[64 * (qnext + g)];
    qnext = qnext + 3 == Q ? 0 : qnext + 3;
  }
  __syncthreads();                                        // the start-state tile is complete
#pragma unroll 1
  for (int t = 0; t < T; ++t) {
    const float *hcur = hs + (t & 1) * R * HP;
    float *hnxt = hs + ((t + 1) & 1) * R * HP;
    const float *hrow = hcur + j * HP;
    // this step's input pre-activations (requested now, long complete when the K loop ends)
    v4f gi[RT][3];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int row = r0 + 4 * rt + j < B ? r0 + 4 * rt + j : B - 1;
#pragma unroll
      for (int g = 0; g < 3; ++g) gi[rt][g] = *(gcf4)(a.gi + ((long long)t * B + row) * L3 + g * L + u0);
    }
    v4f acc[RT][3];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int g = 0; g < 3; ++g) acc[rt][g] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int k4o = 0; k4o < L / 4; k4o += GS_DEPTH) {
#pragma unroll
      for (int u = 0; u < GS_DEPTH; ++u) {
        v4f hq[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) hq[rt] = *reinterpret_cast<const v4f *>(hrow + 4 * rt * HP + 4 * (k4o + u));
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[rt][g] = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[u][g][c], hq[rt][c], acc[rt][g], 0, 0, 0);
        // refill this ring position: the batch 8 further down the stream (wrapping into the next step: same weights)
#pragma unroll
        for (int g = 0; g < 3; ++g) wr[u][g] = stream[64 * (qnext + g)];
        qnext = qnext + 3 == Q ? 0 : qnext + 3;
      }
    }
    // ---- gates (gru cell, torch gate order r, z, n), this lane's 4 units of row 4 rt + j
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int row = r0 + 4 * rt + j;
      const v4f hp = *reinterpret_cast<const v4f *>(hcur + (4 * rt + j) * HP + u0);
      v4f hr, hz, hn, hnew;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
#pragma clang fp contract(off)
        hr[c] = acc[rt][0][c] + bias[0][c];
        hz[c] = acc[rt][1][c] + bias[1][c];
        hn[c] = acc[rt][2][c] + bias[2][c];
        const float r = gs_sigmoid(gi[rt][0][c] + hr[c]);
        const float z = gs_sigmoid(gi[rt][1][c] + hz[c]);
        const float n = gs_tanh(gi[rt][2][c] + r * hn[c]);
        hnew[c] = (1.f - z) * n + z * hp[c];
      }
      *reinterpret_cast<v4f *>(hnxt + (4 * rt + j) * HP + u0) = hnew;
      if (row < B) {
        const long long m = (long long)t * B + row;
        *(gf4)(a.gh + m * L3 + u0) = hr;
        *(gf4)(a.gh + m * L3 + L + u0) = hz;
        *(gf4)(a.gh + m * L3 + 2 * L + u0) = hn;
        *(gf4)(a.state + m * L + u0) = hnew;
        *(gf4)(a.hprev + m * L + u0) = hp;
      }
    }
    __syncthreads();   // h_t complete in the other image before any wave's next K loop reads it
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward: rows of time step t = T - 2 .. 0; NG = 1 accumulator per row tile (d h_{t-1} through W_hh), K = 3 L
// ---------------------------------------------------------------------------------------------------------------------
template <int RT>
__global__ __launch_bounds__(256, 1) void k_gru_scan_bwd(GruScanArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int L = a.L, B = a.B, T = a.T, L3 = 3 * L;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int R = 4 * RT;
  const int GP = L3 + 4;                                  // pitch of the d gh tile
  float *gs = lds;                                        // [R][GP]: d gh of the current step (the B operand)
  const int r0 = blockIdx.x * R;
  const int j = lane & 3, bq = lane >> 2;
  const int u0 = 64 * wave + 4 * bq;
  const int Q = L3 / 4;                                   // slots per step; a batch = three consecutive k4
  gcf4 stream = (gcf4)(a.W + (long long)wave * Q * 256) + lane;   // packed backward stream: W_hh^T rows of the wave's units, K = 3L
  v4f wr[GS_DEPTH][3];
  int qnext = 0;
#pragma unroll
  for (int u = 0; u < GS_DEPTH; ++u) {
#pragma unroll
    for (int i = 0; i < 3; ++i) wr[u][i] = stream[64 * (qnext + i)];
    qnext = qnext + 3 == Q ? 0 : qnext + 3;
  }
  const float *grow_rd = gs + j * GP;
  v4f dh[RT];                                              // carry: d h_t through time (beyond what d state adds), this lane's units
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) dh[rt] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int t = T - 2; t >= 0; --t) {
    // ---- gate backward of step t for this lane's (row, 4 units): d gi, d gh, the direct part of d h_{t-1}
    v4f direct[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int row = r0 + 4 * rt + j;
      const int rowc = row < B ? row : B - 1;
      const long long m = (long long)t * B + rowc;
      const v4f ds = *(gcf4)(a.dstate + m * L + u0);
      const v4f gir = *(gcf4)(a.gi + m * L3 + u0), giz = *(gcf4)(a.gi + m * L3 + L + u0), gin = *(gcf4)(a.gi + m * L3 + 2 * L + u0);
      const v4f ghr = *(gcf4)(a.gh + m * L3 + u0), ghz = *(gcf4)(a.gh + m * L3 + L + u0), ghn = *(gcf4)(a.gh + m * L3 + 2 * L + u0);
      const v4f hp = *(gcf4)(a.hprev + m * L + u0);
      v4f dr, dz, dn, dnr;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
#pragma clang fp contract(off)
        const float d = ds[c] + dh[rt][c];
        const float r = gs_sigmoid(gir[c] + ghr[c]);
        const float z = gs_sigmoid(giz[c] + ghz[c]);
        const float n = gs_tanh(gin[c] + r * ghn[c]);
        const float dn_pre = (d * (1.f - z)) * (1.f - n * n);
        const float dz_pre = (d * (hp[c] - n)) * (z * (1.f - z));
        const float dr_pre = (dn_pre * ghn[c]) * (r * (1.f - r));
        dr[c] = dr_pre; dz[c] = dz_pre; dn[c] = dn_pre; dnr[c] = dn_pre * r;
        direct[rt][c] = d * z;
      }
      if (row < B) {
        *(gf4)(a.dgi + m * L3 + u0) = dr; *(gf4)(a.dgi + m * L3 + L + u0) = dz; *(gf4)(a.dgi + m * L3 + 2 * L + u0) = dn;
        *(gf4)(a.dgh + m * L3 + u0) = dr; *(gf4)(a.dgh + m * L3 + L + u0) = dz; *(gf4)(a.dgh + m * L3 + 2 * L + u0) = dnr;
      }
      float *grow = gs + (4 * rt + j) * GP;
      *reinterpret_cast<v4f *>(grow + u0) = dr;
      *reinterpret_cast<v4f *>(grow + L + u0) = dz;
      *reinterpret_cast<v4f *>(grow + 2 * L + u0) = dnr;
    }
    __syncthreads();   // the d gh tile is complete
    v4f acc[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[rt] = v4f{0.f, 0.f, 0.f, 0.f};
    const int nb = Q / 3;
#pragma unroll 1
    for (int bo = 0; bo < nb; bo += GS_DEPTH) {
#pragma unroll
      for (int u = 0; u < GS_DEPTH; ++u) {
        v4f gq[RT][3];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int i = 0; i < 3; ++i) gq[rt][i] = *reinterpret_cast<const v4f *>(grow_rd + 4 * rt * GP + 4 * (3 * (bo + u) + i));
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[rt] = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[u][i][c], gq[rt][i][c], acc[rt], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 3; ++i) wr[u][i] = stream[64 * (qnext + i)];
        qnext = qnext + 3 == Q ? 0 : qnext + 3;
      }
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int c = 0; c < 4; ++c) dh[rt][c] = direct[rt][c] + acc[rt][c];
    __syncthreads();   // every wave is done reading the d gh tile before the next step overwrites it
  }
  // d h_{-1}: the start state's gradient rows (learned start state: summed over the batch by k_gru_dh0)
  if (a.dh_init) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int row = r0 + 4 * rt + j;
      if (row < B) *(gf4)(a.dh_init + (long long)row * L + u0) = dh[rt];
    }
  }
}

// W_hh [3L, L] -> the scans' streams, one float4 per thread:
//   forward  Pf[w][k4][g][l][c] = W_hh[g L + 64 w + l][4 k4 + c]        (slot q = 3 k4 + g of wave w)
//   backward Pb[w][k4][l][c]    = W_hh[4 k4 + c][64 w + l]              (= W_hh^T[64 w + l][4 k4 + c]; k4 < 3L / 4)
__global__ __launch_bounds__(256) void k_gru_pack(const float *__restrict__ W, int L, float *__restrict__ Pf, float *__restrict__ Pb) {
  const int n4 = 3 * L * L / 4;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n4) return;
  const int l = e & 63;
  if (blockIdx.y == 0) {
    const int Q = 3 * (L / 4);
    const int q = (e >> 6) % Q, w = (e >> 6) / Q;
    const int k4 = q / 3, g = q - 3 * k4;
    *reinterpret_cast<v4f *>(Pf + (long long)e * 4) = *reinterpret_cast<const v4f *>(W + (long long)(g * L + 64 * w + l) * L + 4 * k4);
  } else {
    const int Q = 3 * L / 4;
    const int k4 = (e >> 6) % Q, w = (e >> 6) / Q;
    v4f v;
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = W[(long long)(4 * k4 + c) * L + 64 * w + l];
    *reinterpret_cast<v4f *>(Pb + (long long)e * 4) = v;
  }
}

template <typename K>
hipError_t gs_launch(K kern, const GruScanArgs &a, int R, int lds_floats, hipStream_t s) {
  const int lds_bytes = lds_floats * 4;
  static std::mutex mu;
  {
    std::lock_guard<std::mutex> lk(mu);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(kern, dim3((a.B + R - 1) / R), dim3(a.L), lds_bytes, s, a);
  return hipGetLastError();
}

}  // namespace

bool gru_scan_takes(int B, int L) {
  return plan_switches().gru_scan && L % 64 == 0 && L >= 64 && L <= 256 && B >= 1;
}
static int gru_scan_rt() { return 1; }   // rows per workgroup / 4 (2: measured slower, profiles/r04_gru_scan.txt)
hipError_t gru_scan_fwd_launch(const GruScanArgs &a, hipStream_t s) {
  const int waves = a.L / 64, rt = gru_scan_rt();
  const int lds = 2 * 4 * rt * (a.L + 4);
  (void)waves;
  return rt == 2 ? gs_launch(&k_gru_scan_fwd<2>, a, 8, lds, s) : gs_launch(&k_gru_scan_fwd<1>, a, 4, lds, s);
}
hipError_t gru_scan_bwd_launch(const GruScanArgs &a, hipStream_t s) {
  const int waves = a.L / 64, rt = gru_scan_rt();
  const int lds = 4 * rt * (3 * a.L + 4);
  (void)waves;
  return rt == 2 ? gs_launch(&k_gru_scan_bwd<2>, a, 8, lds, s) : gs_launch(&k_gru_scan_bwd<1>, a, 4, lds, s);
}
hipError_t gru_pack_launch(const float *Whh, int L, float *pack_fwd, float *pack_bwd, hipStream_t s) {
  const int n4 = 3 * L * L / 4;
  hipLaunchKernelGGL(k_gru_pack, dim3((n4 + 255) / 256, 2), dim3(256), 0, s, Whh, L, pack_fwd, pack_bwd);
  return hipGetLastError();
}

}  // namespace fdql
