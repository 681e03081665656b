// Weight-stationary persistent row-block fp32 MFMA GEMM for the forward layers of the critic ensemble (wstat.h).
//
// One 256-thread workgroup per CU serves ONE instance for its whole life and walks that instance's 32-row tiles.  Each of
// its 4 waves owns 64 of the 256 output columns and keeps the 64 x 256 weights of those columns in its 256 AccVGPRs (the
// register file of a CU holds a whole 256 x 256 fp32 layer), so the K loop of a tile issues NO weight loads: 8 MFMAs
// (v_mfma_f32_32x32x2_f32, 512 cycles) per ds_read_b128 of activations.  fp32 MFMA and the vector ALU share one issue
// stream on gfx950 (profiles/r02_rowgemm_notes.txt), so what counts is the number of non-MFMA instructions per tile:
//   * the MFMA computes the TRANSPOSED tile (weights as the A operand, activations as the B operand): a lane ends up with
//     4 consecutive output columns of one row per register quad -> one global_store_dwordx4 per quad, the bias is the
//     accumulators' start value, LeakyReLU is 2 VALU instructions per value;
//   * the finished registers are directly the B operand of v_mfma_f32_4x4x1_16b_f32: the skip head's partial sums
//     (GemmProblem::hf_*) ride as 32 two-pass MFMAs per tile instead of ~900 DPP adds;
//   * the next tile's rows go global -> LDS by LDS-DMA (no staging registers, no ds_write), the previous tile's result
//     (second accumulator set, architectural VGPRs: no v_accvgpr_read) is finished and stored between the MFMA steps.
// Measured on the prototype (tools/proto/wstat.hip, M = 192000, K = N = 256): 131-135 TFLOP/s against 116 for the
// software-pipelined row-block prototype with streamed weights, 110 for k_rowgemm and 92 for the 64x64-tile kernel.
//
// The MFMAs are inline asm: hipcc keeps MFMA sources in architectural VGPRs and would park the 256 stationary registers
// in AccVGPRs behind a v_accvgpr_read per use.  The hazard recogniser does not look into asm, so the rules it would
// enforce are kept by construction (ISA 4.5): a VALU / VMEM reader of an MFMA result sits behind an `anchor()` placed
// after at least one further 8-MFMA step (>= 18 wait states), dependent 4x4x1 MFMAs are 2 wait states apart (s_nop 1),
// back-to-back 32x32x2 MFMAs on one accumulator need none.
#include "wstat.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <type_traits>
#include <utility>

namespace fdql {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void *lds_vp;
typedef const __attribute__((address_space(1))) void *glb_vp;
typedef const __attribute__((address_space(1))) v4f *gcf4;
typedef const __attribute__((address_space(1))) float *gcf;
typedef __attribute__((address_space(1))) float *gf;
typedef __attribute__((address_space(1))) v4f *gf4;
typedef __attribute__((address_space(1))) v2f *gf2;

constexpr int NSTEP = WS_KMAIN / 8;   // 32 steps of 8 k (4 k-pairs x both lane halves)

#ifdef WS_STAMPS   // diagnostic build (tools/ws_stamps.py): shader clock at the seams of a workgroup's life, never inside a K loop
__device__ unsigned long long g_ws_stamps[1024 * 8];
#define WS_STAMP(i) st_[i] = __builtin_amdgcn_s_memtime()
#define WS_STAMP_DECL unsigned long long st_[6] = {0, 0, 0, 0, 0, 0}; int st_tiles = 0
#define WS_STAMP_TILE ++st_tiles
#define WS_STAMP_OUT                                                                    \
  if (threadIdx.x == 0 && blockIdx.x < 1024) {                                          \
    for (int i_ = 0; i_ < 6; ++i_) g_ws_stamps[8 * blockIdx.x + i_] = st_[i_];          \
    g_ws_stamps[8 * blockIdx.x + 6] = (unsigned long long)st_tiles;                     \
  }
#else
#define WS_STAMP(i)
#define WS_STAMP_DECL
#define WS_STAMP_TILE
#define WS_STAMP_OUT
#endif

template <int I, int N, typename F>
__device__ __forceinline__ void sfor(F &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    sfor<I + 1, N>(f);
  }
}
__device__ __forceinline__ unsigned ws_lds_addr(const float *p) { return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float *)p; }
template <int OFF>
__device__ __forceinline__ void ws_rd128(v4f &d, unsigned addr) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(d) : "v"(addr), "n"(OFF)); }
template <int OFF>
__device__ __forceinline__ void ws_rd128a(v4f &d, unsigned addr) {   // destination in the AccVGPRs
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(d) : "v"(addr), "n"(OFF));
}
template <int OFF>
__device__ __forceinline__ void ws_rd32(float &d, unsigned addr) { asm volatile("ds_read_b32 %0, %1 offset:%2" : "=&v"(d) : "v"(addr), "n"(OFF)); }
template <int N>
__device__ __forceinline__ void ws_lgkm_wait() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ int ws_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <typename T>
__device__ __forceinline__ T *ws_uni(T *p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (T *)(((unsigned long long)hi << 32) | lo);
}
// One step (8 k) of a 32-row tile: 4 k-pairs x the wave's 2 column tiles in ONE asm statement (between separate asm
// statements the compiler pads a def -> use pair with s_nop).  The weights are the A operand: D[i = column][j = row].
__device__ __forceinline__ void ws_step_a(f32x16 &a0, f32x16 &a1, const v4f &w0, const v4f &w1, const v4f &x) {   // weights in AccVGPRs
  asm volatile(
      "v_mfma_f32_32x32x2_f32 %0, %2, %10, %0\n\tv_mfma_f32_32x32x2_f32 %1, %6, %10, %1\n\t"
      "v_mfma_f32_32x32x2_f32 %0, %3, %11, %0\n\tv_mfma_f32_32x32x2_f32 %1, %7, %11, %1\n\t"
      "v_mfma_f32_32x32x2_f32 %0, %4, %12, %0\n\tv_mfma_f32_32x32x2_f32 %1, %8, %12, %1\n\t"
      "v_mfma_f32_32x32x2_f32 %0, %5, %13, %0\n\tv_mfma_f32_32x32x2_f32 %1, %9, %13, %1"
      : "+v"(a0), "+v"(a1)
      : "a"(w0.x), "a"(w0.y), "a"(w0.z), "a"(w0.w), "a"(w1.x), "a"(w1.y), "a"(w1.z), "a"(w1.w), "v"(x.x), "v"(x.y), "v"(x.z), "v"(x.w));
}
__device__ __forceinline__ void ws_step_a0(f32x16 &a0, f32x16 &a1, const v4f &w0, const v4f &w1, const v4f &x) {   // first step of a tile: C = 0
  asm volatile(
      "v_mfma_f32_32x32x2_f32 %0, %2, %10, 0\n\tv_mfma_f32_32x32x2_f32 %1, %6, %10, 0\n\t"
      "v_mfma_f32_32x32x2_f32 %0, %3, %11, %0\n\tv_mfma_f32_32x32x2_f32 %1, %7, %11, %1\n\t"
      "v_mfma_f32_32x32x2_f32 %0, %4, %12, %0\n\tv_mfma_f32_32x32x2_f32 %1, %8, %12, %1\n\t"
      "v_mfma_f32_32x32x2_f32 %0, %5, %13, %0\n\tv_mfma_f32_32x32x2_f32 %1, %9, %13, %1"
      : "=&v"(a0), "=&v"(a1)
      : "a"(w0.x), "a"(w0.y), "a"(w0.z), "a"(w0.w), "a"(w1.x), "a"(w1.y), "a"(w1.z), "a"(w1.w), "v"(x.x), "v"(x.y), "v"(x.z), "v"(x.w));
}
__device__ __forceinline__ void ws_step_v(f32x16 &a0, f32x16 &a1, const v4f &w0, const v4f &w1, const v4f &x) {   // weights in VGPRs (narrow segments)
  asm volatile(
      "v_mfma_f32_32x32x2_f32 %0, %2, %10, %0\n\tv_mfma_f32_32x32x2_f32 %1, %6, %10, %1\n\t"
      "v_mfma_f32_32x32x2_f32 %0, %3, %11, %0\n\tv_mfma_f32_32x32x2_f32 %1, %7, %11, %1\n\t"
      "v_mfma_f32_32x32x2_f32 %0, %4, %12, %0\n\tv_mfma_f32_32x32x2_f32 %1, %8, %12, %1\n\t"
      "v_mfma_f32_32x32x2_f32 %0, %5, %13, %0\n\tv_mfma_f32_32x32x2_f32 %1, %9, %13, %1"
      : "+v"(a0), "+v"(a1)
      : "v"(w0.x), "v"(w0.y), "v"(w0.z), "v"(w0.w), "v"(w1.x), "v"(w1.y), "v"(w1.z), "v"(w1.w), "v"(x.x), "v"(x.y), "v"(x.z), "v"(x.w));
}
// Skip-head rider of one register quad: h[q][lane] += sum_c w[c] (lane 4 b + i: head weight row i over the column of
// register c in lane half b / 8) * x[c] (lane 4 b + j: row 4 (b % 8) + j).  Dependent 4x4x1 MFMAs: 2 wait states apart.
__device__ __forceinline__ void ws_rider(v4f &h, const v4f &w, const v4f &x) {
  asm volatile(
      "v_mfma_f32_4x4x1_16b_f32 %0, %1, %5, %0\n\ts_nop 1\n\t"
      "v_mfma_f32_4x4x1_16b_f32 %0, %2, %6, %0\n\ts_nop 1\n\t"
      "v_mfma_f32_4x4x1_16b_f32 %0, %3, %7, %0\n\ts_nop 1\n\t"
      "v_mfma_f32_4x4x1_16b_f32 %0, %4, %8, %0"
      : "+v"(h)
      : "v"(w.x), "v"(w.y), "v"(w.z), "v"(w.w), "v"(x.x), "v"(x.y), "v"(x.z), "v"(x.w));
}
// A reader of MFMA results written by the asm statements above must not be scheduled ahead of the steps that separate
// it from them: volatile asm statements keep their order, and what consumes the "output" of this one follows it.
__device__ __forceinline__ void ws_anchor(f32x16 &a, f32x16 &b) { asm volatile("" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void ws_anchor(v4f &a) { asm volatile("" : "+v"(a)); }
__device__ __forceinline__ float ws_lrelu(float v) {   // max(v, 0.01 v); fmaxf would add a canonicalising v_max
  float t = 0.01f * v, o;
  asm("v_max_f32 %0, %1, %2" : "=v"(o) : "v"(v), "v"(t));
  return o;
}
// gate masks: m = (m << 1) | (x > 0) - the first value packed ends up in bit 31 after 32 of them.  Integer form, no VCC
// (a v_cmp / v_addc pair stalls on its carry every time: + 9 us per forward launch at config 2): x > 0  <=>  its bit pattern is
// a positive integer  <=>  0 - bits is negative (+0 -> 0: not positive, as LeakyReLU'(0) = 0.01 wants; the one pattern that
// goes wrong is -0, which max(v, 0.01 v) only returns for |v| below 1.4e-43).  Plain C++ on purpose (v_sub_u32 + v_alignbit_b32
// from the compiler, the temporary in the value's own register when it can): as an asm statement with a temporary of its own
// the pair got a register that an LDS read issued by an earlier asm statement - a prefetch nobody consumes on that path, so
// dead for the allocator the moment it is issued - was still going to write (wrong forward outputs, found with tools/nan_probe.py).
__device__ __forceinline__ void ws_mask_push(unsigned &m, float x) {
  const unsigned t = 0u - __builtin_bit_cast(unsigned, x);
  m = (m << 1) | (t >> 31);
}
// ... and back out: returns (bit 31 of m) ? v : 0.01 v, m <<= 1
__device__ __forceinline__ float ws_mask_gate(unsigned &m, float v) {
  const float t = 0.01f * v;
  float y;
  asm volatile("v_add_co_u32 %1, vcc, %1, %1\n\tv_cndmask_b32 %0, %2, %3, vcc" : "=v"(y), "+v"(m) : "v"(t), "v"(v) : "vcc");
  return y;
}
__device__ __forceinline__ float ws_sum_halves(float x) {   // x[l] + x[l ^ 32] in every lane
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}

// NSL: narrow 8-k steps inside a tile's K loop (a narrow segment of K <= 32 columns takes ceil(K / 8) of them: the action
// columns of critic layer 0 - 6 at config 2, 17 at config 4).  NST: steps of the LAST narrow segment of a two-output launch,
// applied to the previous tile between its two outputs (0: one output).  HFQ: head-fusion outputs per row (0 = off).
// GM: the launch writes gate masks (GemmProblem::gm_out / gm_out2; instances without one only skip the store).
template <int NSL, int NST, int HFQ, bool GM = false>
__global__ __launch_bounds__(256, 1) void k_wstat(const WsArgs a) {
  static_assert(HFQ == 0 || HFQ == 2, "head-fusion riders: 2 outputs per row");
  constexpr bool DUAL = NST > 0;
  constexpr int NS = NSL + NST;
  extern __shared__ __attribute__((aligned(16))) float lds[];   // two images [32][P], then the per-wave constants
  WS_STAMP_DECL;
  WS_STAMP(0);
  constexpr int P = WS_KMAIN + 8 * NS + 4, IMG = WS_BM * P;   // (P / 4) odd: conflict-free ds_read_b128
  constexpr int LD = WS_N;                                    // row pitch of the outputs (compile-time)
  const int tid = threadIdx.x, lane = tid & 63, wave = ws_uni(tid >> 6);
  const int li = lane & 31, lh = lane >> 5, n0 = wave * 64;

  // ---- this workgroup's instance and its share of the instance's tiles: j0, j0 + stride, ...
  int inst = 0;
  for (int i = 1; i < a.ninst; ++i)
    if ((int)blockIdx.x >= a.wg_first[i]) inst = i;
  inst = ws_uni(inst);
  const int j0 = (int)blockIdx.x - a.wg_first[inst], stride = a.wg_first[inst + 1] - a.wg_first[inst];
  const int nblk = a.blocks_per_inst;   // (M = 32 nblk)
  if (j0 >= nblk) return;
  const WsInst &I = a.inst[inst];   // kernel-argument segment: scalar loads, each field read once
  const float *A0 = ws_uni(I.A[0]);
  // (GM: the output / narrow-input / head-sum pointers are not held in scalar registers over the tile loop but read from the
  // kernel arguments where a tile needs them - ka_ptr below; the mask path's few extra scalars would not fit otherwise)
  // (only the two-output instantiations are short of scalar registers: the others keep every pointer in registers, masks or not)
  constexpr bool KA = GM && DUAL;
  float *const C = KA ? nullptr : ws_uni(I.C), *const C2 = KA ? nullptr : ws_uni(I.C2);
  // DUAL launches may mix two-output instances with plain ones (critic layer 0: online + frozen pass / target pass): a plain
  // instance has no second output and no last narrow segment - those slots are staged from the first segment's memory and
  // never used
  const bool has2 = DUAL && ws_uni(I.C2) != nullptr;
  // Pointers that are used once per tile are read from the kernel-argument segment where they are used, at an offset the
  // optimiser cannot see through (so that the load is not hoisted out of the tile loop): the scalar registers are all taken.
  auto ka_ptr = [&](int field_off) __attribute__((always_inline)) {
    int off = (int)offsetof(WsArgs, inst) + field_off + inst * (int)sizeof(WsInst);
    asm volatile("" : "+s"(off));
    const char __attribute__((address_space(4))) *ka = (const char __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr();
    void *p = *(void *const __attribute__((address_space(4))) *)(ka + off);
    // the scalar load counts in lgkmcnt and returns out of order with LDS operations: it must not be outstanding when the tile
    // loop's counted waits (ws_lgkm_wait) run - both statements are volatile, the load sits between them
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(p));
    return p;
  };
  float *const hf_out_r = KA ? nullptr : ws_uni(I.hf_out), *const hf_out2_r = KA ? nullptr : ws_uni(I.hf_out2);
  unsigned *const gm_out_r = (GM && !KA) ? ws_uni(I.gm_out) : nullptr, *const gm_out2_r = (GM && !KA) ? ws_uni(I.gm_out2) : nullptr;
  auto hf_ptr = [&](int second) __attribute__((always_inline)) {
    if constexpr (KA) return (float *)ka_ptr((int)offsetof(WsInst, hf_out) + second * (int)sizeof(float *));
    else return second ? hf_out2_r : hf_out_r;
  };
  auto c_ptr = [&](int second) __attribute__((always_inline)) {
    if constexpr (KA) return (float *)ka_ptr((int)offsetof(WsInst, C) + second * (int)sizeof(float *));
    else return second ? C2 : C;
  };
  constexpr bool masks = GM;
  // presum: the tile whose sums the last block left in hfx (-1: none yet); -2: the launch writes every plane (one scalar
  // register carries both: this kernel has none to spare)
  int pprv = (HFQ > 0 && a.hf_presum != 0) ? -1 : -2;
  float *const hfx = lds + (2 * IMG + 4 * 2 * 32 + 4 * 8 * 32 + NS * 2048);   // behind the per-wave constants: [2][2][4][32 * HFQ]
  const int lda0 = a.lda[0];

  // ---- stationary main weights: wb[tn][s] = W0[n0 + 32 tn + li][32 (s / 4) + 16 lh + 4 (s % 4) .. + 3]
  // Loaded as whole ROWS (a wave instruction = one 1 KiB row: 8 cache lines, not the 64 lines a lane-per-row load touches:
  // the lane-per-row form took 21-24 k cycles = 1.4 tile times per workgroup, tools/ws_stamps.py) and redistributed through
  // the wave's own slice of LDS - the image / constant areas are not in use yet: 32 rows x (256 + 4) floats per wave and pass,
  // conflict-free ds_read_b128 straight into the AccVGPRs.
  v4f wb[2][NSTEP];
  {
    const float *W0 = ws_uni(I.W[0]);
    const int ldw0 = a.ldw[0];
    constexpr int WP = WS_KMAIN + 4;
    float *const wst = lds + wave * (32 * WP);
    const unsigned wst_rd = ws_lds_addr(wst) + (unsigned)(li * WP + 16 * lh) * 4u;
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      v4f row[32];
#pragma unroll
      for (int r = 0; r < 32; ++r) row[r] = ((gcf4)(ws_uni(W0 + (long long)(n0 + 32 * tn + r) * ldw0)))[(unsigned)lane];
#pragma unroll
      for (int r = 0; r < 32; ++r) *reinterpret_cast<v4f *>(wst + r * WP + lane * 4) = row[r];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (this wave's own writes; nobody else touches its slice)
      sfor<0, NSTEP>([&](auto sc) __attribute__((always_inline)) {
        constexpr int st = decltype(sc)::value;
        ws_rd128a<((st >> 2) * 32 + (st & 3) * 4) * 4>(wb[tn][st], wst_rd);
      });
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // read before the next pass overwrites the slice
    }
    __syncthreads();   // every wave is done with its slice before images and constants land there
  }
  // Per-wave constants in LDS (behind the images), read shortly before their use:
  //   bias of the columns 8 q + 4 lh + c of each column tile                                        [wave][lh][tn][q][c]
  //   head-fusion rider weights: lane 4 b + i holds row i (< HFQ) of the head over those columns    [wave][i][lh][tn][q][c]
  //   narrow-step weights: slot j covers k = k0_j + 4 lh + c of its segment (zero beyond its K)     [wave][slot][tn][lane][c]
  float *const cbias = lds + 2 * IMG, *const cwh = cbias + 4 * 2 * 32, *const cnw = cwh + 4 * 8 * 32;
  // slot j of the narrow steps: segment 0 (k = 8 j ..) inside the K loop, segment 1 (k = 8 (j - NSL) ..) in the tail
  const float *Aseg_r[2] = {KA ? nullptr : ws_uni(I.A[1]), KA ? nullptr : ws_uni((DUAL && has2) ? I.A[2] : I.A[1])};
  auto aseg = [&](int sg) __attribute__((always_inline)) {
    if constexpr (KA) return (const float *)ka_ptr((int)offsetof(WsInst, A) + (int)sizeof(float *) * ((sg == 1 && DUAL && has2) ? 2 : 1));
    else return Aseg_r[sg];
  };
  const int ldaseg_r[2] = {KA ? 0 : a.lda[1], KA ? 0 : a.lda[2]};
  auto ldaseg = [&](int i) __attribute__((always_inline)) {
    if constexpr (KA) {
      int off = (int)offsetof(WsArgs, lda) + 4 * (1 + i);
      asm volatile("" : "+s"(off));
      int v = *(const int __attribute__((address_space(4))) *)((const char __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr() + off);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(v));   // (as ka_ptr: not outstanding at a counted wait)
      return v;
    } else {
      return ldaseg_r[i];
    }
  };
  bool m_ok[NS > 0 ? NS : 1];
  const int m_r = tid >> 3, m_c = tid & 7;   // narrow staging: thread -> (row tid / 8, column tid % 8) of every [32, 8] slot
  {
    const float *bias = ws_uni(I.bias);
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<v4f *>(cbias + (wave * 2 + lh) * 32 + (tn * 4 + q) * 4) = *(gcf4)(bias + n0 + 32 * tn + 8 * q + 4 * lh);
    if constexpr (HFQ > 0) {
      const float *hw = ws_uni(I.hf_w);
      const int i = lane & 3, ic = i < HFQ ? i : 0;
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const v4f x = *(gcf4)(hw + (long long)ic * a.hf_ldw + n0 + 32 * tn + 8 * q + 4 * lh);
          *reinterpret_cast<v4f *>(cwh + ((wave * 4 + i) * 2 + lh) * 32 + (tn * 4 + q) * 4) = i < HFQ ? x : v4f{0.f, 0.f, 0.f, 0.f};
        }
    }
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      // (a plain instance of a two-output launch: its tail slots read the first segment's memory and are never used)
      const int sgl = j < NSL ? 0 : 1, k0 = 8 * (j < NSL ? j : j - NSL);
      const int sg = (DUAL && !has2) ? 0 : sgl;
      const int Ks = a.kminor[sgl], ldw = a.ldw[1 + sg];
      const float *Ws = ws_uni(I.W[1 + sg]);
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
        v4f w;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int k = k0 + 4 * lh + c, kc = k < Ks ? k : Ks - 1;   // clamped, unconditional loads
          const float x = ((gcf)Ws)[(long long)(n0 + 32 * tn + li) * ldw + kc];
          w[c] = k < Ks ? x : 0.f;
        }
        *reinterpret_cast<v4f *>(cnw + (((wave * (NS > 0 ? NS : 1) + j) * 2 + tn) * 64 + lane) * 4) = w;
      }
      m_ok[j] = k0 + m_c < Ks;
    }
    // (lanes that share a slot write the same values; read back by this wave only, after its s_waitcnt below)
  }
  const unsigned cb_addr = ws_lds_addr(cbias) + (unsigned)((wave * 2 + lh) * 32) * 4u;
  const unsigned cw_addr = ws_lds_addr(cwh) + (unsigned)(((wave * 4 + (lane & 3)) * 2 + lh) * 32) * 4u;
  const unsigned nw_addr = ws_lds_addr(cnw) + (unsigned)((wave * (NS > 0 ? NS : 1) * 2 * 64 + lane) * 4) * 4u;   // + (2 slot + tn) KiB

  // ---- addresses
  const unsigned abase = ws_lds_addr(lds) + (unsigned)(li * P + 16 * lh) * 4u;                 // main fragments: k = 16 lh + ...
  const unsigned nbase = ws_lds_addr(lds) + (unsigned)(li * P + WS_KMAIN + 4 * lh) * 4u;       // narrow fragments: k = 4 lh + c
  const unsigned vo_c = (unsigned)(li * LD + 4 * lh);   // lane offset of a row's quad in an output tile
  float *const m_dst = lds + m_r * P + WS_KMAIN + m_c;

  auto dma_row = [&](const float *src_tile, int img, int r) __attribute__((always_inline)) {   // one row: 1 KiB global -> LDS
    __builtin_amdgcn_global_load_lds((glb_vp)(src_tile + (long long)r * lda0 + lane * 4), (lds_vp)(lds + img * IMG + r * P), 16, 0, 0);
  };

  f32x16 acc[2][2];   // [set][tn]: the tile being accumulated and the previous one (being finished)
  v4f hacc[2], hacc2[2];   // rider sums per column tile (second output of a dual launch: hacc2)
  float stm[NS > 0 ? NS : 1];

  // constants of register quad kq = 4 tn + q (bias, rider weights): requested one quad ahead of their use
  v4f cb[2], cw[2];
  auto read_consts = [&](auto kqc) __attribute__((always_inline)) {
    constexpr int kq = decltype(kqc)::value & 7;
    ws_rd128<kq * 16>(cb[kq & 1], cb_addr);
    if constexpr (HFQ > 0) ws_rd128<kq * 16>(cw[kq & 1], cw_addr);
  };
  // operands of narrow step j (fragment from image IMX, the slot's weights): requested one step ahead of their use
  v4f nfr[2], nw0[2], nw1[2];
  auto read_narrow = [&](auto jc, auto imc) __attribute__((always_inline)) {
    constexpr int j = decltype(jc)::value, IMX = decltype(imc)::value;
    ws_rd128<IMX * IMG * 4 + 8 * j * 4>(nfr[j & 1], nbase);
    ws_rd128<(2 * j) * 1024>(nw0[j & 1], nw_addr);
    ws_rd128<(2 * j + 1) * 1024>(nw1[j & 1], nw_addr);
  };
  // finish + store one register quad of a finished set; the rider takes the finished values as its B operand
  unsigned gmk = 0;   // gate mask of the output being finished: sign bits in the order the quads are finished
  auto quad = [&](f32x16 (&pv)[2], float *Cout, v4f (&hs)[2], int pblk, int kq) __attribute__((always_inline)) {
    const int tn = (kq >> 2) & 1, q = kq & 3;
    v4f x;
#pragma unroll
    for (int c = 0; c < 4; ++c) x[c] = ws_lrelu(pv[tn][4 * q + c] + cb[kq & 1][c]);
    gf base = ws_uni((gf)Cout + (long long)pblk * WS_BM * LD + n0 + 32 * tn + 8 * q);
    *(gf4)(&base[vo_c]) = x;
    if constexpr (masks) {   // sign(x) = sign of the pre-activation
#pragma unroll
      for (int c = 0; c < 4; ++c) ws_mask_push(gmk, x[c]);
    }
    if constexpr (HFQ > 0) {
      if (q == 0) hs[tn] = v4f{0.f, 0.f, 0.f, 0.f};
      ws_rider(hs[tn], cw[kq & 1], x);
    }
  };
  // a finished output's gate mask: one dword per lane, [tile][wave][lane]
  // (the pointer is read from the kernel-argument segment where it is used, at an offset the optimiser cannot see through, so
  // that the load is not hoisted out of the tile loop: this kernel has no scalar registers to keep two more pointers in)
  auto mask_store = [&](int second, int pblk) __attribute__((always_inline)) {
    if constexpr (masks) {
      unsigned *mout;
      if constexpr (KA) mout = (unsigned *)ka_ptr((int)offsetof(WsInst, gm_out) + second * (int)sizeof(unsigned *));
      else mout = second ? gm_out2_r : gm_out_r;
      if (mout) ((__attribute__((address_space(1))) unsigned *)ws_uni(mout + ((long long)pblk * 4 + wave) * 64))[(unsigned)lane] = gmk;
    }
  };
  // the head partial sums of a finished tile: plane = wave * 2 + tn (its 32 columns), both lane halves summed.
  // WsArgs::hf_presum (round 4): the eight planes of a tile are summed HERE - each wave leaves its two column tiles' sum in LDS
  // (hfx[parity of the block][output][wave][32 rows x HFQ]), wave 0 adds the four waves in order one block later (behind that
  // block's barrier) and writes ONE plane: the plane-sum launch in front of the head's finish and 7/8 of its input go away.
  auto hf_store = [&](v4f (&hs)[2], int pblk, int par, int o) __attribute__((always_inline)) {
    if constexpr (HFQ > 0) {
      if (pprv != -2) {   // (uniform)
        ws_anchor(hs[0]); ws_anchor(hs[1]);
        const v2f y = {ws_sum_halves(hs[0][0]) + ws_sum_halves(hs[1][0]), ws_sum_halves(hs[0][1]) + ws_sum_halves(hs[1][1])};
        *reinterpret_cast<v2f *>(hfx + (((par * 2 + o) * 4 + wave) * 32 + li) * HFQ) = y;   // (both lane halves write the same sums)
        return;
      }
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
        ws_anchor(hs[tn]);
        const v2f y = {ws_sum_halves(hs[tn][0]), ws_sum_halves(hs[tn][1])};
        gf base = ws_uni((gf)hf_ptr(o) + ((long long)(wave * 2 + tn) * nblk + pblk) * (WS_BM * HFQ));
        *(gf2)(&base[(unsigned)(li * HFQ)]) = y;   // (both lane halves hold and write the same sums)
      }
    }
  };
  // presum: the sums a block left in hfx[par] -> one plane in memory, by wave 0 (block pp's 32 rows x HFQ = 64 floats, one per lane)
  auto hf_emit = [&](int par, int pp) __attribute__((always_inline)) {
    if constexpr (HFQ > 0) {
      if (wave == 0 && pp >= 0) {   // (uniform; pp < 0: nothing pending, or not a presum launch)
        const float *src = hfx + (par * 2) * 4 * 32 * HFQ + lane;
        const float t = (src[0] + src[32 * HFQ]) + (src[2 * 32 * HFQ] + src[3 * 32 * HFQ]);
        ((gf)ws_uni(hf_ptr(0) + (long long)pp * WS_BM * HFQ))[(unsigned)lane] = t;
        if (DUAL && has2) {
          const float *src2 = src + 4 * 32 * HFQ;
          const float t2 = (src2[0] + src2[32 * HFQ]) + (src2[2 * 32 * HFQ] + src2[3 * 32 * HFQ]);
          ((gf)ws_uni(hf_ptr(1) + (long long)pp * WS_BM * HFQ))[(unsigned)lane] = t2;
        }
      }
    }
  };
  // the NST tail steps of a two-output instance: (previous tile) += last narrow segment, fragments from ITS image IMX
  // (intact until this tile's prefetch starts at step 16); step 0's operands were requested a main step earlier
  auto tail_steps = [&](f32x16 (&pv)[2], auto imc) __attribute__((always_inline)) {
    sfor<0, NST>([&](auto jc) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value;
      if constexpr (j + 1 < NST) read_narrow(std::integral_constant<int, NSL + j + 1>{}, imc);
      if constexpr (j > 0) { if constexpr (j + 1 < NST) ws_lgkm_wait<3>(); else ws_lgkm_wait<0>(); }
      asm volatile("" : "+v"(nfr[(NSL + j) & 1]), "+v"(nw0[(NSL + j) & 1]), "+v"(nw1[(NSL + j) & 1]));
      ws_step_v(pv[0], pv[1], nw0[(NSL + j) & 1], nw1[(NSL + j) & 1], nfr[(NSL + j) & 1]);
    });
  };

  // One tile: K loop into `ac` from image IM; the previous tile `pv` (block pblk) is finished and stored, the rows of the
  // next tile (block nxt) are fetched into the other image.
  auto block = [&](auto has_prev, auto imgc, f32x16 (&ac)[2], f32x16 (&pv)[2], int nxt, int pblk, int ppblk) __attribute__((always_inline)) {
    constexpr bool HP = decltype(has_prev)::value;
    constexpr int IM = decltype(imgc)::value;
    constexpr int IOFF = IM * IMG * 4;   // < 64 KiB: fits the ds offset field
    // every wave is done with the other image and this image has landed (each wave waited for its own DMA rows and
    // narrow writes before it arrived here)
    asm volatile("s_barrier" ::: "memory");
    const float *nsrc = ws_uni(A0 + (long long)nxt * WS_BM * lda0);
    v4f af[2];
    float em[8];   // presum: the four waves' sums of the tile before the previous one (first / second output), wave 0
    float *Cb = nullptr, *C2b = nullptr;
    if constexpr (HP) Cb = c_ptr(0);
    ws_rd128<IOFF>(af[0], abase);
    if constexpr (HP) read_consts(std::integral_constant<int, 0>{});
    sfor<0, NSTEP>([&](auto sc) __attribute__((always_inline)) {
      constexpr int s = decltype(sc)::value;
      // fragment of the next step requested; everything older (this step's fragment, the constants / narrow operands
      // requested during the previous step) has arrived: LDS operations complete in order
      if constexpr (s + 1 < NSTEP) {
        constexpr int s1 = s + 1;
        ws_rd128<IOFF + (s1 >> 2) * 128 + (s1 & 3) * 16>(af[s1 & 1], abase);
        ws_lgkm_wait<1>();
      } else {
        ws_lgkm_wait<0>();
      }
      asm volatile("" : "+v"(af[s & 1]));
      if constexpr (s == 0) ws_step_a0(ac[0], ac[1], wb[0][s], wb[1][s], af[s & 1]);
      else ws_step_a(ac[0], ac[1], wb[0][s], wb[1][s], af[s & 1]);
      // ---- side work of step s
      if constexpr (HP) {
        // quad kq is finished at step s: first output kq = s (0..7); second output of a dual launch kq = s - 1 (8..15)
        constexpr int kq = s < 8 ? s : s - 1;
        if constexpr (s < 8) {
          asm volatile("" : "+v"(cb[kq & 1]));
          if constexpr (HFQ > 0) asm volatile("" : "+v"(cw[kq & 1]));
          if constexpr (kq + 1 < 8) read_consts(std::integral_constant<int, kq + 1>{});
        }
        if constexpr (s == 0) ws_anchor(pv[0], pv[1]);
        if constexpr (s < 8) quad(pv, Cb, hacc, pblk, kq);
        if constexpr (s == 8) mask_store(0, pblk);
        // what the block before this one left in hfx[IM ^ 1]: wave 0 requests the four waves' sums at step 3 (uncounted asm
        // reads: in order behind them, step 4's counted wait covers their arrival) and adds / stores them at step 4
        if constexpr (HFQ > 0 && (s == 3 || s == 4)) {
          if (wave == 0 && ppblk >= 0) {   // (uniform)
            constexpr int EB = ((IM ^ 1) * 2) * 4 * 32 * HFQ * 4, WSTEP = 32 * HFQ * 4;   // bytes
            if constexpr (s == 3) {
              const unsigned ea = ws_lds_addr(hfx) + (unsigned)lane * 4u;
              ws_rd32<EB>(em[0], ea); ws_rd32<EB + WSTEP>(em[1], ea); ws_rd32<EB + 2 * WSTEP>(em[2], ea); ws_rd32<EB + 3 * WSTEP>(em[3], ea);
              if (DUAL && has2) {
                ws_rd32<EB + 4 * WSTEP>(em[4], ea); ws_rd32<EB + 5 * WSTEP>(em[5], ea); ws_rd32<EB + 6 * WSTEP>(em[6], ea); ws_rd32<EB + 7 * WSTEP>(em[7], ea);
              }
            } else {
              asm volatile("" : "+v"(em[0]), "+v"(em[1]), "+v"(em[2]), "+v"(em[3]));
              ((gf)ws_uni(hf_ptr(0) + (long long)ppblk * WS_BM * HFQ))[(unsigned)lane] = (em[0] + em[1]) + (em[2] + em[3]);
              if (DUAL && has2) {
                asm volatile("" : "+v"(em[4]), "+v"(em[5]), "+v"(em[6]), "+v"(em[7]));
                ((gf)ws_uni(hf_ptr(1) + (long long)ppblk * WS_BM * HFQ))[(unsigned)lane] = (em[4] + em[5]) + (em[6] + em[7]);
              }
            }
          }
        }
        if constexpr (!DUAL) {
          if constexpr (s == 9) hf_store(hacc, pblk, IM, 0);
        } else {
          if constexpr (s == 10) hf_store(hacc, pblk, IM, 0);
          if constexpr (s >= 7 && s <= 18) {
            if (has2) {   // (uniform: the instance's second output)
              if constexpr (s == 7) read_narrow(std::integral_constant<int, NSL>{}, std::integral_constant<int, IM ^ 1>{});
              if constexpr (s == 8) {
                C2b = c_ptr(1);
                tail_steps(pv, std::integral_constant<int, IM ^ 1>{});   // the previous tile becomes its second output
                read_consts(std::integral_constant<int, 8>{});
              }
              if constexpr (s >= 9 && s < 17) {
                asm volatile("" : "+v"(cb[kq & 1]));
                if constexpr (HFQ > 0) asm volatile("" : "+v"(cw[kq & 1]));
                if constexpr (kq + 1 < 16) read_consts(std::integral_constant<int, kq + 1>{});
                if constexpr (s == 9) ws_anchor(pv[0], pv[1]);
                quad(pv, C2b, hacc2, pblk, kq);
              }
              if constexpr (s == 17) mask_store(1, pblk);
              if constexpr (s == 18) hf_store(hacc2, pblk, IM, 1);
            }
          }
        }
      }
      if constexpr (s == 12) {
#pragma unroll
        for (int u = 0; u < NS; ++u) {
          const int sg = u < NSL ? 0 : 1, k0 = 8 * (u < NSL ? u : u - NSL), ld = ldaseg((DUAL && !has2) ? 0 : sg);
          stm[u] = ((gcf)(aseg(sg) + (long long)nxt * WS_BM * ld))[m_r * ld + (m_ok[u] ? k0 + m_c : 0)];
        }
      }
      if constexpr (s >= 16 && s < 16 + WS_BM / 4) dma_row(nsrc, IM ^ 1, wave + 4 * (s - 16));
      if constexpr (s == 27) {
#pragma unroll
        for (int u = 0; u < NS; ++u) m_dst[(IM ^ 1) * IMG + 8 * u] = m_ok[u] ? stm[u] : 0.f;
      }
      if constexpr (NSL > 0 && s == NSTEP - 1) read_narrow(std::integral_constant<int, 0>{}, imgc);   // the K loop's first narrow step
      asm volatile("" ::: "memory");   // the memory operations of a step stay in their step
    });
    // ---- the narrow steps of this tile
    sfor<0, NSL>([&](auto jc) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value;
      if constexpr (j + 1 < NSL) { read_narrow(std::integral_constant<int, j + 1>{}, imgc); ws_lgkm_wait<3>(); }
      else ws_lgkm_wait<0>();
      asm volatile("" : "+v"(nfr[j & 1]), "+v"(nw0[j & 1]), "+v"(nw1[j & 1]));
      ws_step_v(ac[0], ac[1], nw0[j & 1], nw1[j & 1], nfr[j & 1]);
    });
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // this wave's part of the next image has landed
  };
  using T = std::true_type;
  using F = std::false_type;
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  // ---- first image (once per workgroup)
  int blk = j0;
  {
    const float *src = A0 + (long long)blk * WS_BM * lda0;
#pragma unroll
    for (int u = 0; u < WS_BM / 4; ++u) dma_row(src, 0, wave + 4 * u);
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      const int sg = u < NSL ? 0 : 1, k0 = 8 * (u < NSL ? u : u - NSL), ld = ldaseg((DUAL && !has2) ? 0 : sg);
      const float x = ((gcf)(aseg(sg) + (long long)blk * WS_BM * ld))[m_r * ld + (m_ok[u] ? k0 + m_c : 0)];
      m_dst[8 * u] = m_ok[u] ? x : 0.f;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  }
  WS_STAMP(1);
  int nxt = blk + stride < nblk ? blk + stride : blk;   // a workgroup's last tile prefetches itself again (nobody reads it)
  block(F(), I0(), acc[0], acc[1], nxt, 0, -1);
  WS_STAMP(2);
  WS_STAMP_TILE;
  int prv = blk, set = 1;
  blk += stride;
#pragma unroll 1
  while (blk < nblk) {
    nxt = blk + stride < nblk ? blk + stride : blk;
    block(T(), I1(), acc[1], acc[0], nxt, prv, pprv);
    WS_STAMP_TILE;
    pprv = pprv == -2 ? -2 : prv; prv = blk; blk += stride; set = 0;
    if (blk >= nblk) break;
    nxt = blk + stride < nblk ? blk + stride : blk;
    block(T(), I0(), acc[0], acc[1], nxt, prv, pprv);
    WS_STAMP_TILE;
    pprv = pprv == -2 ? -2 : prv; prv = blk; blk += stride; set = 1;
  }
  WS_STAMP(3);
  // ---- the last tile's result, not overlapped.  Its image is still in place: image (set ^ 1) - the last block ran on it
  auto flush = [&](f32x16 (&pv)[2], auto imc) __attribute__((always_inline)) {
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");   // MFMA results -> VALU readers
    ws_anchor(pv[0], pv[1]);
    sfor<0, 8>([&](auto kc) __attribute__((always_inline)) {
      read_consts(kc);
      ws_lgkm_wait<0>();
      asm volatile("" : "+v"(cb[decltype(kc)::value & 1]), "+v"(cw[decltype(kc)::value & 1]));
      quad(pv, c_ptr(0), hacc, prv, decltype(kc)::value);
    });
    mask_store(0, prv);
    if (DUAL && has2) {
      if constexpr (DUAL) {
        read_narrow(std::integral_constant<int, NSL>{}, imc);
        ws_lgkm_wait<0>();
        tail_steps(pv, imc);
      }
      asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
      ws_anchor(pv[0], pv[1]);
      sfor<0, 8>([&](auto kc) __attribute__((always_inline)) {
        read_consts(kc);
        ws_lgkm_wait<0>();
        asm volatile("" : "+v"(cb[decltype(kc)::value & 1]), "+v"(cw[decltype(kc)::value & 1]));
        quad(pv, c_ptr(1), hacc2, prv, decltype(kc)::value);
      });
      mask_store(1, prv);
    }
    asm volatile("s_nop 7" ::: "memory");   // rider results -> VALU readers
    constexpr int PL = decltype(imc)::value;   // the last block ran on image PL and left tile pprv's sums in hfx[PL]
    if constexpr (HFQ > 0) {
      if (pprv != -2) __syncthreads();   // wave 0 has taken what it had to take from hfx[PL ^ 1] (the last block's step 3)
    }
    hf_store(hacc, prv, PL ^ 1, 0);
    if (DUAL && has2) hf_store(hacc2, prv, PL ^ 1, 1);
    if constexpr (HFQ > 0) {
      if (pprv != -2) {
        __syncthreads();
        hf_emit(PL, pprv);
        hf_emit(PL ^ 1, prv);
      }
    }
  };
  if (set == 1) flush(acc[0], I0());
  else flush(acc[1], I1());
  WS_STAMP(4);
  WS_STAMP_OUT;
}


// ======================================================================================================================
// dgrad form:  C[M, 256] = LeakyReLU'(ref) * ( A0[M, 256] W0 + dY[M, K1] W1 ),  weights K-strided (element (k, n) at W[k*ldw + n]),
// column sums of C per workgroup.  FUSE: A0 is not read but formed while the next tile is staged - the head dgrad of the
// layer above (GemmProblem::fz_*):  A0[m][k] = LeakyReLU'(fz_h[m][k]) * (dY[m][0] fz_w[0][k] + dY[m][1] fz_w[1][k]), written to
// fz_out for the weight gradients, its column sums per workgroup to fz_colsum.
// Same transposed tile as the forward form (a lane holds 4 consecutive columns of ONE row per register quad), so the sums
// over ROWS that the bias gradients need run over lanes: each lane keeps its own running sums over all tiles of the
// workgroup (one v_add per value, as an in-lane sum would cost) and the 32 lanes are added ONCE, when the workgroup ends;
// colsum / fz_colsum therefore hold one partial row per WORKGROUP of the instance (WsArgs::wg_first), not per 64 rows.
#ifndef WS_EXP
#define WS_EXP 0   // timing experiments only (results are wrong with any bit set)
#endif
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float ws_dpp(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, ROW_MASK, 0xf, true));
}
__device__ __forceinline__ float ws_sum32(float x) {   // sum over the 32 lanes of a lane half: valid in lanes 31 and 63
  x += ws_dpp<0x111, 0xf>(x);   // row_shr:1
  x += ws_dpp<0x112, 0xf>(x);   // row_shr:2
  x += ws_dpp<0x114, 0xf>(x);   // row_shr:4
  x += ws_dpp<0x118, 0xf>(x);   // row_shr:8
  x += ws_dpp<0x142, 0xa>(x);   // row_bcast:15 into rows 1 and 3
  return x;
}

// PLAIN: no gate and no column sums (C = A0 W0 + dY W1: one network's share of an input gradient that several networks add up
// to, e.g. d state of the critics), K-strided weights of any row pitch.
// NS: 8-k steps of the narrow segment dY (2 columns at config 2: one step; 25 at config 4: four).
// MASK: the gates come from the forward launches' sign masks (GemmProblem::gm_*: a dword per lane and tile in this very register
// layout, consumed from bit 31 down in the order the quads are finished; the fused loader takes its four columns' nibble of the
// row's dword) instead of from the activations themselves: 32 bytes per row and gate instead of 1 KiB.
template <bool FUSE, bool PLAIN = false, int NS = 1, bool MASK = false>
__global__ __launch_bounds__(256, 1) void k_wstat_grad(const WsArgs a) {
  static_assert(!(MASK && PLAIN), "the plain form has no gate");
  static_assert(!(FUSE && PLAIN), "the fused head dgrad belongs to a gated layer");
  static_assert(!FUSE || NS == 1, "the fused head dgrad reads dY as two columns");
  extern __shared__ __attribute__((aligned(16))) float lds[];   // two images [32][P], column-sum accumulators, narrow weights
  constexpr int P = WS_KMAIN + 8 * NS + 4, IMG = WS_BM * P;
  constexpr int LD = WS_N;
  const int tid = threadIdx.x, lane = tid & 63, wave = ws_uni(tid >> 6);
  const int li = lane & 31, lh = lane >> 5, n0 = wave * 64;

  int inst = 0;
  for (int i = 1; i < a.ninst; ++i)
    if ((int)blockIdx.x >= a.wg_first[i]) inst = i;
  inst = ws_uni(inst);
  const int j0 = (int)blockIdx.x - a.wg_first[inst], stride = a.wg_first[inst + 1] - a.wg_first[inst];
  const int nblk = a.blocks_per_inst;
  if (j0 >= nblk) return;
  const WsInst &I = a.inst[inst];
  const float *A0 = ws_uni(FUSE ? I.fz_h : I.A[0]);
  const float *A1 = ws_uni(I.A[1]);
  const float *ref = PLAIN ? nullptr : ws_uni(I.ref);
  const unsigned *const gm_ref = MASK ? ws_uni(I.gm_ref) : nullptr, *const gm_fz = (MASK && FUSE) ? ws_uni(I.gm_fz) : nullptr;
  float *const C = ws_uni(I.C), *const fz_out = ws_uni(I.fz_out);
  const bool keep_fz = I.fz_keep != 0;   // fused form: is the formed A0 (d pre-activation of the layer above) also stored?
  const int lda0 = FUSE ? LD : a.lda[0], lda1 = PLAIN ? ws_uni(I.lda1) : a.lda[1];
  const int k1 = PLAIN ? ws_uni(I.k1) : a.kminor[0];   // dY's columns (plain form: per instance)

  // ---- stationary weights, K-strided: wb[tn][s][c] = W0[32 (s / 4) + 16 lh + 4 (s % 4) + c][n0 + 32 tn + li]
  v4f wb[2][NSTEP];
  {
    // (gated form: the row pitch LD of the K-strided weights is compile-time - every load is base + lane offset + immediate)
    const int ldw0 = PLAIN ? ws_uni(I.ldw0) : LD;
    gcf W0 = (gcf)ws_uni(I.W[0]) + (16 * lh * ldw0 + n0 + li);
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int s = 0; s < NSTEP; ++s)
#pragma unroll
        for (int c = 0; c < 4; ++c) wb[tn][s][c] = W0[(32 * (s >> 2) + 4 * (s & 3) + c) * ldw0 + 32 * tn];
  }
  // narrow-step weights (K-strided: element (k, n) at W1[k*ldw + n]; zero beyond dY's K) in LDS, read shortly before their
  // use: [wave][slot][tn][lane][c], behind the images and the column-sum accumulators
  float *const cnw = lds + 2 * IMG + (PLAIN ? 0 : 32 * 256);
  {
    const float *W1 = ws_uni(I.W[1]);
    const int Ks = k1, ldw = PLAIN ? ws_uni(I.ldw1) : a.ldw[1];
#pragma unroll
    for (int j = 0; j < NS; ++j)
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
        v4f w;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int k = 8 * j + 4 * lh + c, kc = k < Ks ? k : Ks - 1;
          const float x = ((gcf)W1)[(long long)kc * ldw + n0 + 32 * tn + li];
          w[c] = k < Ks ? x : 0.f;
        }
        *reinterpret_cast<v4f *>(cnw + (((wave * NS + j) * 2 + tn) * 64 + lane) * 4) = w;
      }
  }
  const unsigned nw_addr = ws_lds_addr(cnw) + (unsigned)((wave * NS * 2 * 64 + lane) * 4) * 4u;
  // FUSE: head weight rows q = 0, 1 over this lane's 4 columns of a staged row
  v4f fw0 = {0.f, 0.f, 0.f, 0.f}, fw1 = {0.f, 0.f, 0.f, 0.f};
  if constexpr (FUSE) {
    const float *fzw = ws_uni(I.fz_w);
    fw0 = *(gcf4)(fzw + lane * 4);
    fw1 = *(gcf4)(fzw + (long long)a.fz_ldw + lane * 4);
  }

  const unsigned abase = ws_lds_addr(lds) + (unsigned)(li * P + 16 * lh) * 4u;
  const unsigned nbase = ws_lds_addr(lds) + (unsigned)(li * P + WS_KMAIN + 4 * lh) * 4u;
  const unsigned vo_c = (unsigned)(li * LD + 4 * lh);
  const int m_r = tid >> 3, m_c = tid & 7;
  bool m_ok[NS];
  int m_src[NS];
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    m_ok[j] = 8 * j + m_c < k1;
    m_src[j] = m_r * lda1 + (m_ok[j] ? 8 * j + m_c : 0);
  }
  float *const m_dst = lds + m_r * P + WS_KMAIN + m_c;
  const int m_tile = ws_uni(WS_BM * lda1);

  auto dma_row = [&](const float *src_tile, int img, int r) __attribute__((always_inline)) {
    __builtin_amdgcn_global_load_lds((glb_vp)(src_tile + (long long)r * lda0 + lane * 4), (lds_vp)(lds + img * IMG + r * P), 16, 0, 0);
  };

  f32x16 acc[2][2];
  // running column sums of C over this lane's rows (quad kq = 4 tn + q, component c <-> column 32 tn + 8 q + 4 lh + c): 32
  // private fp32 accumulators per lane.  They live in LDS ([quad][thread] float4: conflict-free b128), read one quad ahead
  // and written back after the add: 32 registers this kernel does not have beside 64 accumulators, the staged rows and
  // the 256 stationary weights (the allocator parked weights in scratch and re-read them every tile).
  float *const csl = lds + 2 * IMG;
  const unsigned cs_addr = ws_lds_addr(csl) + (unsigned)tid * 16u;
  v4f csq[2];
  if constexpr (!PLAIN) {
#pragma unroll
    for (int k = 0; k < 8; ++k) *reinterpret_cast<v4f *>(csl + (k * 256 + tid) * 4) = v4f{0.f, 0.f, 0.f, 0.f};
  }
  auto cs_read = [&](auto kqc) __attribute__((always_inline)) {
    constexpr int kq = decltype(kqc)::value & 7;
    if constexpr (!PLAIN && !(WS_EXP & 2)) ws_rd128<kq * 4096>(csq[kq & 1], cs_addr);
  };
  v4f fcs = {0.f, 0.f, 0.f, 0.f};   // FUSE: running column sums of the formed A0 over the rows this thread stages
  v4f sh[4];                        // FUSE: row pieces in flight (fz_h)
  unsigned shm[4] = {0, 0, 0, 0};   // FUSE + MASK: the rows' mask dwords in flight instead
  unsigned gqn = 0;                 // MASK: gate mask of the tile being accumulated
  // FUSE + MASK: this lane's dword of staged row 8 wave + 4 inside a tile's [4][64] mask block: its columns 4 lane .. 4 lane + 3
  // belong to forward wave lane / 16, lane half lane % 2; the row is that wave's lane index
  const unsigned fzm_lane = (unsigned)((lane >> 4) * 64 + 32 * (lane & 1) + 8 * wave + 4);
  v2f sdz = {0.f, 0.f};             // FUSE: dY of this wave's 8 rows of the tile in flight, row u in lane u
  float stm[NS];
  v4f nfr[2], nw0[2], nw1[2];       // operands of a narrow step (fragment, the slot's weights), requested a step ahead
  auto read_narrow = [&](auto jc, auto imc) __attribute__((always_inline)) {
    constexpr int j = decltype(jc)::value, IMX = decltype(imc)::value;
    ws_rd128<IMX * IMG * 4 + 8 * j * 4>(nfr[j & 1], nbase);
    ws_rd128<(2 * j) * 1024>(nw0[j & 1], nw_addr);
    ws_rd128<(2 * j + 1) * 1024>(nw1[j & 1], nw_addr);
  };
  v4f rq[4];                        // gate references of the quads in flight (requested three steps ahead: an L2 hit takes longer than one or two)

  // FUSE: one staged row piece: h -> LeakyReLU'(h) * (dY . Wh), k_head_dgrad's order and rounding (fma chain over q)
  auto fuse_row = [&](v4f h, float dz0, float dz1, float live) __attribute__((always_inline)) {
    v4f t;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#if WS_EXP & 16
      float x = h[c];
#else
      float x = dz0 * fw0[c];
      x = fmaf(dz1, fw1[c], x);
      x = h[c] > 0.f ? x : 0.01f * x;
#endif
      t[c] = x;
#if !(WS_EXP & 1)
      fcs[c] = fmaf(live, x, fcs[c]);   // live = 0 on the self-prefetch of a workgroup's last tile (counted once already)
#endif
    }
    return t;
  };
  // MASK: the row's dword of the layer's mask -> this lane's four gates (the nibble of its columns moved to the top bits)
  const unsigned fz_lsh = 4u * ((unsigned)(lane & 15) >> 1);
  auto fuse_row_m = [&](unsigned mrow, float dz0, float dz1, float live) __attribute__((always_inline)) {
    v4f t;
    unsigned m = mrow << fz_lsh;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float x = dz0 * fw0[c];
      x = fmaf(dz1, fw1[c], x);
      x = ws_mask_gate(m, x);
      t[c] = x;
      fcs[c] = fmaf(live, x, fcs[c]);
    }
    return t;
  };
  unsigned gq = 0;   // MASK: the gate mask of the tile being finished
  // (uniform bases once per tile, the quad's columns as instruction offsets: one scalar register pair per pointer)
  auto ref_load = [&](v4f &dst, gcf rbase, int kq) __attribute__((always_inline)) {
    if constexpr (PLAIN || MASK) return;
    if (WS_EXP & 4) return;
    const int tn = (kq >> 2) & 1, q = kq & 3;
    dst = *(gcf4)(&ws_uni(rbase + 32 * tn + 8 * q)[vo_c]);
  };
  auto quad = [&](f32x16 (&pv)[2], gf cbase, int kq) __attribute__((always_inline)) {
    const int tn = (kq >> 2) & 1, q = kq & 3;
    v4f x;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float v = pv[tn][4 * q + c];
      if constexpr (PLAIN || (WS_EXP & 4)) {
        x[c] = v;
      } else if constexpr (MASK) {
        x[c] = ws_mask_gate(gq, v);
      } else {
        const float y = rq[kq % 4][c] > 0.f ? v : 0.01f * v;
        x[c] = y;
      }
    }
    if (!(WS_EXP & 32)) *(gf4)(&ws_uni(cbase + 32 * tn + 8 * q)[vo_c]) = x;
    if constexpr (!PLAIN && !(WS_EXP & 2)) {   // (requested a quad ahead: in order behind it, the LDS write below is seen by the next tile's read)
      csq[kq & 1] += x;
      *reinterpret_cast<v4f *>(csl + ((kq & 7) * 256 + tid) * 4) = csq[kq & 1];
    }
  };

  auto block = [&](auto has_prev, auto imgc, f32x16 (&ac)[2], f32x16 (&pv)[2], int cur, int nxt, int pblk, float live) __attribute__((always_inline)) {
    constexpr bool HP = decltype(has_prev)::value;
    constexpr int IM = decltype(imgc)::value;
    constexpr int IOFF = IM * IMG * 4;
    asm volatile("s_barrier" ::: "memory");
    if constexpr (MASK) gq = gqn;   // the previous tile's gates (requested while it was accumulated)
    typedef const __attribute__((address_space(1))) unsigned *gcu;
    gcu nmask_w = (MASK && FUSE) ? (gcu)ws_uni(gm_fz + (long long)nxt * (4 * 64)) : nullptr;
    const float *nsrc = ws_uni(A0 + (long long)nxt * WS_BM * lda0);
    const float *ndz = ws_uni(A1 + (long long)nxt * m_tile);
    // FUSE: this wave stages rows 8 wave .. 8 wave + 7 of the next tile: ONE base per pointer (row 8 wave + 4), the rows
    // as signed instruction offsets (-4 .. +3 KiB)
    gcf4 nsrc_w = (gcf4)ws_uni(nsrc + (8 * wave + 4) * LD);
    gf4 nout_w = (gf4)ws_uni(fz_out + ((long long)nxt * WS_BM + 8 * wave + 4) * LD);
    gf cprev = (gf)ws_uni(C + (long long)pblk * WS_BM * LD + n0);
    gcf rprev = PLAIN ? nullptr : (gcf)ws_uni(ref + (long long)pblk * WS_BM * LD + n0);
    v4f af[2];
    ws_rd128<IOFF>(af[0], abase);
    if constexpr (HP) { ref_load(rq[0], rprev, 0); ref_load(rq[1], rprev, 1); ref_load(rq[2], rprev, 2); }
    if constexpr (HP) cs_read(std::integral_constant<int, 0>{});
    sfor<0, NSTEP>([&](auto sc) __attribute__((always_inline)) {
      constexpr int s = decltype(sc)::value;
      if constexpr (s + 1 < NSTEP) {
        constexpr int s1 = s + 1;
        ws_rd128<IOFF + (s1 >> 2) * 128 + (s1 & 3) * 16>(af[s1 & 1], abase);
        ws_lgkm_wait<1>();
      } else {
        ws_lgkm_wait<0>();
      }
      asm volatile("" : "+v"(af[s & 1]));
      if constexpr (s == 0) ws_step_a0(ac[0], ac[1], wb[0][s], wb[1][s], af[s & 1]);
      else ws_step_a(ac[0], ac[1], wb[0][s], wb[1][s], af[s & 1]);
      // ---- side work of step s
      if constexpr (HP) {
        if constexpr (s == 0) ws_anchor(pv[0], pv[1]);
        if constexpr (s < 8) {
          if constexpr (!PLAIN) asm volatile("" : "+v"(csq[s & 1]));
          if constexpr (s + 3 < 8) ref_load(rq[(s + 3) % 4], rprev, s + 3);   // three steps ahead (four: no further gain), ahead of this quad's store
          if constexpr (s + 1 < 8) cs_read(std::integral_constant<int, s + 1>{});
          quad(pv, cprev, s);
        }
      }
      if constexpr (FUSE) {
        // this wave's 8 rows of the next tile in two groups of 4 (at most 4 row pieces live in registers): requested at
        // steps 2..5 / 10..13, formed 8 steps (~4 k cycles) later at 10..13 / 18..21 - the last global store of a tile is 10
        // steps old when the tile ends (vmcnt completes in order: a younger load would wait for it)
#ifndef WS_FZ_SCHED
#define WS_FZ_SCHED 0   // measured: 0.200 ms against 0.218 ms (early schedule) for critics.dpre1+0 at config 2
#endif
#if WS_FZ_SCHED == 1
        constexpr int ureq = (s >= 2 && s < 6) ? s - 2 : ((s >= 10 && s < 14) ? s - 6 : -1);
        constexpr int uuse = (s >= 10 && s < 14) ? s - 10 : ((s >= 18 && s < 22) ? s - 14 : -1);
#else
        constexpr int ureq = (s >= 6 && s < 10) ? s - 6 : ((s >= 16 && s < 20) ? s - 12 : -1);
        constexpr int uuse = (s >= 14 && s < 18) ? s - 14 : ((s >= 24 && s < 28) ? s - 20 : -1);
#endif
        if constexpr (uuse >= 0) {  // form it, keep it for the weight gradients, put it into the other image
          constexpr int u = uuse;
          const int row = 8 * wave + u;
          // (copies first: __builtin_bit_cast of a vector ELEMENT lvalue reads the vector's first element)
          const float d0 = sdz.x, d1 = sdz.y;
          const float dzu0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, d0), u));
          const float dzu1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, d1), u));
          v4f t;
          if constexpr (MASK) t = fuse_row_m(shm[u & 3], dzu0, dzu1, live);
          else t = fuse_row(sh[u & 3], dzu0, dzu1, live);
          if (keep_fz && !(WS_EXP & 8)) (nout_w + (u - 4) * (LD / 4))[(unsigned)lane] = t;   // (workgroup-uniform: frozen instances have no weight gradients)
          *reinterpret_cast<v4f *>(lds + (IM ^ 1) * IMG + row * P + lane * 4) = t;
        }
        if constexpr (ureq >= 0) {   // (after the use of the register it refills)
          if constexpr (MASK) shm[ureq & 3] = nmask_w[fzm_lane + (ureq - 4)];
          else sh[ureq & 3] = (nsrc_w + (ureq - 4) * (LD / 4))[(unsigned)lane];
          if constexpr (ureq == 0) sdz = *(const __attribute__((address_space(1))) v2f *)(ndz + (8 * wave + (lane & 7)) * lda1);
        }
      } else {
        if constexpr (s >= 16 && s < 16 + WS_BM / 4) dma_row(nsrc, IM ^ 1, wave + 4 * (s - 16));
      }
      if constexpr (s == 8) {
#pragma unroll
        for (int j = 0; j < NS; ++j) stm[j] = ((gcf)ndz)[m_src[j]];
        if constexpr (MASK) gqn = ((gcu)ws_uni(gm_ref + ((long long)cur * 4 + wave) * 64))[(unsigned)lane];   // this tile's gates, used a block later
      }
      if constexpr (s == 26) {
#pragma unroll
        for (int j = 0; j < NS; ++j) m_dst[(IM ^ 1) * IMG + 8 * j] = m_ok[j] ? stm[j] : 0.f;
      }
      if constexpr (s == NSTEP - 1) read_narrow(std::integral_constant<int, 0>{}, imgc);   // the first narrow step's operands
      asm volatile("" ::: "memory");
    });
    sfor<0, NS>([&](auto jc) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value;
      if constexpr (j + 1 < NS) { read_narrow(std::integral_constant<int, j + 1>{}, imgc); ws_lgkm_wait<3>(); }
      else ws_lgkm_wait<0>();
      asm volatile("" : "+v"(nfr[j & 1]), "+v"(nw0[j & 1]), "+v"(nw1[j & 1]));
      ws_step_v(ac[0], ac[1], nw0[j & 1], nw1[j & 1], nfr[j & 1]);
    });
    // this wave's part of the next image has landed: LDS-DMA pieces (vmcnt) or, FUSE, its own LDS writes only - the
    // global stores of the tile need not have completed
#ifndef WS_FZ_WAIT
#define WS_FZ_WAIT 1
#endif
    if constexpr (FUSE && WS_FZ_WAIT) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  };
  using T = std::true_type;
  using F = std::false_type;
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  // ---- first image (once per workgroup)
  int blk = j0;
  {
    const float *src = A0 + (long long)blk * WS_BM * lda0;
    if constexpr (FUSE) {
      const float *dzs = A1 + (long long)blk * m_tile;
      float *out = fz_out + (long long)blk * WS_BM * LD;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int row = 8 * wave + u;
        const v2f dz = *(const __attribute__((address_space(1))) v2f *)(dzs + row * lda1);
        v4f t;
        if constexpr (MASK) {
          const unsigned mrow = ((const __attribute__((address_space(1))) unsigned *)ws_uni(gm_fz + (long long)blk * (4 * 64)))[fzm_lane + (u - 4)];
          t = fuse_row_m(mrow, dz.x, dz.y, 1.f);
        } else {
          const v4f h = ((gcf4)(src + row * LD))[(unsigned)lane];
          t = fuse_row(h, dz.x, dz.y, 1.f);
        }
        if (keep_fz) ((gf4)(out + row * LD))[(unsigned)lane] = t;
        *reinterpret_cast<v4f *>(lds + row * P + lane * 4) = t;
      }
    } else {
#pragma unroll
      for (int u = 0; u < WS_BM / 4; ++u) dma_row(src, 0, wave + 4 * u);
    }
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      const float x = ((gcf)(A1 + (long long)blk * m_tile))[m_src[j]];
      m_dst[8 * j] = m_ok[j] ? x : 0.f;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  }
  int nxt = blk + stride < nblk ? blk + stride : blk;
  block(F(), I0(), acc[0], acc[1], blk, nxt, 0, nxt != blk ? 1.f : 0.f);
  int prv = blk, set = 1;
  blk += stride;
#pragma unroll 1
  while (blk < nblk) {
    nxt = blk + stride < nblk ? blk + stride : blk;
    block(T(), I1(), acc[1], acc[0], blk, nxt, prv, nxt != blk ? 1.f : 0.f);
    prv = blk; blk += stride; set = 0;
    if (blk >= nblk) break;
    nxt = blk + stride < nblk ? blk + stride : blk;
    block(T(), I0(), acc[0], acc[1], blk, nxt, prv, nxt != blk ? 1.f : 0.f);
    prv = blk; blk += stride; set = 1;
  }
  auto flush = [&](f32x16 (&pv)[2]) __attribute__((always_inline)) {
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    ws_anchor(pv[0], pv[1]);
    gf cprev = (gf)ws_uni(C + (long long)prv * WS_BM * LD + n0);
    gcf rprev = PLAIN ? nullptr : (gcf)ws_uni(ref + (long long)prv * WS_BM * LD + n0);
    if constexpr (MASK) gq = gqn;
    sfor<0, 8>([&](auto kc) __attribute__((always_inline)) {
      constexpr int kq = decltype(kc)::value;
      ref_load(rq[kq % 4], rprev, kq);
      cs_read(kc);
      ws_lgkm_wait<0>();
      if constexpr (!PLAIN) asm volatile("" : "+v"(csq[kq & 1]));
      quad(pv, cprev, kq);
    });
  };
  if (set == 1) flush(acc[0]);
  else flush(acc[1]);
  // ---- the workgroup's partial rows of the column sums
  if constexpr (!PLAIN) {
    gf colsum = (gf)ws_uni(I.colsum) + (long long)j0 * WS_N;
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float t = ws_sum32(csl[((4 * tn + (r >> 2)) * 256 + tid) * 4 + (r & 3)]);
        if (li == 31) colsum[n0 + 32 * tn + (r & 3) + 8 * (r >> 2) + 4 * lh] = t;
      }
    if constexpr (FUSE) {
      __syncthreads();   // every wave is done with the images
      *reinterpret_cast<v4f *>(lds + wave * WS_N + lane * 4) = fcs;
      __syncthreads();
      ((gf)ws_uni(I.fz_colsum) + (long long)j0 * WS_N)[tid] = (lds[tid] + lds[WS_N + tid]) + (lds[2 * WS_N + tid] + lds[3 * WS_N + tid]);
    }
  }
}

}  // namespace

bool wstat_from_problems(const GemmProblem *probs, int nprob, WsArgs &args) {
  if (nprob < 1 || nprob > WS_MAX_INST) return false;
  memset(&args, 0, sizeof(args));
  auto aligned = [](const void *p, uintptr_t n) { return (reinterpret_cast<uintptr_t>(p) & (n - 1)) == 0; };
  auto main_of = [](const GemmProblem &p) {
    int m = -1;
    for (int s = 0; s < p.nseg; ++s)
      if (p.seg[s].K == WS_KMAIN) { if (m >= 0) return -1; m = s; }
    return m;
  };
  auto is_dual = [](const GemmProblem &p) { return p.emit_seg >= 0 && p.emit_seg < p.nseg - 1; };
  // the launch's shape: that of its two-output problems if it has any (plain problems may ride along: a critic's target
  // pass beside the online + frozen pass of layer 0), else that of the first problem
  int ref_i = 0;
  bool dual = false;
  for (int i = 0; i < nprob; ++i)
    if (is_dual(probs[i])) { ref_i = i; dual = true; break; }
  // plain dgrad problems (shares of an input gradient) may differ in their narrow segment: the widest one shapes the launch
  bool all_plain = !dual;
  for (int i = 0; i < nprob && all_plain; ++i) {
    const GemmProblem &p = probs[i];
    all_plain = p.epi == EPI_NONE && p.nseg == 2 && main_of(p) >= 0 && !p.seg[main_of(p)].b_kc && main_of(p) == main_of(probs[0]);
  }
  if (all_plain)
    for (int i = 1; i < nprob; ++i)
      if (probs[i].seg[1 - main_of(probs[i])].K > probs[ref_i].seg[1 - main_of(probs[ref_i])].K) ref_i = i;
  const GemmProblem &p0 = probs[ref_i];
  const int main0 = main_of(p0);
  if (main0 < 0 || p0.N != WS_N || p0.M % WS_BM || p0.M < WS_BM || p0.ksplit != 1 || p0.ldc != WS_N) return false;
  // dgrad forms: K-strided weights; gated (EPI_LRELU_GRAD: a hidden layer's pre-activation gradient) or plain (EPI_NONE: a
  // network's share of an input gradient)
  const bool grad = !p0.seg[main0].b_kc;
  const bool plain = grad && p0.epi == EPI_NONE;
  if (grad ? (p0.epi != EPI_LRELU_GRAD && !plain) : p0.epi != EPI_LRELU) return false;
  const int nminor = p0.nseg - 1;
  if (nminor > WS_MAX_MINOR) return false;
  const bool fz = p0.fz_h != nullptr;
  if (plain) {
    if (p0.bias || p0.ref || p0.colsum || p0.hf_w || dual || nminor != 1 || fz) return false;
    // (instances may differ in dY's width / pitches: the launch takes the widest, WsInst carries each one's own)
  } else if (grad) {
    if (p0.bias || !p0.ref || p0.ldref != WS_N || !p0.colsum || p0.hf_w || dual || nminor != 1) return false;
    if (p0.seg[main0].ldb != WS_N) return false;   // K-strided weights: compile-time row pitch
    if (fz && (p0.seg[1 - main0].K != 2 || p0.seg[1 - main0].lda % 2)) return false;   // dY rows read as float2
  } else {
    if (!p0.bias || p0.colsum || p0.ref || fz) return false;
    if (dual && (p0.emit_seg != p0.nseg - 2 || main0 == p0.nseg - 1 || !p0.C2 || p0.ldc2 != WS_N || nminor != 2)) return false;
    if (!dual && nminor > 1) return false;
    if (p0.hf_w && p0.hf_q != 2) return false;
  }
  args.M = p0.M; args.ninst = nprob; args.blocks_per_inst = p0.M / WS_BM;
  args.nminor = nminor; args.dual = dual; args.grad = grad ? (plain ? 2 : 1) : 0; args.fz = fz; args.fz_ldw = p0.fz_ldw;
  args.hf_q = p0.hf_w ? p0.hf_q : 0; args.hf_ldw = p0.hf_ldw;
  int cost[WS_MAX_INST], cost_sum = 0;
  const int dual_cost = 12;
  for (int i = 0; i < nprob; ++i) {
    const GemmProblem &p = probs[i];
    const bool pd = is_dual(p);
    const bool plain_in_dual = dual && !pd;   // one narrow segment and one output fewer than the launch's shape
    if (plain_in_dual && (p.nseg != p0.nseg - 1 || (p.emit_seg >= 0 && p.emit_seg != p.nseg - 1) || p.C2)) return false;
    if (!plain_in_dual && (p.nseg != p0.nseg || p.emit_seg != p0.emit_seg)) return false;
    if (p.M != p0.M || p.N != p0.N || p.ksplit != 1 || p.epi != p0.epi || main_of(p) != main0 ||
        (p.hf_w != nullptr) != (p0.hf_w != nullptr) || p.hf_q != p0.hf_q || p.hf_ldw != p0.hf_ldw || p.ldc != p0.ldc ||
        (pd && p.ldc2 != p0.ldc2) || (p.bias != nullptr) != (p0.bias != nullptr) || (p.ref != nullptr) != (p0.ref != nullptr) ||
        p.ldref != p0.ldref || (p.colsum != nullptr) != (p0.colsum != nullptr) || (p.fz_h != nullptr) != fz || p.fz_ldw != p0.fz_ldw)
      return false;
    if (pd && (!p.C2 || (p.hf_w && !p.hf_out2))) return false;
    if (p.hf_w && (!p.hf_out || !aligned(p.hf_out, 8) || (p.hf_out2 && !aligned(p.hf_out2, 8)))) return false;
    if (!aligned(p.C, 16) || (p.C2 && !aligned(p.C2, 16)) || (p.bias && !aligned(p.bias, 16)) || (p.ref && !aligned(p.ref, 16))) return false;
    if (fz && (!p.fz_w || !p.fz_out || !p.fz_colsum || !aligned(p.fz_h, 16) || !aligned(p.fz_out, 16) || !aligned(p.fz_w, 4)))
      return false;
    WsInst I;
    memset(&I, 0, sizeof(I));
    int m = 0;
    for (int s = 0; s < p.nseg; ++s) {
      const GemmSeg &sg = p.seg[s], &s0 = p0.seg[s];
      const int slot = s == main0 ? 0 : 1 + m++;
      if (!sg.a_kc || sg.b_kc != (grad ? 0 : 1)) return false;
      if (plain ? (slot == 0 ? (sg.K != s0.K || sg.lda != s0.lda) : sg.K > s0.K) : (sg.K != s0.K || sg.lda != s0.lda || sg.ldb != s0.ldb)) return false;
      if (slot == 0) I.ldw0 = sg.ldb;
      if (slot == 1) { I.k1 = sg.K; I.lda1 = sg.lda; I.ldw1 = sg.ldb; }
      if (slot > 0 && (sg.K < 1 || sg.K > (plain ? 40 : 32))) return false;   // (plain form: up to 5 steps - an actor's 2 x 17 logits)
      if (slot == 0 && !fz && (sg.lda % 4 || !aligned(sg.A, 16))) return false;   // rows move as 16-byte pieces
      if (slot > 0 && fz && !aligned(sg.A, 8)) return false;
      I.A[slot] = sg.A; I.W[slot] = sg.B;
    }
    I.bias = p.bias; I.C = p.C; I.C2 = p.C2;
    I.hf_w = p.hf_w; I.hf_out = p.hf_out; I.hf_out2 = p.hf_out2;
    I.ref = p.ref; I.colsum = p.colsum;
    I.fz_h = p.fz_h; I.fz_w = p.fz_w; I.fz_out = p.fz_out; I.fz_colsum = p.fz_colsum; I.fz_keep = p.fz_discard ? 0 : 1;
    I.gm_out = p.gm_out; I.gm_out2 = p.gm_out2; I.gm_ref = p.gm_ref; I.gm_fz = p.gm_fz;
    args.inst[i] = I;
    cost[i] = pd ? dual_cost : 11;   // a two-output instance finishes twice as many register quads per tile
    cost_sum += cost[i];
  }
  {
    int m = 0;
    for (int s = 0; s < p0.nseg; ++s) {
      const int slot = s == main0 ? 0 : 1 + m++;
      args.lda[slot] = p0.seg[s].lda; args.ldw[slot] = p0.seg[s].ldb;
      if (slot > 0) args.kminor[slot - 1] = p0.seg[s].K;
    }
    // 8-k steps: the narrow segments in order; a two-output launch's last segment is its tail
    int ns = 0;
    for (int sg = 0; sg < nminor; ++sg) {
      const int steps = (args.kminor[sg] + 7) / 8;
      const bool tail = dual && sg == nminor - 1;
      for (int j = 0; j < steps; ++j) {
        if (ns >= WS_MAX_SLOTS) return false;
        args.slot_seg[ns] = sg; args.slot_k0[ns] = 8 * j;
        ++ns;
      }
      (tail ? args.nslot_tail : args.nslot_loop) += steps;
    }
    // instantiated forms (wstat_launch)
    const int L = args.nslot_loop, Tn = args.nslot_tail;
    if (grad) {
      if (!((L == 1) || (L == 4 && !fz) || ((L == 2 || L == 5) && plain))) return false;
    } else {
      const bool small = (L == 0 && Tn == 0) || (L == 1 && Tn == 0) || (L == 1 && Tn == 1);
      const bool wide = (L == 3 && Tn == 0) || (L == 3 && Tn == 3);
      if (!(small || (wide && !args.hf_q))) return false;
    }
  }
  // workgroups per instance: a share of the CUs in proportion to the instance's work (at most one per tile)
  int dev = 0;
  static int ncu_of[64];
  static std::mutex mu;
  {
    std::lock_guard<std::mutex> lk(mu);
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    if (!ncu_of[dev]) {
      hipDeviceProp_t pr;
      if (hipGetDeviceProperties(&pr, dev) != hipSuccess) return false;
      ncu_of[dev] = pr.multiProcessorCount;
    }
  }
  const int ncu = cu_budget(ncu_of[dev]);
  if (nprob > ncu) return false;
  // Workgroups per instance: one per CU in all, dealt so that the slowest instance finishes as early as possible (tiles come in
  // whole numbers: 392 tiles over 26 workgroups are 16 each, over 27 they are 15).
  int per[WS_MAX_INST], cap = args.blocks_per_inst;
  // the dgrad form's column sums: one partial row per workgroup in buffers sized for one per 64 rows
  if (grad && !plain && cap > (p0.M + 63) / 64) cap = (p0.M + 63) / 64;
  const int tiles = args.blocks_per_inst;
  for (int i = 0; i < nprob; ++i) per[i] = 1;
  auto time_of = [&](int i, int w) { return (long long)cost[i] * ((tiles + w - 1) / w); };
  for (int left = ncu - nprob; left > 0; --left) {
    int worst = -1;
    for (int i = 0; i < nprob; ++i)
      if (per[i] < cap && (worst < 0 || time_of(i, per[i]) > time_of(worst, per[worst]))) worst = i;
    if (worst < 0) break;
    ++per[worst];
  }
  for (int i = 0; i < nprob; ++i)   // a workgroup that does not lower its instance's tile count only repeats the prologue
    while (per[i] > 1 && (tiles + per[i] - 2) / (per[i] - 1) == (tiles + per[i] - 1) / per[i]) --per[i];
  // gated dgrad form: the caller reduces wstat_colsum_rows() partial rows of column sums for EVERY instance, so every instance
  // gets the same number of workgroups (the smallest share: dealing by remainder can leave them one apart)
  if (grad && !plain) {
    int mn = per[0];
    for (int i = 1; i < nprob; ++i) mn = per[i] < mn ? per[i] : mn;
    for (int i = 0; i < nprob; ++i) per[i] = mn;
  }
  args.wg_first[0] = 0;
  for (int i = 0; i < nprob; ++i) args.wg_first[i + 1] = args.wg_first[i] + per[i];
  return true;
}

#ifdef WS_STAMPS
extern "C" int wstat_debug_stamps(unsigned long long *out, int cap) {
  static unsigned long long h[1024 * 8];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_ws_stamps), sizeof(h)) != hipSuccess) return -1;
  const int n = cap < 1024 * 8 ? cap : 1024 * 8;
  for (int i = 0; i < n; ++i) out[i] = h[i];
  return n;
}
#endif

double wstat_flops(const WsArgs &a) {
  double k = WS_KMAIN;
  for (int s = 0; s < a.nminor; ++s) k += a.kminor[s];
  return 2.0 * a.M * (double)WS_N * k * a.ninst;
}

template <typename K>
static hipError_t ws_launch_kernel(K kern, int lds_bytes, bool (&attr)[64], const WsArgs &a, hipStream_t s) {
  // the opt-in to > 64 KiB of dynamic LDS belongs to the (device, function) pair
  static std::mutex mu;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
  {
    std::lock_guard<std::mutex> lk(mu);
    if (!attr[dev]) {
      e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
      if (e != hipSuccess) return e;
      attr[dev] = true;
    }
  }
  hipLaunchKernelGGL(kern, dim3(a.wg_first[a.ninst]), dim3(256), lds_bytes, s, a);
  return hipGetLastError();
}
template <int NSL, int NST, int HFQ, bool GM>
static hipError_t ws_launch_gm(const WsArgs &a, hipStream_t s) {
  static bool attr[64];
  // two images + per-wave constants (bias, rider weights, 8 KiB of narrow weights per slot)
  constexpr int lds_need = (2 * WS_BM * (WS_KMAIN + 8 * (NSL + NST) + 4) + 4 * 2 * 32 + 4 * 8 * 32 + (NSL + NST) * 2048 + 2 * 2 * 4 * 32 * HFQ) * 4;
  constexpr int lds_stage = 4 * 32 * (WS_KMAIN + 4) * 4;   // the weight load's transposition area (4 waves x 32 rows)
  constexpr int lds_bytes = lds_need > lds_stage ? lds_need : lds_stage;
  return ws_launch_kernel(&k_wstat<NSL, NST, HFQ, GM>, lds_bytes, attr, a, s);
}
template <int NSL, int NST, int HFQ>
static hipError_t ws_launch(const WsArgs &a, hipStream_t s) {
  bool gm = false;
  for (int i = 0; i < a.ninst; ++i) gm = gm || a.inst[i].gm_out || a.inst[i].gm_out2;
  return gm ? ws_launch_gm<NSL, NST, HFQ, true>(a, s) : ws_launch_gm<NSL, NST, HFQ, false>(a, s);
}
template <bool FUSE, bool PLAIN, int NS, bool MASK = false>
static hipError_t ws_launch_grad(const WsArgs &a, hipStream_t s) {
  static bool attr[64];
  // images + column-sum accumulators + narrow weights
  constexpr int lds_bytes = 2 * WS_BM * (WS_KMAIN + 8 * NS + 4) * 4 + (PLAIN ? 0 : 32 * 256 * 4) + NS * 2048 * 4;
  return ws_launch_kernel(&k_wstat_grad<FUSE, PLAIN, NS, MASK>, lds_bytes, attr, a, s);
}

hipError_t wstat_launch(const WsArgs &a, hipStream_t s) {
  const int L = a.nslot_loop, T = a.nslot_tail;
  if (a.grad == 2)
    return L == 5 ? ws_launch_grad<false, true, 5>(a, s)
                  : (L == 4 ? ws_launch_grad<false, true, 4>(a, s) : (L == 2 ? ws_launch_grad<false, true, 2>(a, s) : ws_launch_grad<false, true, 1>(a, s)));
  if (a.grad) {
    if (a.use_masks) {   // (set by the plan when the forward launches wrote the masks: every instance carries gm_ref, and gm_fz if fused)
      if (a.fz) return ws_launch_grad<true, false, 1, true>(a, s);
      return L == 4 ? ws_launch_grad<false, false, 4, true>(a, s) : ws_launch_grad<false, false, 1, true>(a, s);
    }
    if (a.fz) return ws_launch_grad<true, false, 1>(a, s);
    return L == 4 ? ws_launch_grad<false, false, 4>(a, s) : ws_launch_grad<false, false, 1>(a, s);
  }
  if (L == 3) return T == 3 ? ws_launch<3, 3, 0>(a, s) : ws_launch<3, 0, 0>(a, s);
  if (L == 0) return a.hf_q ? ws_launch<0, 0, 2>(a, s) : ws_launch<0, 0, 0>(a, s);
  if (T == 1) return a.hf_q ? ws_launch<1, 1, 2>(a, s) : ws_launch<1, 1, 0>(a, s);
  return a.hf_q ? ws_launch<1, 0, 2>(a, s) : ws_launch<1, 0, 0>(a, s);
}

}  // namespace fdql
