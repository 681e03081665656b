// Weight-stationary persistent row-block GEMM (wstat.hip) for the forward layers of the critic ensemble: a launch is a set
// of INSTANCES (critic k x {target, online, frozen}) of one Linear layer of 256 outputs,
//   C[M, 256] = LeakyReLU( A0[M, 256] W0^T + A1[M, K1] W1^T (+ A2[M, K2] W2^T) + bias )      K1, K2 <= 32 (the action columns)
// and every workgroup belongs to ONE instance for its whole life: the instance's 256 x 256 weights sit in the AccVGPRs
// of its four waves (64 output columns x 256 k = 256 registers per lane), so the K loop issues no weight loads at all.
#pragma once
#include "common.h"

namespace fdql {

constexpr int WS_BM = 32;          // rows per tile
constexpr int WS_N = 256;          // output columns (4 waves x 64)
constexpr int WS_KMAIN = 256;      // K of the main segment
constexpr int WS_MAX_MINOR = 2;    // narrow K-segments beside it (K <= 32 each: ceil(K / 8) MFMA steps of 8 k)
constexpr int WS_MAX_SLOTS = 8;    // 8-k steps of all narrow segments together
constexpr int WS_MAX_INST = 16;    // the table travels in the kernel arguments (scalar loads): 16 x 176 B + header < 4 KiB

struct WsInst {
  const float *A[1 + WS_MAX_MINOR];   // activations of the segments, [0] = main; row-major, K contiguous
  const float *W[1 + WS_MAX_MINOR];   // weights of the segments, K contiguous: element (n, k) at W[n*ldw + k]
  const float *bias;                  // [256]
  float *C, *C2;                      // outputs [M, 256]; C2: second output of a dual launch
  const float *hf_w;                  // head fusion (common.h, GemmProblem::hf_*): head weight rows over this layer's columns
  float *hf_out, *hf_out2;
  // dgrad form
  const float *ref;                   // activation output whose sign gates the gradient
  float *colsum;                      // [workgroups of the instance, 256] column sums of C (one partial row per workgroup)
  const float *fz_h, *fz_w;           // fused head dgrad (GemmProblem::fz_*)
  float *fz_out, *fz_colsum;          // fz_colsum: [workgroups of the instance, 256]
  // plain dgrad form: the networks whose shares are summed differ in their narrow segment (critic: dz, 2 columns of a head of
  // pitch 774; actor: d logits, 12 columns of a head of pitch 512) and in the row pitch of their layer-0 weights
  int k1, lda1, ldw0, ldw1;
  int fz_keep, pad;                   // fused form: 1 = store the formed A0 rows to fz_out (0: only the GEMM consumes them)
  unsigned *gm_out, *gm_out2;         // forward: gate masks of the outputs (GemmProblem::gm_*), null: none
  const unsigned *gm_ref, *gm_fz;     // dgrad forms with WsArgs::use_masks: the masks that stand in for ref / fz_h
};

struct WsArgs {
  int M;                     // rows per instance, multiple of 32
  int ninst, blocks_per_inst;
  int nminor, kminor[WS_MAX_MINOR];
  int nslot_loop, nslot_tail;          // 8-k steps of the narrow segments: inside a tile's K loop / of a two-output launch's last segment
  int slot_seg[WS_MAX_SLOTS], slot_k0[WS_MAX_SLOTS];   // slot -> (narrow segment, its first k)
  int lda[1 + WS_MAX_MINOR], ldw[1 + WS_MAX_MINOR];
  int dual;                  // 1: C = f(all segments but the last), C2 = f(all)
  int hf_q, hf_ldw;          // head fusion: outputs per row (2), 0 = off
  int use_masks;             // dgrad forms: 1 = gate by WsInst::gm_ref / gm_fz (written by the forward launches) instead of ref / fz_h
  int hf_presum;             // head fusion: 1 = a tile's eight column planes are summed in the kernel, hf_out holds ONE plane [M][hf_q]
  int grad;                  // dgrad forms - weights K-strided (element (k, n) at W[k*ldw + n]): 1: x *= LeakyReLU'(ref), column sums;
                             // 2: plain (no gate, no sums: one network's share of an input gradient)
  int fz, fz_ldw;            // dgrad form: A0 formed from (fz_h, dY = A1, fz_w) while it is staged
  int wg_first[WS_MAX_INST + 1];   // workgroups [wg_first[i], wg_first[i+1]) serve instance i (block j, j + n, ... of it)
  WsInst inst[WS_MAX_INST];
};

// Can these problems (one launch group) run as one weight-stationary launch?  Fills args when they can.
// Forward problems: EPI_LRELU with a bias.  dgrad problems (EPI_LRELU_GRAD, K-strided weights, one narrow segment = dY, ref
// and colsum set; optionally the fused head dgrad): NOTE the column sums (colsum, fz_colsum) then hold
// wstat_colsum_rows(args) partial rows per instance instead of one per 64 rows - the caller sizes its reduction by it.
bool wstat_from_problems(const GemmProblem *probs, int nprob, WsArgs &args);
inline int wstat_colsum_rows(const WsArgs &a) { return a.wg_first[1] - a.wg_first[0]; }
hipError_t wstat_launch(const WsArgs &args, hipStream_t stream);
double wstat_flops(const WsArgs &a);

}  // namespace fdql
