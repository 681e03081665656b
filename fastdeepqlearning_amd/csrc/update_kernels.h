// Argument blocks and launchers of the non-GEMM update kernels (kernels.hip).
#pragma once
#include "common.h"

namespace fdql {

// Small state that lives on the device so a captured graph can be replayed unchanged.
struct DevState {
  int step;             // optimiser step (torch.optim.Adam state['step'])
  float alpha_cur;      // exp(log_alpha) as of the previous step (soft_actor_critic.py:41,152)
  float alpha_next;
  float neg_step_size;  // -lr / (1 - beta1^(step+1)): of the step about to be applied (k_loss_finish)
  float bc2_sqrt;       // sqrt(1 - beta2^(step+1))
  unsigned loss_blocks_done;   // k_loss workgroups that have published their partial row (fused finish; back to 0 when the step's loss is done)
  float pad[2];
};

constexpr int LOSS_NPART = 8;

struct PolicyFwdArgs {
  const float *logits;  // [M, 2A]
  const float *noise;   // [M, A] or null -> Philox
  float *noise_out;     // [M, A] or null
  float *action;        // [M, A]
  float *logp;          // [M]
  uint32_t which;       // RNG stream id
  const float *sub;     // optional [M, A]: also write diff = action - sub (the critic-input delta between
  float *diff;          //   cat(s, pi) and cat(s, a); see the dual layer-0 problem in agent.hip)
};

struct LossArgs {
  int M, B, Nq, Nt, G;
  int distributional, lowerbound, max_entropy;
  float gamma, target_entropy, half_inv_nq;
  const DevState *st;
  const float *log_alpha;
  const float *z_target;   // [M, Nq]
  const float *q_pred;     // [M, Nq]
  const float *z_frozen;   // [M, Nq]
  const float *logp_next;  // [M]
  const float *logp;       // [M]
  const float *reward, *task_done, *mc_return;  // [T*B]; row m reads m + B (the "next" record)
  const float *w;          // [M]
  float *dz, *dzf;         // [M, Nq]
  float *td_target;        // [M, Nt]
  float *q_loss, *pi_loss, *alpha_loss;  // [M]
  float *partials;         // [blocks, LOSS_NPART]
  // optional (few workgroups: temporal_len 2): the LAST workgroup to publish its partial row also does k_loss_finish's work
  // - one launch less in a latency-bound step; the partial rows are summed in the same fixed order either way
  const struct LossFinishArgs *fin;
};
struct LossFinishArgs {   // lives in the workspace (one pointer in the kernel arguments)
  int nblocks, M, Nq, pad;
  DevState *st;
  float *scalars, *dlog_alpha;
  double lr, b1, b2;
};

struct AdamArgs {
  long long n;
  float *params, *m, *v;
  const float *grads;
  const float *slabs;   // optional [nslab][n]: gradient = sum of the slabs (written to grads_out), else read from grads
  float *grads_out;
  int nslab;
  float grad_scale, one_minus_b1, b2, one_minus_b2, eps;
  DevState *st;
  float *targets;
  long long tgt_begin, tgt_end;
  float tau, one_minus_tau;
  int hard;
  float *frozen;
  long long frozen_begin, frozen_end;
};

// ---- act(): one Linear layer over a few rows, input = K-segments (cat never materialised)
constexpr int ACT_MAX_SEG = 8;
struct ActSeg {
  const float *ptr;
  int ld, width;
};
struct ActLayerArgs {
  ActSeg in[ACT_MAX_SEG];
  int nseg;
  const float *W;     // [N, ldw] row-major (torch Linear.weight)
  int ldw;
  const float *bias;  // [N]
  float *out;         // [rows, ldo]
  int ldo, N, rows, leaky;
  // Optional pre-layer (a first hidden layer over FEW input columns - the observation): every workgroup of the launch
  // recomputes h = LeakyReLU(pre_W cat(in[0 .. pre_nseg)) + pre_bias) [rows, pre_N <= 256] into LDS; the segment whose ptr is
  // null (width pre_N) reads it.  One dependent launch less per act(); 0: off.
  const float *pre_W, *pre_bias;
  int pre_ldw, pre_N, pre_nseg;
};
struct ActPolicyArgs {
  const float *logits;  // [rows, ld]: (mean | log_std) or discrete logits
  int ld, rows, A, discrete;
  const uint8_t *exploit_mask;  // [rows] or null (= explore everywhere)
  const float *noise;           // [rows, A] N(0,1) (continuous) / U(0,1) (discrete), or null -> Philox(seed, counter)
  uint64_t seed, counter;
  float *action;                // [rows, A] continuous; [rows] action index as float (discrete)
  float *log_prob, *explore, *exploit;  // optional
};
hipError_t act_layer_launch(const ActLayerArgs &a, hipStream_t s);
hipError_t act_policy_launch(const ActPolicyArgs &a, hipStream_t s);
// the policy's (narrow: N <= 32, A <= 16 / <= 32 discrete) last layer and the sample / select in ONE launch: a 1024-thread workgroup per 8 rows
// computes the logits into LDS and continues with the policy head (p.logits / p.ld are ignored)
bool act_head_policy_takes(const ActLayerArgs &l, const ActPolicyArgs &p);
hipError_t act_head_policy_launch(const ActLayerArgs &l, const ActPolicyArgs &p, hipStream_t s);

hipError_t head_dgrad_launch(const HeadDgradProblem *dev, int n, int total_blocks, hipStream_t s);
hipError_t skinny_wgrad_launch_host(const SkinnyWgradProblem *host, const SkinnyWgradProblem *dev, int n,
                                    int total_blocks, hipStream_t s);
// Streaming form (kernels.hip, k_stream_wgrad): Nout <= 36 (STREAM_WGRAD_MAX_OUT) and X of exactly 256 columns (pitch 256: four single-wave
// workgroups per slab) or of <= 32 columns (one).
bool stream_wgrad_takes(const SkinnyWgradProblem &p);
int stream_wgrad_finalize(SkinnyWgradProblem *p, int n);
hipError_t stream_wgrad_launch(const SkinnyWgradProblem *host, const SkinnyWgradProblem *dev, int n, int total_blocks, hipStream_t s);
hipError_t loss_launch(const LossArgs &a, hipStream_t s);
// use_bootstrap_minibatch_nstep (soft_actor_critic.py:102-132, deepQlearning.py:226-228), SAC-min only:
//   bound[b] = sum_t gamma^t r[t+1][b] + gamma^(T-1) td_target[T-2][b];
//   term[b][j] = prod_t mask[t+1][b] * relu(bound[b] - q_pred[0][b][j]) * prod_t is_contiguous[t][b]
// loss += mean_{b,j} term / (world * T); d loss / d q_pred[0] added into dz; loss value -> partial_row[0]
struct BootArgs {
  int T, B, Nq;
  float gamma, scale;           // scale = 1 / (B * Nq * world_size * T)
  const float *reward, *task_done, *contig, *td_target, *q_pred;
  float *dz;
  float *partial_row;           // [LOSS_NPART]
};
hipError_t boot_lowerbound_launch(const BootArgs &a, hipStream_t s);
hipError_t loss_finish_launch(const float *partials, int nblocks, int M, int Nq, DevState *st, float *scalars,
                              float *dlog_alpha, double lr, double b1, double b2, hipStream_t s);
hipError_t reduce_slabs_range_launch(const float *slabs, int nslab, long long stride, long long first, long long count, float *grads,
                                     hipStream_t s);
hipError_t reduce_slabs_launch(const float *slabs, int nslab, long long n, float *grads, hipStream_t s);
hipError_t adam_launch(const AdamArgs &a, hipStream_t s);
hipError_t prep_launch(const float *task_done, const float *episode_step, int T, int B, int burn_in, int cumprod,
                       float inv_gb, float *w, float *contig, DevState *st, const float *log_alpha, hipStream_t s);
// ---- pixel encoder (design of this build; no reference): convolutions as im2col + the grouped GEMM
struct ConvGeom {
  int C, H, W;      // input feature map
  int k, s;         // square kernel, stride (no padding)
  int OH, OW;       // output positions
};
// row (img, oy, ox) of col = the k x k x C window at (oy*s, ox*s).  Layer 0 reads the batch's NCHW frames (times
// `scale`) and orders K as (c, ky, kx); later layers read the previous layer's NHWC output [img*H*W, C] (C % 4 == 0)
// and order K as (ky, kx, c), so both sides move contiguous runs.  The weights [Cout, K] use the same K order.
hipError_t im2col_launch(const float *in, int nhwc, float scale, long long n_img, const ConvGeom &g, float *col, hipStream_t s);
// d(pre-activation of the previous layer) = col2im(dcol) * LeakyReLU'(act_prev), NHWC [n_img*H*W, C]:
// a gather over the <= ceil(k/s)^2 windows that cover a pixel (fixed order, no atomics)
hipError_t col2im_mask_launch(const float *dcol, const float *act_prev, long long n_img, const ConvGeom &g, float *dpre_prev,
                              hipStream_t s);
// column sums of a tall matrix X [R, C] (C <= 256) in two levels: partial[blk][c] over COLSUM_TALL_ROWS rows per block
constexpr int COLSUM_TALL_ROWS = 1024;
inline int colsum_tall_blocks(long long R) { return (int)((R + COLSUM_TALL_ROWS - 1) / COLSUM_TALL_ROWS); }
hipError_t colsum_tall_launch(const float *X, long long R, int C, int ld, float *partial, hipStream_t s);
// dst[e] = sum_p part[p][e] for e < n (fixed order)
hipError_t summaries_launch(const float *q_pred, int M, int Nq, const float *ic, int Tm1, int B, int temporal_len, const float *grads,
                            const long long *ranges_dev, int nranges, float *out_dev, hipStream_t s);
// dst[rows, cols] = sum of nparts partials + per-32-row column sums cs [ceil(rows / 32), cols]  (cols % 4 == 0)
hipError_t sum_parts_colsum_launch(const float *part, int nparts, int rows, int cols, float *dst, float *cs, hipStream_t s);
hipError_t reduce_partials_launch(const float *part, int nparts, long long n, float *dst, hipStream_t s);
// the same for `ninst` independent instances laid out back to back: part [ninst][nparts][n] -> dst [ninst][n]
hipError_t reduce_partials_batched_launch(const float *part, int ninst, int nparts, long long n, float *dst, hipStream_t s);

// Finish of the critics' skip heads when the hidden layers' parts were formed in the producing GEMMs' epilogues
// (head fusion): out[row, k*Q + q] = bias_k[q] + cat(s, a)[row] . Wh_k[q, 0 : L + A] + sum over planes of parts[plane][row][q].
// ONE launch for all 3C instances instead of a partial-sum reduction launch plus a head GEMM that streams cat(s, a)
// once per instance (198 MB at config 2): instances that share their state rows form a group - online and frozen
// critics read s_cur with the SAME weights (critic_frozen is the copy taken when the actor loss is formed), the targets
// read s_next - and a wave multiplies 16 state rows by all C*Q <= 16 head columns of the group at once
// (v_mfma_f32_16x16x4_f32, operands straight from global memory), so every state row is read once per group.
constexpr int HEAD_FINISH_MAX_SETS = 16;
struct HeadFinishGroup {
  const float *s;                            // state rows [M, lds], L columns
  int lds, nsets, nvar;                      // nsets weight sets (critics), nvar instances per set that differ in the action block
  const float *Wh[HEAD_FINISH_MAX_SETS];     // head weight of set k: row q at Wh[k][q*ldw + col], input columns first (L state, A action)
  const float *bias[HEAD_FINISH_MAX_SETS];
  int ldw;
  const float *a[2];                         // per variant: action rows [M, lda[v]], A columns
  int lda[2];
  float *out[2];                             // per variant: [M, ldo[v]], set k writes columns k*Q .. k*Q+Q-1
  int ldo[2];
  const float *parts[HEAD_FINISH_MAX_SETS][2];   // per (set, variant): [M][Q], the hidden layers' parts with the planes summed - or,
                                                 // HeadFinishArgs::sum_planes, plane 0 of the instance's `planes` planes [planes][M][Q]
};
struct HeadFinishArgs {
  int M, L, A, Q, planes, ngroups;
  int sum_planes;   // 1: the kernel adds the planes itself (in plane order): no reduction launch in front of it
  int plane_step;   // ... planes 0, plane_step, 2 plane_step, ... (`planes` of them); 0 = 1.  > 1: one pre-summed plane per hidden layer
  HeadFinishGroup g[2];
};
hipError_t head_finish_launch(const HeadFinishArgs &a, hipStream_t s);
// ---- GRU joiner (torch.nn.GRU cell, gate order r, z, n; encoder.py:40-42)
// start state of the scan: mode 0 zeros, 1 rows copied from src [B, L], 2 src [L] repeated over the batch
hipError_t gru_h0_launch(int mode, const float *src, float *h0, int B, int L, hipStream_t s);
// one time step over `rows` rows: gi, gh [rows, 3L] (input / recurrent pre-activations incl. biases), hprev, h [rows, L];
// hprev_save (optional): copy of hprev in the [N, L] table the W_hh weight gradient reads
// gh_parts: nparts K-split partial sums [nparts][rows, 3L] of W_hh h_prev (no bias) -> summed with b_hh into gh
// (kept for the backward pass); nparts == 0: gh already holds the complete pre-activation
hipError_t gru_cell_fwd_launch(const float *gi, float *gh, const float *gh_parts, int nparts, const float *b_hh,
                               const float *hprev, float *h, float *hprev_save, int rows, int L, hipStream_t s);
// backward of one step: dh = dstate (or 0) + carry_a + carry_b (or 0); gates recomputed from gi, gh;
// writes d gi, d gh [rows, 3L] and the direct part of d hprev (dh * z); the part through W_hh is a GEMM on d gh
// carry_b: nparts_b K-split partial sums [nparts_b][rows, L] of d gh_{t+1} W_hh
hipError_t gru_cell_bwd_launch(const float *dstate, const float *carry_a, const float *carry_b, int nparts_b,
                               const float *gi, const float *gh, const float *hprev, float *dgi, float *dgh,
                               float *dh_direct, int rows, int L, hipStream_t s);
// d encoder.hidden_state[l] = sum_b (a[b][l] + b[b][l])  (learned start state), fixed order
hipError_t gru_dh0_launch(const float *a, const float *b, int nparts_b, int B, int L, float *out, hipStream_t s);
// The whole scan as ONE persistent launch (gruscan.hip): a workgroup owns 4 (or 8) batch rows for all T steps, the recurrent
// weights streamed through an LDS-DMA ring from copies packed in stream order (gru_pack_launch, once per update).
struct GruScanArgs {
  int T, B, L, pad;
  const float *W, *bhh;      // W: the scan's packed weight stream (forward / backward copy); bhh: forward only
  const float *gi;           // [T*B, 3L]
  const float *h0;           // forward: start state [B, L]
  float *gh, *state, *hprev; // forward: outputs [T*B, 3L], [T*B, L], [T*B, L]; backward: gh / hprev are read
  const float *dstate;       // backward: [(T-1)*B, L]
  float *dgi, *dgh;          // backward: outputs [(T-1)*B, 3L]
  float *dh_init;            // backward: optional [B, L]: d h_{-1} per row (learned start state)
};
bool gru_scan_takes(int B, int L);
hipError_t gru_scan_fwd_launch(const GruScanArgs &a, hipStream_t s);
hipError_t gru_scan_bwd_launch(const GruScanArgs &a, hipStream_t s);
hipError_t gru_pack_launch(const float *Whh, int L, float *pack_fwd, float *pack_bwd, hipStream_t s);
// prep: optional - the work of prep_launch as extra workgroups of this launch (continuous policies; a launch of its own
// in front of the Gumbel kernel)
struct PrepArgs {
  const float *task_done, *episode_step;
  int T, B, burn_in, cumprod;
  float inv_global_batch;
  float *w, *contig;
  DevState *st;
  const float *log_alpha;
};
hipError_t policy_fwd_launch(const PolicyFwdArgs &a0, const PolicyFwdArgs &a1, int nprob, int M, int A,
                             const DevState *st, uint64_t seed, int discrete, hipStream_t s, const PrepArgs *prep = nullptr);
hipError_t onehot_launch(const float *action, int rows, int n, float *out, hipStream_t s);
// fin (optional): k_loss_finish's work rides in this launch as one extra workgroup (continuous policies)
hipError_t policy_bwd_launch(const float *logits, const float *noise, const float *action, const float *dpi_parts,
                             int nparts, float *dpi_sum, const float *w, const DevState *st, int M, int A,
                             float *dlogits, int discrete, hipStream_t s, const float *loss_partials = nullptr,
                             const LossFinishArgs *fin = nullptr);
// The last hidden layer's pre-activation gradient under a head of up to 32 outputs, gated by the forward launches' sign masks
// (GemmProblem::gm_*, wstat.hip) instead of the activations:  dpre[m][n] = mask(m, n) ? x : 0.01 x,  x = sum_q dY[m][q] Wh[q][n]
// (fma chain over q), 256 columns, per-64-row column sums; `ninst` instances of M rows in one launch (config 4: the 25-quantile
// critics' rank-25 product was a 0.25 ms tile launch that read h again; here 32 bytes of mask per row instead of 1 KiB).
constexpr int HDM_MAX_INST = 16, HDM_MAXQ = 32;
struct HeadDgradMaskedArgs {
  int M, Q, ninst, lddy, ldw;
  const float *dY[HDM_MAX_INST];      // [M, lddy]
  const float *Wh[HDM_MAX_INST];      // head rows over this layer's columns: Wh[q * ldw + n]
  const unsigned *gm[HDM_MAX_INST];   // [M / 32][4][64] sign masks of the layer's activations
  float *dpre[HDM_MAX_INST];          // [M, 256]
  float *colsum[HDM_MAX_INST];        // [M / 64, 256]
};
hipError_t head_dgrad_masked_launch(const HeadDgradMaskedArgs &a, hipStream_t s);
// ... and the pre-activation gradient of the actor's last hidden layer (256 wide, continuous policy, 2A <= 16) in the same launch:
// dpre = LeakyReLU'(h) * (d logits Wh), column sums per 64 rows
bool policy_bwd_dpre_takes(int discrete, int A, int hidden);
hipError_t policy_bwd_dpre_launch(const float *logits, const float *noise, const float *action, const float *dpi_parts, int nparts,
                                  float *dpi_sum, const float *w, const DevState *st, int M, int A, float *dlogits, const float *Wh,
                                  int ldw, const float *h, float *dpre, float *colsum, hipStream_t s,
                                  const float *loss_partials = nullptr, const LossFinishArgs *fin = nullptr);

inline int loss_blocks(int M, int G) { return (M + (256 / G) - 1) / (256 / G); }
bool loss_wave_form(int distributional, int Nq);   // kernels.hip: the wave-per-row TQC loss takes this shape (then G = 64)

}  // namespace fdql
