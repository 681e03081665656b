// Small-block forward chain (fwdchain.hip, round 6): encoder MLP -> joiner MLP -> online actor + target actor of the reference's
// default architecture (franQ/Agent/components/encoder.py:52-67, models/mlp.py:88-94: every SkipHeadMLP with ONE hidden layer of
// 256, latent and encoder features 256) on 16- or 32-row blocks, two workgroups per CU, the weights streamed eight 16-k groups
// ahead.  k_chain (chain.hip) runs the same networks on 64- / 32-row blocks, one workgroup per CU with its fragments one group
// ahead: a block's eight dependent layers are then bound by the latency of their weight requests (0.10 ms at 100 blocks whatever
// the batch); here a layer is bound by its MFMAs.  One rank's share of a data-parallel batch and temporal_len 2 are the cases.
#pragma once
#include "common.h"

namespace fdql {

constexpr int F3_W = 256;          // width of every hidden layer, of the encoder's features and of the latent state
constexpr int F3_MAX_IN = 3;       // observation segments (torch.cat of obs, achieved goal, desired goal)
constexpr int F3_MAX_K0 = 384;     // their total width
constexpr int F3_MAX_P = 64;       // actor head outputs (2 A, or A logits)

struct Fwd3Mlp {                   // SkipHeadMLP, one hidden layer: h = LeakyReLU(W0 x + b0), out = Wh cat(x, h) + bh
  const float *W0, *b0;            // [256, din] (row pitch din), [256]
  const float *Wh, *bh;            // [dout, din + 256] (row pitch din + 256), [dout]
};

struct Fwd3Args {
  int N, M, B;                     // rows T B, rows with a gradient (T - 1) B, windows per time step
  int bm;                          // rows per workgroup: 16 or 32; N % bm == 0 and B % bm == 0
  int nin, K0;                     // observation segments and their total width
  const float *in[F3_MAX_IN];
  int in_ld[F3_MAX_IN], in_w[F3_MAX_IN];
  Fwd3Mlp enc, joi, act, act_t;    // act: rows [0, M) on the online weights; act_t: rows [B, N) on the target weights, stored at row - B
  int P;                           // actor head outputs
  float *enc_h, *enc_out, *joi_h, *state;   // [N, 256] each
  float *act_h;                    // [M, 256]
  float *act_out, *act_t_out;      // [M, P] each
};

bool fwd3_takes(const Fwd3Args &a);             // geometry the kernel has (alignment included)
hipError_t fwd3_launch(const Fwd3Args &a, hipStream_t stream);
inline double fwd3_flops(const Fwd3Args &a) {
  const double w = F3_W;
  return 2.0 * a.N * w * (a.K0 + (a.K0 + w) + w + 2 * w) + 2.0 * 2 * a.M * (w * w + 2 * w * a.P);
}

}  // namespace fdql
