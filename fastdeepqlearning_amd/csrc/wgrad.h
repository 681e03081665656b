// Output-stationary weight-gradient kernel (wgrad.hip): the dense 256 x 256 blocks of the weight gradients,
//   dW[n][k] = sum over the M rows of G[m][n] * X[m][k]         (G = d pre-activation [M, 256], X = the layer's input [M, 256])
// (franQ trains by autograd's addmm backward, mlp.py:88-94).  A launch is a set of such problems; every workgroup belongs to
// ONE problem for its whole life and keeps its 256 x 256 partial result in the AccVGPRs of its four waves (64 rows x 256
// columns = 256 registers per lane) while it walks its share of the rows in 32-row tiles - no per-tile epilogue at all.
// The partial of workgroup j of a problem goes to K-split slab j of the gradient arena (the slab sum that the tile
// kernels' K-split already needs adds them up in a fixed order).
#pragma once
#include "common.h"

namespace fdql {

constexpr int WG_BM = 32;          // rows per tile
constexpr int WG_N = 256;          // block size (both ways)
constexpr int WG_MAX_INST = 32;

struct WgInst {
  const float *G, *X;   // [M, 256] row-major, row pitch 256
  float *dW;            // slab 0 destination: element (n, k) at dW[n*ldw + k]
  int ldw, pad;
  // Riders: the few-column / few-row weight-gradient blocks that share an operand with this block (the action columns of a
  // critic's layer 0, the skip head's rows over this layer's input) ride in its k-steps as v_mfma_f32_4x4x1_16b_f32 on
  // the operand registers the block has loaded anyway, instead of re-reading G / X from HBM in a launch of their own.
  const float *X2;      // narrow-input rider: X2[M, nx2 <= 8] (row pitch ldx2) under the same G:
  float *dW2;           //   dW2[n][a] = sum_m G[m][n] X2[m][a]  at dW2[n*ldw2 + a]  (slab 0)
  const float *G2;      // narrow-output rider: G2[M, ng2 <= 4] (row pitch ldg2) over the same X:
  float *dW3;           //   dW3[q][k] = sum_m G2[m][q] X[m][k]  at dW3[q*ldw3 + k]  (slab 0)
  int nx2, ldx2, ldw2, ng2, ldg2, ldw3;
};

struct WgArgs {
  int M, ninst, blocks_per_inst, ncu;
  int nslab;                 // slabs of the gradient arena: slabs [per, nslab) of a block are cleared by its workgroups
  long long slab_stride;     // floats between consecutive slabs
  int wg_first[WG_MAX_INST + 1];
  WgInst inst[WG_MAX_INST];
};

// Does this K-split weight-gradient problem (Builder::wgrad_gemm) have the kernel's form?
bool wgrad_stat_takes(const GemmProblem &p);
// Fills args for a group of such problems (all with the same row count); false: too many / no workgroups to give.
bool wgrad_stat_from_problems(const GemmProblem *probs, int nprob, int nslab, long long slab_stride, WgArgs &args);
// Can this K-split weight-gradient problem ride with instance `inst` of args (same split, one operand shared)?  Attaches it
// (X2 or G2 slot) and returns true when it can; FDQL_WGRAD_RIDERS=0: never.  Call wgrad_stat_balance() after the last one.
bool wgrad_stat_add_rider(WgArgs &args, int inst, const GemmProblem &p);
// Deals the chip's workgroups to the blocks by cost (riders make a block's tiles dearer): fills wg_first.
bool wgrad_stat_balance(WgArgs &args);
hipError_t wgrad_stat_launch(const WgArgs &args, hipStream_t stream);
double wgrad_stat_flops(const WgArgs &a);

}  // namespace fdql
