// Persistent, software-pipelined row-block GEMM (rowgemm.hip) for the critic ensemble's dense layers: a launch is a
// set of INSTANCES (critic k x {target, online, frozen}) of one Linear layer - same shapes, different pointers.
//   C[M, 256] = epi( A0[M, 256] W0 + A1[M, K1] W1 (+ A2[M, K2] W2) + bias )        K1, K2 <= 8 (e.g. the 6 action columns)
#pragma once
#include "common.h"

namespace fdql {

constexpr int RG_BM = 64;          // rows per tile
constexpr int RG_N = 256;          // output columns (4 waves x 64)
constexpr int RG_KMAIN = 256;      // K of the main segment
constexpr int RG_MAX_MINOR = 2;    // narrow K-segments beside it, one 8-k MFMA step each
constexpr int RG_MAX_INST = 24;    // the table travels in the kernel arguments (scalar loads): 24 x 144 B < 4 KiB

struct RowGemmInst {
  const float *A[1 + RG_MAX_MINOR];   // activations of the segments, [0] = main; row-major, K contiguous
  const float *W[1 + RG_MAX_MINOR];   // weights of the segments (first k of the segment)
  const float *bias;                  // [256] or null
  float *C, *C2;                      // outputs [M, ldc]; C2: second output of a dual launch
  const float *ref;                   // GRAD: activation output whose sign gates the gradient
  float *colsum;                      // GRAD: optional [M / 64, 256] column sums of the stored tile
  const float *hf_w;                  // head fusion (common.h, GemmProblem::hf_*): head weight rows over this layer's columns
  float *hf_out, *hf_out2;
  const float *fz_h, *fz_w;           // fused head dgrad (GemmProblem::fz_*)
  float *fz_out, *fz_colsum;
};

struct RowGemmArgs {
  int M;                     // rows per instance, multiple of 64
  int ninst, blocks_per_inst;
  int nminor, kminor[RG_MAX_MINOR];
  int lda[1 + RG_MAX_MINOR], ldw[1 + RG_MAX_MINOR];
  int ks;                    // 1: weights K-strided, element (k, n) at W[k*ldw + n] (dgrad); 0: K-contiguous W[n*ldw + k]
  int grad;                  // 1: x *= LeakyReLU'(ref) (no bias), column sums; 0: x = LeakyReLU(x + bias)
  int dual;                  // 1: C = f(all segments but the last), C2 = f(all)
  int hf_q, hf_ldw;          // head fusion: outputs per row (1, 2, 4, 8), 0 = off
  int fz, fz_ldw;            // fused head dgrad: A of the main segment formed from (fz_h, dY, fz_w)
  int ldc, ldc2, ldref;
  RowGemmInst inst[RG_MAX_INST];
};

// Can these problems (one launch group) run as one row-block launch?  Fills args / inst when they can.
bool rowgemm_from_problems(const GemmProblem *probs, int nprob, RowGemmArgs &args);
hipError_t rowgemm_launch(const RowGemmArgs &args, hipStream_t stream);
double rowgemm_flops(const RowGemmArgs &a);
int rowgemm_read_life(unsigned long long *out, int cap);   // diagnostic: (shader cycles, 100 MHz ticks) per workgroup of the last launch

}  // namespace fdql
