// Row-block dgrad kernel (rowdgrad.hip) for the single-network input gradients of the backward pass - the joiner's and the
// encoder's hidden layers and the gradient of the encoder's output (franQ: autograd through mlp.py:88-94):
//   C[M, 256] = gate( sum over <= 2 segments of A_s[M, 256] W_s )           W_s K-strided: element (k, n) at W_s[k*ldw + n]
// with gate(x) = x * LeakyReLU'(ref) (or none) and per-64-row column sums of C (the bias gradients).  A workgroup owns 64 rows:
// the A tiles go to LDS once (LDS-DMA), the weights stream global -> registers as the B operand of v_mfma_f32_16x16x4_f32
// WITHOUT a transpose: a lane's 16-byte load of a weight row is four consecutive n of one k, i.e. the [4 k x 16 n] operand
// of four MFMAs whose outputs are the column sets {4 j + c}.
#pragma once
#include "common.h"

namespace fdql {

constexpr int RD_BM = 64, RD_N = 256, RD_K = 256, RD_MAX_SEG = 2;

struct RowDgradArgs {
  int M, nseg, gate;                 // gate: 1 = x *= (ref > 0 ? 1 : 0.01)
  const float *A[RD_MAX_SEG];        // [M, 256] row-major, 16-byte aligned
  const float *W[RD_MAX_SEG];        // K-strided [256 (k), ldw], 4-byte aligned
  int ldw[RD_MAX_SEG];
  const float *ref;                  // [M, 256] (gate)
  float *C;                          // [M, 256]
  float *colsum;                     // [M / 64, 256] or null
};

// Does this problem have the kernel's form?  Fills args when it does.  FDQL_ROWDGRAD=0: never.
bool rowdgrad_from_problem(const GemmProblem &p, RowDgradArgs &args);
hipError_t rowdgrad_launch(const RowDgradArgs &args, hipStream_t stream);
inline double rowdgrad_flops(const RowDgradArgs &a) { return 2.0 * a.M * (double)RD_N * RD_K * a.nseg; }

}  // namespace fdql
