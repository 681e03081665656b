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
  // Segment 0 as the SUM of nsum arrays of the same shape (the per-network shares of an input gradient, agent.hip "dstate"): the
  // workgroup adds its 64 rows of every share in index order while it stages them, writes the sum to sum_out (= A[0]: the weight
  // gradients read it) and the rows' column sums to sum_colsum [M / 64, 256] - the launch that summed the shares (and wrote and
  // re-read the sum) goes away.  nsum == 0: A[0] is read as it is.
  const float *sum_parts;
  int nsum;
  long long sum_stride;              // floats between consecutive shares
  float *sum_out, *sum_colsum;
};

// Narrow-output dgrads of several networks in one launch (k_rowdot): out_p[M, A <= 16] = X_p[M, 256] W_p + D_p[M, Q <= 32] V_p with
// K-strided W_p (element (k, a) at W_p[k*ldw + a]) and V_p - the gradient of the policy's action through each frozen critic
// (d pi = dpre_0 W_0[:, action columns] + dz W_head[:, action columns]).  HBM-bound: X is read once, 16 rows per wave and tile.
constexpr int RDOT_MAX_PROB = 8;
struct RowDotArgs {
  int M, A, Q, nprob, wgs_per_prob;
  const float *X[RDOT_MAX_PROB];     // [M, 256], 16-byte aligned
  const float *W[RDOT_MAX_PROB];     // K-strided [256, ldw]
  const float *D[RDOT_MAX_PROB];     // [M, lddy] narrow segment (Q columns) or null
  const float *V[RDOT_MAX_PROB];     // K-strided [Q, ldv]
  float *out[RDOT_MAX_PROB];         // [M, ldo]
  int ldw, lddy, ldv, ldo;
};
bool rowdot_from_problems(const GemmProblem *probs, int nprob, RowDotArgs &args);
hipError_t rowdot_launch(const RowDotArgs &args, hipStream_t stream);
inline double rowdot_flops(const RowDotArgs &a) { return 2.0 * a.M * (double)a.A * (RD_K + a.Q) * a.nprob; }

// Does this problem have the kernel's form?  Fills args when it does.  FDQL_ROWDGRAD=0: never.
bool rowdgrad_from_problem(const GemmProblem &p, RowDgradArgs &args, int bm = RD_BM);   // bm < 64: only as a member of a chain launch
// Three dependent single-network dgrads of the same row blocks (64, 32 or 16 rows) in ONE launch (k_rowdgrad_chain<RT>): the summed input gradient
// x0 (RowDgradArgs::sum_*), then  x1 = gate1(x0 W1),  x2 = x0 W2a + x1 W2b,  x3 = gate3(x2 W3)  with x0, x1, x2 resident in LDS
// between the layers - the joiner's hidden-layer dgrad, d enc and the encoder's hidden-layer dgrad of a config-2-shaped plan
// (one hidden layer each).  Every xi is also written to memory (the weight gradients read them) with its per-block column sums.
struct RowChainArgs {
  int M, bm;   // bm: rows per workgroup, 64 (round 4), 32 or 16 (round 6: small batches); every column-sum array has M / bm rows
  const float *sum_parts; int nsum; long long sum_stride; float *x0, *cs0;
  const float *W1; int ldw1; const float *ref1; float *x1, *cs1;
  const float *W2a, *W2b; int ldw2a, ldw2b; float *x2, *cs2;
  const float *W3; int ldw3; const float *ref3; float *x3, *cs3;
};
// Do three consecutive launches (the first with a folded sum) form such a chain?  Fills args when they do.  FDQL_NO_ROWDGRAD_CHAIN: never.
bool rowchain_from_launches(const RowDgradArgs &l1, const RowDgradArgs &l2, const RowDgradArgs &l3, RowChainArgs &args, int bm = RD_BM);
hipError_t rowchain_launch(const RowChainArgs &args, hipStream_t stream);
inline double rowchain_flops(const RowChainArgs &a) { return 2.0 * a.M * (double)RD_N * RD_K * 4; }
bool rowdgrad_fold_sum(RowDgradArgs &args, const float *parts, int nsum, long long stride, float *sum_out, float *sum_colsum);   // FDQL_NO_DSTATE_SUM_FOLD: never
hipError_t rowdgrad_launch(const RowDgradArgs &args, hipStream_t stream);
inline double rowdgrad_flops(const RowDgradArgs &a) { return 2.0 * a.M * (double)RD_N * RD_K * a.nseg; }

}  // namespace fdql
