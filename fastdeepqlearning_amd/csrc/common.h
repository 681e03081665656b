// Internal declarations shared by the HIP translation units of libfdql_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string>
#include <vector>

#include "../../include/fdql.h"

namespace fdql {

void set_error(const char *fmt, ...);

#define FDQL_HIP(call)                                                                        \
  do {                                                                                        \
    hipError_t _e = (call);                                                                   \
    if (_e != hipSuccess) {                                                                   \
      fdql::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
      return FDQL_EHIP;                                                                       \
    }                                                                                         \
  } while (0)

#define FDQL_REQUIRE(cond, ...)      \
  do {                               \
    if (!(cond)) {                   \
      fdql::set_error(__VA_ARGS__);  \
      return FDQL_EINVAL;            \
    }                                \
  } while (0)

// Compute units the persistent one-workgroup-per-CU launches (wstat / wgrad / conv) may occupy: all of them.  (Round 4 measured
// leaving 16 / 32 CUs to a collective's channel kernels - FDQL_CU_RESERVE, profiles/r04_dp_overlap.txt: a net loss; the knob is gone.)
inline int cu_budget(int ncu) { return ncu; }

// ---------------------------------------------------------------------------------------
// Plan switches: every environment variable the library reads, in ONE place (agent.hip, plan_switches_refresh).  Read when an
// agent is created (and by the test hooks); each alternative they select is a parity case of tests/ - the second implementation
// a kernel family is checked against.  Defaults = what bench.py measures.
// ---------------------------------------------------------------------------------------
struct PlanSwitches {
  int chain = 1;                  // FDQL_CHAIN: 0 never / 1 encoder -> joiner -> actors as one k_chain launch when the batch fills the chip /
                                  //   2 ("enc") that chain whatever the size / 3 ("all") the critics as chain programs too
  long long rows_min_tiles = 224; // FDQL_ROWGEMM: "0" no row-block / stationary launches (1 << 60), "all" every eligible group whatever its
  bool rows_all = false;          //   size (1; also drops the size thresholds of k_wgrad_stat and k_rowdgrad), a number = the threshold
  bool rowdgrad = true;           // FDQL_ROWDGRAD=0: single-network dgrads on the tile kernel
  bool rowdgrad_chain = true;     // FDQL_NO_ROWDGRAD_CHAIN: the three dgrads behind d state as launches of their own
  bool wgrad_stat = true;         // FDQL_WGRAD_STAT=0: dense weight gradients ride in the dgrad launches (tile kernel K-split)
  bool wgrad_riders = true;       // FDQL_WGRAD_RIDERS=0: head / action-column gradients on their own launches
  int stream_wgrad = 1;           // FDQL_STREAM_WGRAD: 0 narrow gradients on the tile kernels / 1 streaming launch where slabs are short / 2 always
  bool small_gemm = true;         // FDQL_SMALL_GEMM=0: small stages on the tile kernels
  bool colsum_stream = true;      // FDQL_NO_COLSUM_STREAM: column sums in k_skinny_wgrad's launch
  bool gate_masks = true;         // FDQL_NO_GATE_MASKS
  bool head_dgrad_masked = true;  // FDQL_NO_HEAD_DGRAD_MASKED
  bool head_fuse = true;          // FDQL_NO_HEAD_FUSE
  bool head_presum = true;        // FDQL_NO_HEAD_PRESUM
  bool dual = true;               // FDQL_NO_DUAL
  bool fuse_dpre1 = true;         // FDQL_NO_FUSE_DPRE1
  bool policy_dpre_fuse = true;   // FDQL_NO_POLICY_DPRE_FUSE
  bool small_folds = true;        // FDQL_NO_SMALL_FOLDS: prep and the loss finish as launches of their own
  bool loss_wave = true;          // FDQL_LOSS_WAVE=0: thread-per-atom TQC loss
  bool gru_scan = true;           // FDQL_GRU_SCAN=0: step-by-step GRU launches
  bool act_fuse = true;           // FDQL_ACT_NO_FUSE: act() one launch per layer
  bool implicit_conv = true;      // FDQL_NO_IMPLICIT_CONV: conv layers on im2col + GEMM + col2im
  bool graph = false;             // FDQL_GRAPH=1: hipGraph replay of a plan
  bool no_buckets = false, force_buckets = false;   // FDQL_NO_BUCKETS / FDQL_FORCE_BUCKETS (data-parallel plans)
  int plan_cache = -1;            // FDQL_PLAN_CACHE: finished plans kept (-1: default)
};
const PlanSwitches &plan_switches();   // as of the last refresh
void plan_switches_refresh();          // re-reads the environment (fdql_agent_create, the fdql_test_* hooks)

// ---------------------------------------------------------------------------------------
// Grouped, K-segmented fp32 MFMA GEMM (gemm.hip)
//   C[M,N] (+split slabs) = epi( sum_seg A_seg[M,K_seg] * B_seg[K_seg,N] + bias )
// Operand element (r,k): kc=1 -> P[r*ld + k] (K contiguous), kc=0 -> P[k*ld + r].
// ---------------------------------------------------------------------------------------
constexpr int GEMM_MAX_SEG = 24;

struct GemmSeg {
  const float *A;
  const float *B;
  int lda, ldb;
  int K;
  int a_kc, b_kc;
};

enum GemmEpilogue { EPI_NONE = 0, EPI_LRELU = 1, EPI_LRELU_GRAD = 2, EPI_ADD_REF = 3 /* x += ref[row][col] */ };
enum GemmShape { GEMM_128x128 = 0, GEMM_128x32 = 1, GEMM_32x128 = 2, GEMM_64x128 = 3, GEMM_64x128_DUAL = 4, GEMM_64x64 = 5, GEMM_64x64_HF = 6,
                 GEMM_SMALL = 7 /* smallgemm.hip: 64x32 tiles, K split over the 16 waves of a workgroup (small batches) */, GEMM_NSHAPES = 8 };

struct GemmProblem {
  int M, N;
  int nseg;
  int ksplit;              // >1 only with nseg == 1: K range cut in ksplit slabs
  float *C;
  int ldc;
  long long split_stride;  // floats between consecutive K-split slabs of C
  const float *bias;       // [N] or null
  int epi;
  const float *ref;        // EPI_LRELU_GRAD: activation output whose sign gates the gradient
  int ldref;
  float *colsum;           // optional [ceil(M/64), N]: column sums of the stored values per 64-row block
  int emit_seg;            // -1, or a DUAL problem (64x128 tiles): C = f(sum over segments <= emit_seg),
  float *C2;               //   C2 = f(sum over ALL segments), same bias / activation: two outputs that share
  int ldc2;                //   their leading K-segments in one pass (ksplit == 1, no colsum)
  // Head fusion (shapes GEMM_64x64_HF and the dual one): besides storing the activation tile, its part of the skip
  // head's dot product is formed in the epilogue - hf_out[(plane*M + row)*hf_q + q] = sum over the wave's 32 columns
  // of x[row][col] * hf_w[q*hf_ldw + col], plane = tile_n*2 + (wave column) - so the head no longer re-reads the
  // hidden activations.  hf_q in {1, 2, 4, 8}; hf_w == null: off.  hf_out2: same for the second output of a dual problem.
  const float *hf_w;
  int hf_ldw, hf_q;
  float *hf_out, *hf_out2;
  // Fused head dgrad (weight-stationary row-block kernel, wstat.hip, only): the 256-wide segment's A operand is not read from memory
  // but formed while the tile is staged,  A[m][k] = LeakyReLU'(fz_h[m][k]) * sum_q dY[m][q] * fz_w[q*fz_ldw + k]  with dY
  // the problem's narrow segment (that is k_head_dgrad's formula: the gradient of the last hidden layer under a narrow
  // head), written to fz_out [M, 256] for the weight gradients, its per-64-row column sums to fz_colsum.
  const float *fz_h, *fz_w;
  int fz_ldw;
  int fz_discard;          // 1: nobody reads fz_out afterwards (frozen critics have no weight gradients): the weight-stationary kernel skips the store
  float *fz_out, *fz_colsum;
  // Gate masks (weight-stationary kernels, wstat.hip, only; round 4): a forward layer leaves the SIGN of its activations -
  // one bit per value, a dword per lane and 32-row tile in the kernel's own register layout - and the backward launches that gate
  // with LeakyReLU'(h) read those 32 bytes per row instead of the 1 KiB row (EPI_LRELU_GRAD's ref; fz_h).  gm_out / gm_out2:
  // written by a forward problem (second output of a dual one); gm_ref / gm_fz: stand in for ref / fz_h when the launch is told
  // that the producing launches wrote them (WsArgs::use_masks, decided with the kernels).  [M / 32][4][64] dwords each; null: none.
  unsigned *gm_out, *gm_out2;
  const unsigned *gm_ref, *gm_fz;
  int tiles_m, tiles_n;    // filled by gemm_finalize
  int tile_start;          // first block id of this problem in its launch
  GemmSeg seg[GEMM_MAX_SEG];
};

// Fills tiles_* / tile_start for a launch group; returns the total number of blocks.
int gemm_finalize(GemmProblem *probs, int nprob, int shape);
void gemm_set_dense_shape(int shape);
int gemm_dense_shape();                         // tile shape of the dense problems (FDQL_GEMM_DENSE_SHAPE)
int gemm_pick_shape(const GemmProblem &p, int dense_shape);
bool gemm_shape_is_dense(int shape);
double gemm_flops(const GemmProblem &p);
double gemm_bytes(const GemmProblem &p);
// probs_dev: device copy of the finalized group (all problems of one tile shape).
hipError_t gemm_launch(const GemmProblem *probs_dev, int nprob, int total_blocks, int shape, hipStream_t stream);
// Small-batch kernel (smallgemm.hip): does it take the problem's form (no K-split slabs, no second output / head fusion)?
bool gemm_small_takes(const GemmProblem &p);
hipError_t gemm_small_launch(const GemmProblem *probs_dev, int nprob, int total_blocks, hipStream_t stream);

// Which problem of a launch group a workgroup belongs to (problems sorted by their first block id): ONE vector load round -
// lane i looks at problem i, ballot, highest set bit - instead of a chain of up to nprob dependent loads; with a handful
// of workgroups per launch (temporal_len 2) that chain was a visible part of every multi-problem launch.
#if defined(__HIPCC__)
template <typename P, int P::*START>
__device__ __forceinline__ int find_problem(const P *probs, int nprob, int bid, int lane) {
  int pi = 0;
  for (int base = 0; base < nprob; base += 64) {
    const int i = base + lane;
    const bool ge = i < nprob && bid >= (((const __attribute__((address_space(1))) P *)probs)[i].*START);
    const unsigned long long m = __ballot(ge);
    if (m) pi = base + 63 - __builtin_clzll(m);
  }
  return __builtin_amdgcn_readfirstlane(pi);
}
#endif

// ---------------------------------------------------------------------------------------
// Column-sum / narrow reductions (kernels.hip): bias gradients
// ---------------------------------------------------------------------------------------
constexpr int SKINNY_MAX_OUT = 32;
constexpr int STREAM_WGRAD_MAX_OUT = 36;   // k_stream_wgrad over a 256-wide X: up to nine groups of four outputs

// dW(q, k) (slab s) = sum_{m in split s} dY[m, q] * X[m, k], written to dW[q*sq + k*sk].
// dY == null: dY = 1, Nout = 1 (column sums of X: bias gradients).
struct SkinnyWgradProblem {
  int M, Nout, K;
  const float *dY;
  int lddy;
  const float *X;
  int ldx;
  float *dW;               // slab 0 destination
  long long sq, sk;
  long long split_stride;
  int nsplit;
  int block_start, col_blocks;
};
int skinny_wgrad_finalize(SkinnyWgradProblem *p, int n);

// Gradient of the LAST hidden layer of an MLP whose head is narrow (dout <= HEAD_DGRAD_MAXQ, e.g. the
// Q quantile outputs of a critic):  dpre[m][n] = LeakyReLU'(h[m][n]) * sum_q dY[m][q] * Wh[q][n]
// - a rank-Q outer product, pure HBM streaming (read h, write dpre), not worth an MFMA tile -
// plus the per-64-row column sums the bias gradient is reduced from.
constexpr int HEAD_DGRAD_MAXQ = 8;
struct HeadDgradProblem {
  int M, N, Q;
  const float *dY;         // [M, lddy], columns 0..Q-1 used
  int lddy;
  const float *Wh;         // head weight rows q, already offset to this layer's columns: Wh[q*ldw + n]
  int ldw;
  const float *h;          // [M, N] activation output of the layer
  float *dpre;             // [M, N]
  float *colsum;           // [ceil(M/64), N]
  int block_start, col_blocks;
};
int head_dgrad_finalize(HeadDgradProblem *p, int n);

}  // namespace fdql
