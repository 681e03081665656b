// Row-block chain kernel (chain.hip): a workgroup owns 64 rows of a batch and runs a short PROGRAM of layer
// operations on them with the activations resident in LDS - several Linear layers of an MLP (or of several MLPs that
// feed each other) per launch, no activation round trip through HBM between them.
#pragma once
#include "common.h"

namespace fdql {

constexpr int CH_BM = 64;          // rows per workgroup (or 32: chain_launch's bm)
constexpr int CH_THREADS = 256;    // 4 waves, one per SIMD
constexpr int CH_MAX_SEG = 4;      // K-segments of one operation (torch.cat of up to 4 row blocks)
constexpr int CH_MAX_OPS = 16;      // operations per program (the program is copied to LDS)
constexpr int CH_LDS_FLOATS = 39616;   // dynamic LDS available to the images: 160 KiB minus the program copy and a reserve

enum ChainOpKind { CH_END = 0, CH_LOAD = 1, CH_GEMM = 2, CH_NARROW = 3 };
enum ChainFlags {
  CHF_ZERO = 1,      // GEMM: clear the accumulators first
  CHF_EMIT = 2,      // GEMM: run the epilogue (bias, activation, stores) after the K loop
  CHF_BEGIN = 8,     // NARROW: clear the head accumulators first
  CHF_FINISH = 16,   // NARROW: add the bias and store the wave's 16 rows
  CHF_HBEGIN = 32,   // GEMM with a head rider: clear the head accumulators first
};
enum ChainAct { CHA_NONE = 0, CHA_LRELU = 1 };

// Activation operand of a segment: LDS slot (float offset, row pitch); weights: global memory, K contiguous (element (n, k) at W[n*ldw + k])
struct ChainSeg {
  const float *W;
  int ldw;
  int slot, pitch;   // LDS image [bm][pitch] of the activation block, columns >= K zero up to the next multiple of 8
  int K;
};
// CH_LOAD: global rows -> LDS slot columns [col, col + width)
struct ChainLoadSeg {
  const float *src;
  int ld, width, col;
};

struct ChainOp {
  int kind, nseg, N, flags, act;
  int slot, pitch, kpad;        // LOAD: destination image; columns [sum of widths, kpad) are zeroed.  NARROW: `slot` = staging
                                // area of N * (roundup(max K, 16) + 4) floats for the head weights of one segment
  int out_slot, out_pitch;      // GEMM: LDS image that receives the output tile (-1: none)
  int ldo;
  int row_lo, row_hi, row_shift;   // global rows written / referenced: row_lo <= row < row_hi, at index row - row_shift
  const float *bias;
  float *out;                   // global output [*, ldo] or null
  const float *hw;              // GEMM (K-contiguous weights, N > 192): a narrow skip head's weight block over this op's
  int hldw, hN;                 // input columns rides in the K loop: head row n at hw[n*hldw + k], hN <= 32 outputs (CH_NARROW
                                // + CHF_FINISH later stores them); null: none
  ChainSeg seg[CH_MAX_SEG];
  ChainLoadSeg ld[CH_MAX_SEG];
};

struct ChainProblem {
  int rows;          // rows of the batch this program runs over (blocks of 64)
  int op_start;      // first op in the launch's op array
  int nops;          // operations of the program, the closing CH_END included
  int block_start;   // first workgroup id of this problem in its launch
  int lds_floats;    // LDS the program needs
};

inline int chain_pitch(int K) { return (K + 7) / 8 * 8 + 4; }   // (pitch / 4) odd: conflict-free ds_read_b128 across 32 rows
inline int chain_kpad(int K) { return (K + 7) / 8 * 8; }

// Fills block_start; returns the number of workgroups.  bm: rows per workgroup, 64 or 32 (images are [bm][pitch]).
int chain_finalize(ChainProblem *probs, int nprob, int bm);
hipError_t chain_launch(const ChainProblem *probs_dev, int nprob, const ChainOp *ops_dev, int total_blocks, int lds_floats, int bm,
                        hipStream_t stream);
double chain_op_flops(const ChainOp &op, int rows);
int chain_read_stamps(unsigned long long *out, int cap);   // diagnostic (chain_enable_stamps): s_memtime per operation of the last launch
void chain_enable_stamps(int on);

}  // namespace fdql
