// Internal declarations of the agent's translation units (round 5: agent.hip split in four):
//   agent.hip        the C ABI of an agent's life: create / bind / update (the launch list's runner) / scalars / stats, and the
//                    environment switches (plan_switches_refresh)
//   agent_plan.hip   arena layout, workspace carve, the plan builder (build_plan) and its table upload (upload_tables)
//   agent_act.hip    act(): encoder -> joiner -> actor on a few rows
//   agent_debug.hip  the fdql_test_* / fdql_debug_* hooks
// plan_builder.h holds the builders the plan and the test hooks share.
#pragma once
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <functional>
#include <memory>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "chain.h"
#include "wstat.h"
#include "wgrad.h"
#include "rowdgrad.h"
#include "fwdchain.h"
#include "conv.h"
#include "common.h"
#include "update_kernels.h"

namespace fdql {

static inline int64_t pad4(int64_t n) { return (n + 3) / 4 * 4; }

struct TensorInfo {
  std::string name;
  int arena;       // 0 trainable, 1 targets, 2 frozen
  int64_t off;     // floats from the arena base
  int rows, cols;  // weight [rows, cols]; bias [rows, 0]; scalar [0, 0]
};

struct MlpDesc {
  int din = 0, dout = 0;
  std::vector<int> hid;
  std::vector<int64_t> w_off, b_off;  // offsets in the trainable arena
  int64_t hw_off = 0, hb_off = 0;
  int head_ld() const {
    int s = din;
    for (int h : hid) s += h;
    return s;
  }
  int in_of(int i) const { return i == 0 ? din : hid[i - 1]; }
};

struct SegIn {
  const float *ptr;
  int ld, width;
};

// One MLP evaluated on one set of rows (e.g. critic 3 of critic_target on the "next" rows).
struct MlpInst {
  const MlpDesc *d = nullptr;
  const float *wbase = nullptr;  // arena base the weights are read from (minus the arena's origin offset)
  int64_t worigin = 0;           // offset to subtract from the MlpDesc offsets for this arena
  std::vector<SegIn> in;
  int rows = 0;
  std::vector<float *> h;
  float *out = nullptr;
  int ldout = 0;
  std::vector<float *> dpre;
  std::vector<unsigned *> gm;    // per hidden layer: gate mask of h (GemmProblem::gm_*), [ceil(rows / 32)][4][64] dwords; empty: none
  std::vector<float *> dpre_cs;  // per hidden layer: [ceil(rows/64), hid] column sums of dpre (bias gradients)
  std::vector<int> dpre_cs_rows; // partial rows actually written there (0: one per 64 rows; the weight-stationary dgrad
                                 // launch writes one per workgroup of the instance, wstat.h)
  const float *W(int i) const { return wbase + (d->w_off[i] - worigin); }
  const float *Bv(int i) const { return wbase + (d->b_off[i] - worigin); }
  const float *HW() const { return wbase + (d->hw_off - worigin); }
  const float *HB() const { return wbase + (d->hb_off - worigin); }
};

// K-split of the per-step recurrent GEMMs of the GRU scan ([B, L] x [L, 3L] forward, [B, 3L] x [3L, L] backward):
// at B = 256 they are 16-48 workgroups walking K serially; the splits trade that for a partial sum in the gate kernel
constexpr int GRU_KSPLIT_FWD = 4, GRU_KSPLIT_BWD = 8;
// Few rows (temporal_len 2, small batches): d state as one problem is a handful of workgroups walking all 2(C+1)
// K-segments serially; below this row count each network's contribution is its own problem and a reduction sums them
constexpr int DSTATE_SPLIT_MAX_ROWS = 4096;   // (round 5: 1024 -> 4096 - one rank's share of a 256-window global batch, 1568 / 3136 rows: the one-problem
                                           // form was a 79 / 62 us launch at 29 / 40 TF, 100 - 200 tiles walking 12 K-segments each)
constexpr int ROWS_BM = 64;   // rows a row-block launch is counted in (the threshold FDQL_ROWGEMM / rows_min_tiles is in 64-row tiles)
constexpr size_t PLAN_CACHE_DEFAULT = 11;   // finished plans kept besides the current one (FDQL_PLAN_CACHE): e.g. 4 shards x a 3-buffer sample pool
// K-split of a conv weight gradient over its R = images * positions rows: ~4096 rows per workgroup, at most 1024 parts
inline int conv_wsplit(long long R) { return (int)std::max<long long>(1, std::min<long long>(1024, (R + 4095) / 4096)); }

enum StageKind { ST_GEMM, ST_SKINNY_WGRAD, ST_FUNC, ST_HEAD_DGRAD, ST_CHAIN, ST_WGRAD_STAT };

// one launch of a row-block kernel: the weight-stationary one (wstat.hip), the single-network dgrad forms (rowdgrad.hip)
struct RowsLaunch {
  bool ws = false;
  bool rd = false;      // single-network dgrad on 64-row blocks (rowdgrad.h)
  bool dot = false;     // narrow-output dgrads of several networks (rowdgrad.h, k_rowdot)
  bool chain3 = false;  // this launch and the two row-block dgrad launches behind it as one (rowdgrad.h, k_rowdgrad_chain)
  RowChainArgs rch;
  WsArgs wa;
  RowDgradArgs rda;
  RowDotArgs rdot;
  hipError_t launch(hipStream_t s) const {
    if (dot) return rowdot_launch(rdot, s);
    if (chain3) return rowchain_launch(rch, s);
    return rd ? rowdgrad_launch(rda, s) : wstat_launch(wa, s);
  }
};

struct GemmSub {
  std::vector<GemmProblem> probs;
  void *dev = nullptr;
  int blocks = 0;
};

struct Stage {
  StageKind kind;
  std::string name;
  std::vector<GemmProblem> gemm;
  GemmSub sub[GEMM_NSHAPES];  // the problems of `gemm`, grouped by tile shape (one launch each)
  int chain_bm = CH_BM;       // ST_CHAIN: rows per workgroup (64, or 32: chain.h)
  bool stream = false;        // ST_SKINNY_WGRAD: the streaming form (k_stream_wgrad: 256-wide X, one workgroup per slab)
  bool try_rows = false;      // ST_GEMM: groups of like problems may run on the weight-stationary row-block kernel (wstat.hip)
  std::vector<RowsLaunch> rows;    // the groups that do (one launch each); their problems are not in `sub`
  std::vector<SkinnyWgradProblem> swg;
  std::vector<HeadDgradProblem> hdg;
  WgArgs wga;                         // ST_WGRAD_STAT: the dense 256 x 256 weight-gradient blocks (wgrad.h), one launch
  std::vector<ChainProblem> cprobs;   // ST_CHAIN: programs (chain.h) and their operations
  std::vector<ChainOp> cops;
  void *cops_dev = nullptr;
  int lds_floats = 0;
  void *dev = nullptr;  // device copy of the table
  int blocks = 0;
  double flops = 0, bytes = 0;
  int phase = FDQL_PHASE_GRAD;
  int gpart = 1;  // FDQL_PHASE_GRAD stages of a bucketed plan: 0 = up to the critics' gradients (FDQL_PHASE_GRAD_CRITICS), 1 = the rest
  int when = 0;   // 0: whenever its phase runs; 1: only in a split (GRAD / APPLY) call; 2: only in a FDQL_PHASE_ALL call
  bool mfma = false;  // ST_FUNC: an MFMA kernel of its own (the implicit-GEMM convolutions, the small-block forward chain): its flops count as GEMM flops
  const char *prof = nullptr;   // ST_FUNC: prefix of the stage's name in fdql_agent_profile_update (default "k:" / "conv:")
  bool off = false;   // decided with the kernels (upload_tables): the stage has nothing left to do in this plan
  // head fusion (critics): 1 = a hidden layer's launch that leaves head partial sums, 2 = their plane sum, 3 = the head's finish.
  // When every layer runs weight-stationary, the kernels sum a tile's planes themselves (WsArgs::hf_presum): stage 2 is switched
  // off and stage 3 adds one plane per layer as it reads them (hfin_presum instead of hfin_plain).
  int hf_role = 0;
  // gate masks (GemmProblem::gm_*): 1 = a critics' forward layer whose weight-stationary launch writes them, 2 = a critics'
  // backward launch that may gate by them - when EVERY stage of role 1 runs weight-stationary (else nobody writes or reads them)
  int gm_role = 0;
  // the one problem of this stage runs on the row-block dgrad kernel with its first segment formed as a sum of shares
  // (RowDgradArgs::sum_*; build_plan has checked that the kernel takes it)
  int rd_max_blocks = 0;      // > 0: this stage's own limit for the row-block dgrad kernel (a member of a planned dgrad chain)
  int rd_chain_bm = 0;        // 32 / 16: planned as a member of a chain launch on blocks of that many rows (k_rowdgrad_chain<2> / <1>)
  bool needs_masks = false;   // reads gate masks: runs only when every critics' forward layer of the plan is weight-stationary (upload_tables)
  bool masks_fallback = false;   // ... and the GEMM stage that does the same work from h when they are not
  bool chained = false;   // switched off because the launch runs inside the chain launch of an earlier stage (RowsLaunch::chain3)
  bool fold_sum = false;
  const float *fold_parts = nullptr;
  int fold_n = 0;
  long long fold_stride = 0;
  float *fold_out = nullptr, *fold_cs = nullptr;
  std::shared_ptr<HeadFinishArgs> hfin;
  HeadFinishArgs hfin_plain, hfin_presum;
  bool hfin_can_presum = false;
  std::function<hipError_t(hipStream_t)> fn;
  bool runs_in(int call_phase) const {
    if (off) return false;
    if (call_phase == FDQL_PHASE_ALL) return when != 1;
    if (call_phase == FDQL_PHASE_GRAD_CRITICS || call_phase == FDQL_PHASE_GRAD_REST)
      return phase == FDQL_PHASE_GRAD && when != 2 && gpart == (call_phase == FDQL_PHASE_GRAD_REST ? 1 : 0);
    return phase == call_phase && when != 2;
  }
};

}  // namespace fdql

using namespace fdql;

struct fdql_agent {
  fdql_agent_config_t cfg;
  MlpDesc enc_obs, joiner, actor;
  std::vector<MlpDesc> critic;
  int64_t log_alpha_off = 0;
  // pixel encoder (cfg.img_c > 0): geometry and arena offsets of each conv layer; conv_feat = flattened output width
  // fast_*: the layer's forward / data gradient / weight gradient run on the implicit-GEMM kernels (conv.h), decided at create
  // from the geometry (the workspace has no column matrix for them); else im2col + grouped GEMM + col2im
  struct ConvLayer { ConvGeom g; int cout; int64_t w_off, b_off; bool fast_fwd = false, fast_dgrad = false, fast_wgrad = false; };
  std::vector<ConvLayer> conv;
  int conv_feat = 0;
  int hf_planes = 0;   // partial-sum planes per critic instance (head fusion)
  // GRU joiner (cfg.joiner_gru): offsets of weight_ih_l0 [3L,F], weight_hh_l0 [3L,L], bias_ih_l0, bias_hh_l0 [3L],
  // encoder.hidden_state [L] in the trainable arena
  int64_t gru_wih = 0, gru_whh = 0, gru_bih = 0, gru_bhh = 0, gru_h0 = 0;
  int64_t n_train = 0, tgt_begin = 0, tgt_end = 0, crit_begin = 0, crit_end = 0;
  std::vector<TensorInfo> tensors;

  // bound memory
  float *params = nullptr, *grads = nullptr, *adam_m = nullptr, *adam_v = nullptr, *targets = nullptr, *frozen = nullptr;
  char *ws = nullptr;
  int64_t ws_bytes = 0, ws_need = 0;
  bool bound = false;

  // geometry
  int T, B, N, M, A, L, Nq, Nt, nsplit;

  // workspace carve (offsets in bytes); filled by carve()
  int64_t carve_top = 0;
  std::map<std::string, std::pair<int64_t, int64_t>> named;  // name -> (byte offset, float count)

  // update / act / set_* / scalars on one handle are serialised (the facade's trainer thread runs train_step while the
  // Runner's agent thread calls act(): franQ/Agent/deepQlearning.py:83-94 vs :155-187)
  std::mutex mu;
  // plan: the launch list for one set of batch pointers.  A few finished plans are kept (keyed by their batch pointers)
  // so that a caller who alternates between two or three sample buffers does not rebuild and re-upload tables every step.
  fdql_batch_t batch = {};
  bool plan_ready = false;
  std::vector<Stage> stages;
  void *tables_dev = nullptr;
  // hipGraph of one FDQL_PHASE_ALL update of a plan: captured on `cap_stream` the second time the plan runs with the
  // same per-call values (seed, noise pointers), replayed on the caller's stream from then on.  Everything that changes
  // from step to step (optimiser step, Philox counter, lagged alpha) lives in device memory, so the node parameters
  // never change; other per-call values fall back to the eager launch list.
  struct PlanGraph {
    hipGraphExec_t exec = nullptr;
    uint64_t seed = 0;
    const float *noise_t = nullptr, *noise_a = nullptr;
    int eager_runs = 0;   // eager FDQL_PHASE_ALL runs of this plan with the key above
    void reset() {
      if (exec) (void)hipGraphExecDestroy(exec);
      exec = nullptr;
      eager_runs = 0;
    }
  };
  PlanGraph graph;
  hipStream_t cap_stream = nullptr;
  int use_graph = 0;    // FDQL_GRAPH (read at create): "1" replay, default eager launches
  long long graph_launches = 0;
  struct CachedPlan { fdql_batch_t batch; std::vector<Stage> stages; void *tables_dev; PlanGraph graph; };
  std::vector<CachedPlan> plan_cache;   // most recently stashed last; a hit moves a plan out (it becomes current): the front is the least recently used
  size_t plan_cache_max = PLAN_CACHE_DEFAULT;
  long long plans_built = 0;
  long long rows_min_tiles = 224;   // FDQL_ROWGEMM: "0" never, "all" always, a number = the threshold; default: groups of about one 64-row
                                    // tile per CU (a weight-stationary workgroup with 2 + 2 32-row tiles still beats the tile kernels:
                                    // config 4 at 128 windows per GPU; round 6: 256 -> 224 - config 2 at 32 windows, 10
                                    // layer-0 problems x 24 tiles = 240: critics.fwd0 0.0447 -> 0.0373 ms, critics.dpre0 0.0392 -> 0.0323,
                                    // step 0.346 -> 0.329 ms)
  int rowdgrad_min_blocks = 128;    // 64-row blocks a single-network dgrad needs for the row-block dgrad kernel (FDQL_ROWDGRAD_MIN_BLOCKS)
  int rowdot_min_rows = 2048;       // rows from which a stage of narrow-output dgrads runs on k_rowdot (config 2: 3136 rows 18.4 against 23.4 us on 128x32 tiles; 1568 rows 18.2 against 16.4 on the small-batch kernel)
  int rowdgrad_max_blocks = 256;    // ... and may have: one round of workgroups (config 4 at B = 1024, 784 blocks = 3.06 rounds: the tile
                                    // kernel's 3136 tiles are the better fit there: 0.138 against 0.153 ms for d enc)
  long long wgrad_stat_min_tiles = 1536;   // 32-row tiles (blocks x M / 32) from which the dense weight gradients run as one output-stationary
                                    //   launch (1 with FDQL_ROWGEMM=all, 6 x n with FDQL_ROWGEMM=n): 1536 = 8 tiles for each of 192 workgroups.
                                    //   A workgroup writes a 256 KB partial whatever its rows - config 2 at 64 windows (1470 tiles: 120
                                    //   workgroups of 12) 0.541 ms with the launch, 0.502 with the gradients riding in the dgrad launches; at
                                    //   128 windows (2940 tiles) 0.722 against 0.753; the 5-block launch of the two-bucket plan at 256
                                    //   windows (1960 tiles) 1.241 against 1.257
  bool wgrad_stat_pays(long long nblk) const { return nblk * (M / 32) >= wgrad_stat_min_tiles; }
  int small_max_tiles = 128;        // a GEMM stage of at most this many 64x64 tiles runs on the small-batch kernel (smallgemm.hip); 0: never
  // d state = sum over the online critics and the actor: one problem accumulating every network's K-segments, or - few rows
  // (a handful of workgroups would walk all segments serially), or many rows with critics the weight-stationary kernel
  // takes (wstat.h, plain dgrad form) - one problem per network into partials + a reduction
  bool dstate_split = false;
  // Data-parallel plans (world_size > 1) finish the critics' gradients - and log_alpha's: arena range [grad_bucket, n_train) -
  // right after the critics' backward (FDQL_PHASE_GRAD_CRITICS), so that their all-reduce runs beside the actor / encoder
  // backward (FDQL_PHASE_GRAD_REST); n_train: the plan is not bucketed
  int64_t grad_bucket = 0;
  bool no_buckets = false;   // FDQL_NO_BUCKETS, latched at create: the plan builder and fdql_agent_grad_bucket must agree
  bool force_buckets = false;
  bool bucketed() const { return (cfg.world_size > 1 || force_buckets) && !no_buckets; }
  // where the current step stands in the split-phase protocol (fdql_agent_update): a phase out of order is FDQL_ESTATE
  // instead of an optimiser step on a half-stale gradient arena
  enum StepState { STEP_NONE, STEP_CRITICS_DONE, STEP_GRAD_DONE };
  StepState step_state = STEP_NONE;
  const float *noise_t = nullptr, *noise_a = nullptr;  // per-call (read by the policy stage lambdas)
  uint64_t seed = 0;

  DevState *st() const { return reinterpret_cast<DevState *>(ws + named.at("dev_state").first); }
  float *buf(const std::string &n) const { return reinterpret_cast<float *>(ws + named.at(n).first); }
  bool has_buf(const std::string &n) const { return named.count(n) != 0; }
  float *alloc(const std::string &n, int64_t floats) {
    const int64_t bytes = (floats * 4 + 255) / 256 * 256;
    named[n] = {carve_top, floats};
    carve_top += bytes;
    return ws ? reinterpret_cast<float *>(ws + named[n].first) : nullptr;
  }
};


namespace fdql {
constexpr int STREAM_WGRAD_MAX_SLAB_ROWS = 640;   // K-split slabs up to this many rows: narrow weight gradients on the streaming launch
int layout(fdql_agent *a);            // agent_plan.hip: parameter arenas under the reference's state_dict names
void carve(fdql_agent *a);            // ... the workspace
int build_plan(fdql_agent *a);        // ... the launch list for a->batch (ends with upload_tables)
int upload_tables(fdql_agent *a);
}  // namespace fdql
