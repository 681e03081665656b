// Host-side plan builder and runner of one SAC/TQC gradient step.
//
// fdql_agent_create() lays out the parameter arenas with the reference's state_dict names;
// the first fdql_agent_update() (or a change of batch pointers) builds a fixed list of
// launch stages — grouped GEMM problem tables, skinny-head tables, fused loss/policy
// kernels — uploads the tables once, and every later update just replays the launches.
// All per-step varying scalars (Adam step, bias corrections, lagged alpha) live in device
// memory (DevState), so the stage list is replayable without host-side changes.
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
#include "rowgemm.h"
#include "wstat.h"
#include "wgrad.h"
#include "rowdgrad.h"
#include "conv.h"
#include "common.h"
#include "update_kernels.h"

namespace fdql {

static thread_local char g_err[1024] = "";
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

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
constexpr int DSTATE_SPLIT_MAX_ROWS = 1024;
constexpr size_t PLAN_CACHE_DEFAULT = 11;   // finished plans kept besides the current one (FDQL_PLAN_CACHE): e.g. 4 shards x a 3-buffer sample pool
// K-split of a conv weight gradient over its R = images * positions rows: ~4096 rows per workgroup, at most 1024 parts
inline int conv_wsplit(long long R) { return (int)std::max<long long>(1, std::min<long long>(1024, (R + 4095) / 4096)); }

enum StageKind { ST_GEMM, ST_SKINNY_WGRAD, ST_FUNC, ST_HEAD_DGRAD, ST_CHAIN, ST_WGRAD_STAT };

// one launch of a row-block kernel: the weight-stationary one (wstat.hip, forward forms) or the streamed-weights one
struct RowsLaunch {
  bool ws = false;
  bool rd = false;      // single-network dgrad on 64-row blocks (rowdgrad.h)
  bool dot = false;     // narrow-output dgrads of several networks (rowdgrad.h, k_rowdot)
  bool chain3 = false;  // this launch and the two row-block dgrad launches behind it as one (rowdgrad.h, k_rowdgrad_chain)
  RowChainArgs rch;
  RowGemmArgs rg;
  WsArgs wa;
  RowDgradArgs rda;
  RowDotArgs rdot;
  hipError_t launch(hipStream_t s) const {
    if (dot) return rowdot_launch(rdot, s);
    if (chain3) return rowchain_launch(rch, s);
    return rd ? rowdgrad_launch(rda, s) : (ws ? wstat_launch(wa, s) : rowgemm_launch(rg, s));
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
  bool try_rows = false;      // ST_GEMM: groups of like problems may run on the persistent row-block kernel (rowgemm.hip)
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
  bool mfma = false;  // ST_FUNC: an MFMA kernel of its own (the implicit-GEMM convolutions): its flops count as GEMM flops
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
  long long rows_min_tiles = 256;   // FDQL_ROWGEMM: "0" never, "all" always, a number = the threshold; default: groups with at least one
                                    // 64-row tile per CU (a weight-stationary workgroup with 2 + 2 32-row tiles still beats the tile
                                    // kernels: config 4 at 128 windows per GPU)
  int rowdgrad_min_blocks = 128;    // 64-row blocks a single-network dgrad needs for the row-block dgrad kernel (FDQL_ROWDGRAD_MIN_BLOCKS)
  int rowdot_min_rows = 4096;       // rows from which a stage of narrow-output dgrads runs on k_rowdot
  int rowdgrad_max_blocks = 256;    // ... and may have: one round of workgroups (config 4 at B = 1024, 784 blocks = 3.06 rounds: the tile
                                    // kernel's 3136 tiles are the better fit there: 0.138 against 0.153 ms for d enc)
  int wgrad_stat_factor = 4;        // x rows_min_tiles 32-row tiles for the output-stationary weight-gradient launch (4 per workgroup)
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

namespace {

void add_mlp(fdql_agent *a, MlpDesc &m, const std::string &prefix, int din, const int32_t *hid, int nh, int dout,
             int64_t &top) {
  m.din = din;
  m.dout = dout;
  m.hid.assign(hid, hid + nh);
  for (int i = 0; i < nh; ++i) {
    const int in = m.in_of(i);
    m.w_off.push_back(top);
    a->tensors.push_back({prefix + ".feature_extractor." + std::to_string(i) + ".0.weight", 0, top, m.hid[i], in});
    top += pad4((int64_t)m.hid[i] * in);
    m.b_off.push_back(top);
    a->tensors.push_back({prefix + ".feature_extractor." + std::to_string(i) + ".0.bias", 0, top, m.hid[i], 0});
    top += pad4(m.hid[i]);
  }
  m.hw_off = top;
  a->tensors.push_back({prefix + ".head.weight", 0, top, dout, m.head_ld()});
  top += pad4((int64_t)dout * m.head_ld());
  m.hb_off = top;
  a->tensors.push_back({prefix + ".head.bias", 0, top, dout, 0});
  top += pad4(dout);
}

int layout(fdql_agent *a) {
  const fdql_agent_config_t &c = a->cfg;
  int64_t top = 0;
  a->conv.clear();
  a->conv_feat = 0;
  if (c.img_c > 0) {
    int ci = c.img_c, h = c.img_h, w = c.img_w;
    for (int i = 0; i < c.n_conv; ++i) {
      fdql_agent::ConvLayer L;
      L.g.C = ci; L.g.H = h; L.g.W = w; L.g.k = c.conv_k[i]; L.g.s = c.conv_s[i];
      L.g.OH = (h - L.g.k) / L.g.s + 1; L.g.OW = (w - L.g.k) / L.g.s + 1;
      L.cout = c.conv_out[i];
      const int K = ci * L.g.k * L.g.k;
      const std::string pre = "encoder.visible_layer_encoders.obs_2d.conv." + std::to_string(i);
      a->tensors.push_back({pre + ".weight", 0, top, L.cout, K});
      L.w_off = top; top += pad4((int64_t)L.cout * K);
      a->tensors.push_back({pre + ".bias", 0, top, L.cout, 0});
      L.b_off = top; top += pad4(L.cout);
      const bool u8_in = i == 0 && c.obs_2d_u8;
      if (i > 0 || u8_in) {   // (a float32 NCHW first layer has no implicit-GEMM kernel)
        L.fast_fwd = conv_fwd_takes(L.g, L.cout, u8_in);
        L.fast_wgrad = conv_wgrad_takes(L.g, L.cout, u8_in) && L.b_off == L.w_off + (int64_t)L.cout * K;
        L.fast_dgrad = i > 0 && conv_dgrad_takes(L.g, L.cout);
      }
      a->conv.push_back(L);
      ci = L.cout; h = L.g.OH; w = L.g.OW;
    }
    a->conv_feat = ci * h * w;
  }
  add_mlp(a, a->enc_obs, "encoder.visible_layer_encoders.obs_1d", c.obs_dim + 2 * c.goal_dim + a->conv_feat, c.enc_hidden,
          c.n_enc_hidden, c.enc_features, top);
  if (c.joiner_gru) {   // nn.GRU(hidden_features, latent, 1) + learnable start state (encoder.py:41-42)
    const int L3 = 3 * c.latent;
    a->joiner = MlpDesc();
    a->joiner.din = c.enc_features; a->joiner.dout = c.latent;
    a->tensors.push_back({"encoder.hidden_state", 0, top, c.latent, 0});
    a->gru_h0 = top; top += pad4(c.latent);
    a->tensors.push_back({"encoder.joiner.weight_ih_l0", 0, top, L3, c.enc_features});
    a->gru_wih = top; top += pad4((int64_t)L3 * c.enc_features);
    a->tensors.push_back({"encoder.joiner.weight_hh_l0", 0, top, L3, c.latent});
    a->gru_whh = top; top += pad4((int64_t)L3 * c.latent);
    a->tensors.push_back({"encoder.joiner.bias_ih_l0", 0, top, L3, 0});
    a->gru_bih = top; top += pad4(L3);
    a->tensors.push_back({"encoder.joiner.bias_hh_l0", 0, top, L3, 0});
    a->gru_bhh = top; top += pad4(L3);
  } else {
    add_mlp(a, a->joiner, "encoder.joiner", c.enc_features, c.joint_hidden, c.n_joint_hidden, c.latent, top);
  }
  a->tgt_begin = top;
  const int pi_out = c.discrete ? c.act_dim : 2 * c.act_dim;
  add_mlp(a, a->actor, "actor_critic.actor", c.latent, c.pi_hidden, c.n_pi_hidden, pi_out, top);
  a->crit_begin = top;
  a->critic.resize(c.n_critics);
  for (int k = 0; k < c.n_critics; ++k)
    add_mlp(a, a->critic[k], "actor_critic.critic.nets." + std::to_string(k), c.latent + c.act_dim, c.critic_hidden,
            c.n_critic_hidden, c.n_quantiles, top);
  a->crit_end = top;
  a->tgt_end = top;
  a->log_alpha_off = top;
  a->tensors.push_back({"actor_critic.log_alpha", 0, top, 0, 0});
  top += 4;
  a->n_train = top;
  // mirrored arenas
  const size_t n0 = a->tensors.size();
  for (size_t i = 0; i < n0; ++i) {
    const TensorInfo &t = a->tensors[i];
    if (t.off >= a->tgt_begin && t.off < a->tgt_end) {
      TensorInfo u = t;
      u.arena = 1;
      u.off = t.off - a->tgt_begin;
      size_t p;
      if ((p = u.name.find(".actor.")) != std::string::npos) u.name.replace(p, 7, ".actor_target.");
      else if ((p = u.name.find(".critic.")) != std::string::npos) u.name.replace(p, 8, ".critic_target.");
      a->tensors.push_back(u);
    }
  }
  for (size_t i = 0; i < n0; ++i) {
    const TensorInfo &t = a->tensors[i];
    if (t.off >= a->crit_begin && t.off < a->crit_end) {
      TensorInfo u = t;
      u.arena = 2;
      u.off = t.off - a->crit_begin;
      const size_t p = u.name.find(".critic.");
      u.name.replace(p, 8, ".critic_frozen.");
      a->tensors.push_back(u);
    }
  }
  return 0;
}

// ------------------------------------------------------------------------------ workspace
void carve(fdql_agent *a) {
  a->carve_top = 0;
  a->named.clear();
  const fdql_agent_config_t &c = a->cfg;
  const int64_t N = a->N, M = a->M, Nq = a->Nq;
  a->alloc("dev_state", 64);
  a->alloc("scalars", 16);
  a->alloc("w", M);
  a->alloc("is_contiguous", M);
  auto mlp_bufs = [&](const std::string &p, const MlpDesc &d, int64_t rows, bool bwd, bool out) {
    for (size_t i = 0; i < d.hid.size(); ++i) {
      a->alloc(p + ".h" + std::to_string(i), rows * d.hid[i]);
      if (bwd) {
        a->alloc(p + ".dpre" + std::to_string(i), M * d.hid[i]);
        a->alloc(p + ".cs" + std::to_string(i), ((M + 63) / 64) * d.hid[i]);
      }
    }
    if (out) a->alloc(p + ".out", rows * d.dout);
  };
  for (size_t i = 0; i < a->conv.size(); ++i) {   // im2col matrix, NHWC output, and their gradients over the M images
    const fdql_agent::ConvLayer &L = a->conv[i];
    const int64_t pos = (int64_t)L.g.OH * L.g.OW, K = (int64_t)L.g.C * L.g.k * L.g.k;
    const std::string p = "conv" + std::to_string(i);
    if (!L.fast_fwd || !L.fast_wgrad) a->alloc(p + ".col", N * pos * K);   // (the implicit-GEMM kernels have no column matrix)
    a->alloc(p + ".out", N * pos * L.cout);
    a->alloc(p + ".dpre", M * pos * L.cout);
    if (i > 0 && !L.fast_dgrad) a->alloc(p + ".dcol", M * pos * K);
    if (L.fast_wgrad) {   // one (dW, db) partial per slab of the output-stationary launch, then one reduction into slab 0
      a->alloc(p + ".wpart", (int64_t)conv_wgrad_slabs(L.g, L.cout, i == 0, M) * ((int64_t)L.cout * K + L.cout));
    } else {
      // weight gradient: K-split of its own over the M*pos rows (far more rows than the slab count serves), then a
      // reduction of the partials into slab 0; bias gradient: two-level column sum
      a->alloc(p + ".wpart", (int64_t)conv_wsplit(M * pos) * L.cout * K);
      a->alloc(p + ".bpart", (int64_t)(colsum_tall_blocks(M * pos) + colsum_tall_blocks(colsum_tall_blocks(M * pos))) * L.cout);
    }
  }
  mlp_bufs("enc_obs", a->enc_obs, N, true, true);
  if (c.joiner_gru) {
    const int64_t L3 = 3 * c.latent, Bw = a->B;
    a->alloc("gru.gi", N * L3);          // W_ih e + b_ih for every row
    a->alloc("gru.gh", N * L3);          // W_hh h_{t-1} + b_hh, step by step
    a->alloc("gru.hprev", N * c.latent); // h_{t-1} per row (start state for t = 0)
    a->alloc("gru.h0", Bw * c.latent);
    a->alloc("gru.dgi", M * L3);
    a->alloc("gru.dgh", M * L3);
    a->alloc("gru.dhz0", Bw * c.latent); // direct part of d h_{t-1} (dh * z), double-buffered over t
    a->alloc("gru.dhz1", Bw * c.latent);
    a->alloc("gru.dhw", GRU_KSPLIT_BWD * Bw * c.latent);  // part of d h_{t-1} through W_hh, K-split partials
    a->alloc("gru.ghp", GRU_KSPLIT_FWD * Bw * L3);        // K-split partials of W_hh h_{t-1} for the current step
    a->alloc("gru.wpack_f", (int64_t)c.latent * L3);      // persistent scans (gruscan.hip): W_hh packed in the forward scan's stream order
    a->alloc("gru.wpack_b", (int64_t)c.latent * L3);      // ... and W_hh^T in the backward scan's
    a->alloc("gru.dh_init", Bw * c.latent);               // ... and d h_{-1} per row (learned start state)
  } else {
    mlp_bufs("joiner", a->joiner, N, true, false);
  }
  a->alloc("state", N * c.latent);
  mlp_bufs("actor_t", a->actor, M, false, true);
  mlp_bufs("actor", a->actor, M, true, true);
  a->alloc("next_action", M * c.act_dim);
  a->alloc("next_log_pi", M);
  a->alloc("pi", M * c.act_dim);
  a->alloc("log_pi", M);
  a->alloc("noise_actor", M * c.act_dim);
  a->alloc("pi_diff", M * c.act_dim);
  if (c.discrete) a->alloc("action_onehot", N * c.act_dim);
  for (int k = 0; k < c.n_critics; ++k) {
    const std::string s = std::to_string(k);
    mlp_bufs("crit_t" + s, a->critic[k], M, false, false);
    mlp_bufs("crit" + s, a->critic[k], M, true, false);
    mlp_bufs("crit_f" + s, a->critic[k], M, true, false);
    for (size_t i = 0; i < a->critic[k].hid.size(); ++i)   // gate masks of the online and the frozen pass (wstat.hip; 32 bytes per row)
      for (const char *pre : {"crit", "crit_f"}) a->alloc(pre + s + ".gm" + std::to_string(i), (int64_t)((M + 31) / 32) * 256);
  }
  {   // head fusion: per critic instance (3C of them) the partial head sums of every hidden layer, then their total
    int planes = 0;
    for (int h : a->critic[0].hid) planes += ((h + 63) / 64) * 2;
    a->hf_planes = planes;
    const int q = c.n_quantiles;
    const bool can_fuse = planes > 0 && (q == 1 || q == 2 || q == 4 || q == 8);   // same rule as the plan below
    a->alloc("hf.parts", can_fuse ? (int64_t)3 * c.n_critics * planes * M * q : 1);
    a->alloc("hf.sum", can_fuse ? (int64_t)3 * c.n_critics * M * q : 1);
  }
  a->alloc("next_z", M * Nq);
  a->alloc("q_pred", M * Nq);
  a->alloc("q_frozen", M * Nq);
  a->alloc("td_target", M * (a->Nt > 0 ? a->Nt : 1));
  a->alloc("dz", M * Nq);
  a->alloc("dzf", M * Nq);
  a->alloc("q_loss", M);
  a->alloc("pi_loss", M);
  a->alloc("alpha_loss", M);
  a->alloc("dpi", M * c.act_dim);
  a->alloc("dlogits", M * a->actor.dout);
  a->alloc("dstate", M * c.latent);
  if (a->dstate_split) a->alloc("dstate.parts", (int64_t)(c.n_critics + 1) * M * c.latent);
  a->alloc("denc", M * c.enc_features);
  a->alloc("cs.dstate", ((M + 31) / 32) * c.latent);   // per 64 rows (one-problem d state) or per 32 rows (sum of shares)
  a->alloc("cs.denc", ((M + 63) / 64) * c.enc_features);
  a->alloc("dpi_part", (int64_t)c.n_critics * M * c.act_dim);
  a->alloc("loss_partials", (int64_t)loss_blocks((int)M, 256) * LOSS_NPART + (int64_t)M * LOSS_NPART + LOSS_NPART);
  a->alloc("slabs", (int64_t)a->nsplit * a->n_train);
  a->alloc("loss_fin_args", 32);   // LossFinishArgs (update_kernels.h): k_loss's fused finish
  {   // fdql_agent_summaries: 4 scalars + one norm per trainable tensor, then the int64 (offset, count) table (8-byte aligned)
    int64_t nt = 0;
    for (const TensorInfo &t : a->tensors) nt += t.arena == 0;
    a->alloc("summaries", ((4 + nt + 1) & ~(int64_t)1) + 4 * nt + 4);
  }
}

// --------------------------------------------------------------------------- plan builder
constexpr int STREAM_WGRAD_MAX_SLAB_ROWS = 640;

struct Builder {
  fdql_agent *a;
  std::vector<Stage> &st;
  Builder(fdql_agent *ag) : a(ag), st(ag->stages) {}
  // dense 256 x 256 weight-gradient blocks: candidates for the output-stationary launch (wgrad.h), each with the index of the
  // stage it rides in otherwise (-1: the tail stage)
  std::vector<std::pair<GemmProblem, int>> wg_cand;
  // The candidates gathered so far become ONE output-stationary launch (stage `name`, appended) when there are enough row
  // tiles for every workgroup to amortise its 256 KiB partial result; else they ride in their host stages / `fallback`.
  void flush_wgrad_stat(const std::string &name, Stage &fallback) {
    if (wg_cand.empty()) return;
    std::vector<GemmProblem> probs;
    for (auto &pc : wg_cand) probs.push_back(pc.first);
    Stage wst;
    wst.kind = ST_WGRAD_STAT; wst.name = name;
    const long long tiles = (long long)probs.size() * (probs[0].seg[0].K / WG_BM);
    if (tiles >= a->wgrad_stat_factor * a->rows_min_tiles && wgrad_stat_from_problems(probs.data(), (int)probs.size(), a->nsplit, a->n_train, wst.wga)) {
      // the few-column / few-row gradients that share an operand with one of the blocks (a critic's action columns, its
      // skip head's rows over the state and over h0) ride with it instead of re-reading the operand in the tail launches
      for (size_t i = 0; i < fallback.gemm.size();) {
        bool taken = false;
        for (int k = 0; k < wst.wga.ninst && !taken; ++k) taken = wgrad_stat_add_rider(wst.wga, k, fallback.gemm[i]);
        if (taken) fallback.gemm.erase(fallback.gemm.begin() + i);
        else ++i;
      }
      wgrad_stat_balance(wst.wga);
      wst.flops = wgrad_stat_flops(wst.wga);
      wst.bytes = 8.0 * wst.wga.M * WG_N * wst.wga.ninst;
      st.push_back(wst);
    } else {
      for (auto &pc : wg_cand) (pc.second >= 0 ? st[pc.second] : fallback).gemm.push_back(pc.first);
    }
    wg_cand.clear();
  }

  Stage &gemm_stage(const std::string &name) {
    st.emplace_back();
    st.back().kind = ST_GEMM;
    st.back().name = name;
    return st.back();
  }
  Stage &func_stage(const std::string &name, std::function<hipError_t(hipStream_t)> fn, int phase = FDQL_PHASE_GRAD) {
    st.emplace_back();
    st.back().kind = ST_FUNC;
    st.back().name = name;
    st.back().fn = std::move(fn);
    st.back().phase = phase;
    return st.back();
  }

  static GemmProblem new_gemm(int M, int N, float *C, int ldc) {
    GemmProblem p;
    memset(&p, 0, sizeof(p));
    p.M = M; p.N = N; p.C = C; p.ldc = ldc; p.ksplit = 1; p.emit_seg = -1;
    return p;
  }
  static void add_seg(GemmProblem &p, const float *A, int lda, int a_kc, const float *B, int ldb, int b_kc, int K) {
    if (K <= 0) return;
    GemmSeg &s = p.seg[p.nseg++];
    s.A = A; s.lda = lda; s.a_kc = a_kc; s.B = B; s.ldb = ldb; s.b_kc = b_kc; s.K = K;
  }

  // forward hidden layer i of inst -> problem
  GemmProblem fwd_layer(const MlpInst &m, int i) {
    const MlpDesc &d = *m.d;
    GemmProblem p = new_gemm(m.rows, d.hid[i], m.h[i], d.hid[i]);
    if (i == 0) {
      int col = 0;
      for (const SegIn &s : m.in) {
        add_seg(p, s.ptr, s.ld, 1, m.W(0) + col, d.din, 1, s.width);
        col += s.width;
      }
    } else {
      add_seg(p, m.h[i - 1], d.hid[i - 1], 1, m.W(i), d.hid[i - 1], 1, d.hid[i - 1]);
    }
    p.bias = m.Bv(i);
    p.epi = EPI_LRELU;
    if ((size_t)i < m.gm.size()) p.gm_out = m.gm[i];
    return p;
  }
  // head: out = W_head cat(in, h_0..h_{n-1}) + b   (mlp.py:93-94); narrow heads get the 128x32 tile
  GemmProblem fwd_head(const MlpInst &m) {
    const MlpDesc &d = *m.d;
    const int ld = d.head_ld();
    GemmProblem p = new_gemm(m.rows, d.dout, m.out, m.ldout);
    int col = 0;
    for (const SegIn &s : m.in) { add_seg(p, s.ptr, s.ld, 1, m.HW() + col, ld, 1, s.width); col += s.width; }
    for (size_t i = 0; i < d.hid.size(); ++i) { add_seg(p, m.h[i], d.hid[i], 1, m.HW() + col, ld, 1, d.hid[i]); col += d.hid[i]; }
    p.bias = m.HB();
    return p;
  }
  // the head restricted to the MLP's inputs (the hidden activations' part comes from the fused partial sums)
  GemmProblem fwd_head_inputs_only(const MlpInst &m) {
    const MlpDesc &d = *m.d;
    const int ld = d.head_ld();
    GemmProblem p = new_gemm(m.rows, d.dout, m.out, m.ldout);
    int col = 0;
    for (const SegIn &s : m.in) { add_seg(p, s.ptr, s.ld, 1, m.HW() + col, ld, 1, s.width); col += s.width; }
    p.bias = m.HB();
    return p;
  }
  int head_col_of_hidden(const MlpDesc &d, int i) const {
    int col = d.din;
    for (int j = 0; j < i; ++j) col += d.hid[j];
    return col;
  }
  // dpre_i = (dY Wh[:, cols(h_i)] + dpre_{i+1} W_{i+1}) * lrelu'(h_i)
  GemmProblem bwd_dpre(const MlpInst &m, int i, const float *dY, int lddy) {
    const MlpDesc &d = *m.d;
    GemmProblem p = new_gemm(m.rows, d.hid[i], m.dpre[i], d.hid[i]);
    add_seg(p, dY, lddy, 1, m.HW() + head_col_of_hidden(d, i), d.head_ld(), 0, d.dout);
    if (i + 1 < (int)d.hid.size()) add_seg(p, m.dpre[i + 1], d.hid[i + 1], 1, m.W(i + 1), d.hid[i], 0, d.hid[i + 1]);
    p.epi = EPI_LRELU_GRAD;
    p.ref = m.h[i];
    p.ldref = d.hid[i];
    p.colsum = m.dpre_cs[i];
    if ((size_t)i < m.gm.size()) p.gm_ref = m.gm[i];
    return p;
  }
  // last hidden layer under a narrow head: the rank-dout outer product as a streaming kernel
  static bool narrow_head_last(const MlpDesc &d, int i) {
    return i + 1 == (int)d.hid.size() && d.dout <= HEAD_DGRAD_MAXQ;
  }
  HeadDgradProblem bwd_dpre_head(const MlpInst &m, int i, const float *dY, int lddy) {
    const MlpDesc &d = *m.d;
    HeadDgradProblem p;
    memset(&p, 0, sizeof(p));
    p.M = m.rows; p.N = d.hid[i]; p.Q = d.dout;
    p.dY = dY; p.lddy = lddy;
    p.Wh = m.HW() + head_col_of_hidden(d, i); p.ldw = d.head_ld();
    p.h = m.h[i]; p.dpre = m.dpre[i]; p.colsum = m.dpre_cs[i];
    return p;
  }
  // K-segments of d(input columns [col, col+width)) = dY Wh[:, cols] + dpre_0 W_0[:, cols]
  void input_grad_segs(const MlpInst &m, const float *dY, int lddy, int col, GemmProblem &p) {
    const MlpDesc &d = *m.d;
    add_seg(p, dY, lddy, 1, m.HW() + col, d.head_ld(), 0, d.dout);
    if (!d.hid.empty()) add_seg(p, m.dpre[0], d.hid[0], 1, m.W(0) + col, d.din, 0, d.hid[0]);
  }
  // dW[nout, width] (slabs) = dOut[R, nout]^T X[R, width], K-split over the R rows
  void wgrad_gemm(int R, const float *dOut, int ldo, int nout, const float *X, int ldx, int width, float *dst, int ldw,
                  Stage &gs, Stage &narrow) {
    if (width <= 0) return;
    GemmProblem p = new_gemm(nout, width, dst, ldw);
    add_seg(p, dOut, ldo, 0, X, ldx, 0, R);
    p.ksplit = a->nsplit;
    p.split_stride = a->n_train;
    if (wgrad_stat_takes(p)) {   // decided once every weight gradient of the plan is known (build_plan)
      int host = -1;
      for (size_t i = 0; i < st.size(); ++i)
        if (&st[i] == &gs) host = (int)i;
      wg_cand.push_back({p, host});
      return;
    }
    // narrow problems (other tile shapes = other launches) are pooled in one stage at the end
    // (33..36 outputs over a 256-wide input - the 2 x 17 logits of config 4's actor head - pick a square tile shape, 20 TF for an
    // HBM-bound product: they go with the narrow ones, where the streaming launch takes them)
    const bool streams = nout > 32 && nout <= STREAM_WGRAD_MAX_OUT && width == 256 && ldx == 256;
    (gemm_shape_is_dense(gemm_pick_shape(p, gemm_dense_shape())) && !streams ? gs : narrow).gemm.push_back(p);
  }
  // bias gradient = column sums of dOut over R rows, from per-64-row partials `cs` when the dgrad GEMM left them
  void wgrad_bias(int R, const float *dOut, int ldo, int nout, const float *cs, float *dst, Stage &ws, int cs_rows = 0) {
    SkinnyWgradProblem p;
    memset(&p, 0, sizeof(p));
    p.Nout = 1; p.K = nout; p.dY = nullptr;
    if (cs) { p.M = cs_rows > 0 ? cs_rows : (R + 63) / 64; p.X = cs; p.ldx = nout; }
    else { p.M = R; p.X = dOut; p.ldx = ldo; }
    p.dW = dst; p.sq = 0; p.sk = 1; p.split_stride = a->n_train; p.nsplit = a->nsplit;
    ws.swg.push_back(p);
  }
  // The narrow weight gradients left in `from` after the riders were dealt (flush_wgrad_stat) that have the streaming
  // form - a few outputs over a 256-wide input, or a few input columns under a 256-wide output gradient (roles swapped) -
  // move to the one-launch streaming stage `to` (kernels.hip, k_stream_wgrad); FDQL_STREAM_WGRAD=0: none.
  // The ones that are narrow BOTH ways (a skip head's rows over the action columns: 2 x 6) join the column sums' launch
  // (`skinny`, k_skinny_wgrad: any K) - left on the tile kernel they were a 40 us launch of their own for 2 MB.
  void take_stream_wgrads(Stage &from, Stage &to, Stage &skinny) {
    const char *env = getenv("FDQL_STREAM_WGRAD");
    if (env && env[0] == '0') return;
    // few rows (temporal_len 2): the narrow gradients stay GEMM problems of the tail stage, which the small-batch kernel takes
    // in its one launch - a streaming launch of their own is 15 us of a 0.28 ms step (0.277 -> 0.264 ms)
    if (!(env && env[0] == '2') && a->small_max_tiles > 0 && a->M <= 1024 && gemm_dense_shape() == GEMM_64x64 &&
        gemm_variant() == GEMM_DEFAULT_VARIANT)
      return;
    // Short K-split slabs (a few hundred rows, config 2: the tile kernels' narrow launches are one memory round trip per K
    // iteration there, 0.7-2.4 TB/s): everything that has the form.  Long slabs (config 4 at B = 1024: 1568 rows): the
    // 32x128 tile streams the head rows at 5 TB/s - better than the 4.4 TB/s here - but the 128x32 launch of the few-input-
    // column gradients does 1.8 TB/s: only those move, and the tiny ones stay where they are (the column sums' launch
    // would take them one column per thread).  FDQL_STREAM_WGRAD=2: everything, whatever the slab length (tests).
    const bool all = (env && env[0] == '2') || a->M / a->nsplit <= STREAM_WGRAD_MAX_SLAB_ROWS;
    for (size_t i = 0; i < from.gemm.size();) {
      const GemmProblem &p = from.gemm[i];
      SkinnyWgradProblem q;
      memset(&q, 0, sizeof(q));
      bool ok = p.nseg == 1 && p.ksplit == a->nsplit && p.split_stride == a->n_train && !p.bias && p.epi == EPI_NONE && !p.colsum && !p.C2 &&
                !p.seg[0].a_kc && !p.seg[0].b_kc;
      if (ok && p.M <= SKINNY_MAX_OUT && p.N <= 64) {   // narrow both ways
        const GemmSeg &sg = p.seg[0];
        q.M = sg.K; q.Nout = p.M; q.K = p.N; q.dY = sg.A; q.lddy = sg.lda; q.X = sg.B; q.ldx = sg.ldb;
        q.dW = p.C; q.sq = p.ldc; q.sk = 1; q.split_stride = p.split_stride; q.nsplit = p.ksplit;
        const bool tiny = q.Nout <= 2 && q.K <= 8;   // (k_skinny_wgrad's threads-over-rows path: 0.001 ms beside the column sums)
        if (tiny && !all) { ++i; continue; }
        if (!tiny && !stream_wgrad_takes(q)) { ++i; continue; }
        (tiny ? skinny : to).swg.push_back(q);
        from.gemm.erase(from.gemm.begin() + i);
        continue;
      }
      if (ok) {
        const GemmSeg &sg = p.seg[0];   // dW[nout = p.M][width = p.N] = dOut[R, nout]^T X[R, width]
        q.M = sg.K; q.K = 256; q.ldx = 256; q.dW = p.C; q.split_stride = p.split_stride; q.nsplit = p.ksplit;
        // (33..36 outputs - config 4's 2 x 17 logits - are two row tiles on the 32-row tile shape and a 20 TF launch on the square
        // one: they stream whatever the slab length)
        if ((all || p.M > 32) && p.N == 256 && sg.ldb == 256 && p.M <= STREAM_WGRAD_MAX_OUT) {   // few outputs over a 256-wide input
          q.Nout = p.M; q.dY = sg.A; q.lddy = sg.lda; q.X = sg.B; q.sq = p.ldc; q.sk = 1;
        } else if (p.M == 256 && sg.lda == 256 && p.N <= 32) {   // few input columns: dW^T[a][n] = X2[R, a]^T dOut[R, n]
          q.Nout = p.N; q.dY = sg.B; q.lddy = sg.ldb; q.X = sg.A; q.sq = 1; q.sk = p.ldc;
        } else {
          ok = false;
        }
        ok = ok && stream_wgrad_takes(q);
      }
      if (ok) {
        to.swg.push_back(q);
        from.gemm.erase(from.gemm.begin() + i);
      } else {
        ++i;
      }
    }
  }
  // The column sums (bias gradients) and the tiny both-ways-narrow gradients of `skinny` that the streaming kernel takes join
  // its launch when there is one: a launch less at the end of the step (FDQL_NO_COLSUM_STREAM: k_skinny_wgrad keeps them).
  void colsums_into_stream(Stage &skinny, Stage &to) {
    if (to.swg.empty() || getenv("FDQL_NO_COLSUM_STREAM")) return;
    std::vector<SkinnyWgradProblem> first;   // (short waves: dispatched ahead of the streaming ones, they end under them)
    for (size_t i = 0; i < skinny.swg.size();) {
      if (stream_wgrad_takes(skinny.swg[i])) {
        first.push_back(skinny.swg[i]);
        skinny.swg.erase(skinny.swg.begin() + i);
      } else {
        ++i;
      }
    }
    to.swg.insert(to.swg.begin(), first.begin(), first.end());
  }
  // all weight / bias gradients of one MLP instance into the K-split slabs.  Every weight
  // gradient is a K-split GEMM (the narrow ones on the 128x32 / 32x128 tiles); bias gradients
  // come from the per-tile column sums the dgrad GEMMs leave behind (dy_cs / dpre_cs), or
  // directly from dY when dY is narrow (dz, d logits).
  void wgrads(const MlpInst &m, const float *dY, int lddy, const float *dy_cs, Stage &gs, Stage &narrow, Stage &ws, int dy_cs_rows = 0) {
    const MlpDesc &d = *m.d;
    float *slab = a->buf("slabs");
    const long long P = a->n_train;
    const int S = a->nsplit, R = m.rows;
    auto gemm_w = [&](const float *dOut, int ldo, int nout, const float *X, int ldx, int width, float *dst, int ldw) {
      wgrad_gemm(R, dOut, ldo, nout, X, ldx, width, dst, ldw, gs, narrow);
    };
    auto bias_w = [&](const float *dOut, int ldo, int nout, const float *cs, float *dst, int cs_rows = 0) {
      wgrad_bias(R, dOut, ldo, nout, cs, dst, ws, cs_rows);
    };
    for (size_t i = 0; i < d.hid.size(); ++i) {
      float *dst = slab + d.w_off[i];
      if (i == 0) {
        int col = 0;
        for (const SegIn &s : m.in) { gemm_w(m.dpre[0], d.hid[0], d.hid[0], s.ptr, s.ld, s.width, dst + col, d.din); col += s.width; }
      } else {
        gemm_w(m.dpre[i], d.hid[i], d.hid[i], m.h[i - 1], d.hid[i - 1], d.hid[i - 1], dst, d.hid[i - 1]);
      }
      bias_w(m.dpre[i], d.hid[i], d.hid[i], m.dpre_cs[i], slab + d.b_off[i], i < m.dpre_cs_rows.size() ? m.dpre_cs_rows[i] : 0);
    }
    float *dst = slab + d.hw_off;
    const int ld = d.head_ld();
    int col = 0;
    for (const SegIn &s : m.in) { gemm_w(dY, lddy, d.dout, s.ptr, s.ld, s.width, dst + col, ld); col += s.width; }
    for (size_t i = 0; i < d.hid.size(); ++i) { gemm_w(dY, lddy, d.dout, m.h[i], d.hid[i], d.hid[i], dst + col, ld); col += d.hid[i]; }
    bias_w(dY, lddy, d.dout, dy_cs, slab + d.hb_off, dy_cs_rows);
  }
};

// ------------------------------------------------------------------------ chain programs (chain.h)
// Emits the operations of MLP forward passes for the row-block chain kernel.  `ok` turns false as soon as something
// does not fit the kernel (a layer wider than 256 columns, more K-segments than an operation holds, LDS exhausted,
// too many operations): the caller then drops the stage and keeps the per-layer GEMM launches.
struct ChainImg { int slot = -1, pitch = 0, K = 0, size = 0; };
struct RowWin { int lo, hi, shift; };

struct ChainBuilder {
  Stage &st;
  bool ok = true;
  int op_start = 0, rows = 0, peak = 0;
  int stage_top = CH_LDS_FLOATS;      // weight staging areas of the CH_NARROW operations: carved downwards from the top of
                                      // LDS, alive for the whole program (they are filled before its first operation)
  std::vector<std::pair<int, int>> used;   // live LDS ranges (offset, size) of the program being built
  int bm;                             // rows per workgroup: images are [bm][pitch]
  explicit ChainBuilder(Stage &s) : st(s), bm(s.chain_bm) {}

  void begin(int nrows) { rows = nrows; op_start = (int)st.cops.size(); used.clear(); peak = 0; stage_top = CH_LDS_FLOATS; }
  void end() {
    ChainOp e;
    memset(&e, 0, sizeof(e));
    e.kind = CH_END;
    st.cops.push_back(e);
    if ((int)st.cops.size() - op_start > CH_MAX_OPS) ok = false;
    ChainProblem p;
    memset(&p, 0, sizeof(p));
    const int need = stage_top < CH_LDS_FLOATS ? CH_LDS_FLOATS : peak;   // staging areas sit at the top of the budget
    p.rows = rows; p.op_start = op_start; p.nops = (int)st.cops.size() - op_start; p.lds_floats = need;
    st.cprobs.push_back(p);
    st.lds_floats = std::max(st.lds_floats, need);
  }
  int alloc(int size) {   // first fit; sizes are multiples of 4 floats (16-byte aligned images)
    size = (size + 3) & ~3;
    std::sort(used.begin(), used.end());
    int at = 0;
    for (auto &u : used) {
      if (u.first - at >= size) break;
      at = u.first + u.second;
    }
    if (at + size > stage_top) { ok = false; return 0; }
    used.push_back({at, size});
    peak = std::max(peak, at + size);
    return at;
  }
  void release(const ChainImg &im) {
    for (size_t i = 0; i < used.size(); ++i)
      if (used[i].first == im.slot) { used.erase(used.begin() + i); return; }
  }
  ChainImg image(int K, int min_size = 0) {
    ChainImg im;
    im.K = K; im.pitch = chain_pitch(K); im.size = std::max(bm * im.pitch, min_size);
    im.slot = alloc(im.size);
    return im;
  }
  static ChainOp new_op(int kind) {
    ChainOp o;
    memset(&o, 0, sizeof(o));
    o.kind = kind; o.out_slot = -1;
    return o;
  }
  ChainImg load(const std::vector<SegIn> &segs, int min_size = 0) {
    int K = 0;
    for (auto &s : segs) K += s.width;
    ChainImg im = image(K, min_size);
    ChainOp o = new_op(CH_LOAD);
    if ((int)segs.size() > CH_MAX_SEG) { ok = false; return im; }
    o.slot = im.slot; o.pitch = im.pitch; o.kpad = chain_kpad(K); o.nseg = (int)segs.size();
    int col = 0;
    for (size_t i = 0; i < segs.size(); ++i) {
      o.ld[i].src = segs[i].ptr; o.ld[i].ld = segs[i].ld; o.ld[i].width = segs[i].width; o.ld[i].col = col;
      col += segs[i].width;
    }
    st.cops.push_back(o);
    return im;
  }
  // one Linear layer over cat(ins): W rows of pitch ldw, the k-th input image reads columns starting at its offset
  // in the concatenation.  dst: the LDS image that receives the result (may alias a dying input; slot -1: none)
  void gemm(const std::vector<ChainImg> &ins, const float *W, int ldw, int N, const float *bias, int act, const ChainImg &dst,
            float *out, int ldo, RowWin win, const float *rider_w = nullptr, int rider_ld = 0, int rider_n = 0, bool rider_begin = false) {
    if (N > 256 || (int)ins.size() > CH_MAX_SEG) { ok = false; return; }
    ChainOp o = new_op(CH_GEMM);
    o.N = N; o.flags = CHF_ZERO | CHF_EMIT | (rider_w && rider_begin ? CHF_HBEGIN : 0); o.act = act; o.bias = bias; o.nseg = (int)ins.size();
    o.hw = rider_w; o.hldw = rider_ld; o.hN = rider_n;
    if (rider_w) st.flops += 2.0 * (std::min(rows, win.hi) - win.lo) * (double)rider_n * [&] { int k = 0; for (auto &im : ins) k += im.K; return k; }();
    int col = 0;
    for (size_t i = 0; i < ins.size(); ++i) {
      o.seg[i].W = W + col; o.seg[i].ldw = ldw; o.seg[i].slot = ins[i].slot; o.seg[i].pitch = ins[i].pitch; o.seg[i].K = ins[i].K;
      col += ins[i].K;
    }
    o.out_slot = dst.slot; o.out_pitch = dst.pitch;
    o.out = out; o.ldo = ldo; o.row_lo = win.lo; o.row_hi = win.hi; o.row_shift = win.shift;
    st.cops.push_back(o);
    st.flops += chain_op_flops(o, std::min(rows, win.hi) - win.lo);
  }
  // part of a narrow head (N <= 32) over cat(ins) starting at column `col0` of the head weight
  void narrow(const std::vector<ChainImg> &ins, const float *W, int ldw, int col0, int N, bool begin, bool finish,
              const float *bias, float *out, int ldo, RowWin win, int scratch_slot) {
    if (N > 32 || (int)ins.size() > CH_MAX_SEG) { ok = false; return; }
    ChainOp o = new_op(CH_NARROW);
    o.N = N; o.flags = (begin ? CHF_BEGIN : 0) | (finish ? CHF_FINISH : 0); o.nseg = (int)ins.size();
    int col = col0;
    for (size_t i = 0; i < ins.size(); ++i) {
      o.seg[i].W = W + col; o.seg[i].ldw = ldw; o.seg[i].slot = ins[i].slot; o.seg[i].pitch = ins[i].pitch; o.seg[i].K = ins[i].K;
      col += ins[i].K;
    }
    int need = 0;
    for (auto &im : ins) need += N * (((im.K + 15) & ~15) + 4);
    need = (need + 3) & ~3;
    stage_top -= need;
    if (stage_top < peak) ok = false;   // (images allocated later are checked against stage_top in alloc())
    for (auto &u : used) if (u.first + u.second > stage_top) ok = false;
    (void)scratch_slot;
    o.slot = stage_top; o.bias = bias; o.out = out; o.ldo = ldo;
    o.row_lo = win.lo; o.row_hi = win.hi; o.row_shift = win.shift;
    st.cops.push_back(o);
    st.flops += chain_op_flops(o, std::min(rows, win.hi) - win.lo);
  }

  // SkipHeadMLP forward (mlp.py:88-94) on images already in LDS.
  //   in_dies: the input images are not needed after this MLP (their LDS may be reused).
  //   hidden activations go to m.h[i] (global, rows of `win`) when store_h; the head output to m.out (global) and,
  //   for a wide head, to the returned LDS image.
  // Narrow head (dout <= 32): the head's dot product is accumulated piecewise (CH_NARROW, per-wave 16-row tiles) as soon as
  // each block of its input exists, so a layer's input image can be overwritten in place by its output: one image per MLP.
  ChainImg mlp(const MlpInst &m, std::vector<ChainImg> ins, bool in_dies, bool store_h, RowWin win, bool want_out_image) {
    const MlpDesc &d = *m.d;
    const int nh = (int)d.hid.size(), ld_head = d.head_ld();
    ChainImg none;
    if (d.dout <= 32 && !want_out_image) {   // (an output that feeds the next MLP from LDS takes the GEMM path)
      // The head's block over a layer's INPUT rides in that layer's K loop when the layer is wide enough for every wave to
      // own a column tile (chain.hip, RIDER); otherwise it is a CH_NARROW pass of its own.  The block over the last
      // hidden activation always is one (nothing follows it to ride in).
      auto rides = [&](int i) { return i < nh && d.hid[i] > 192; };
      bool begun = false;
      if (!rides(0)) { narrow(ins, m.HW(), ld_head, 0, d.dout, true, nh == 0, m.HB(), m.out, m.ldout, win, 0); begun = true; }
      int col = 0;
      std::vector<ChainImg> cur = ins;
      bool cur_dies = in_dies;
      for (int i = 0; i < nh; ++i) {
        // the layer's output image: in place of its (single, dying) input when possible, else a new one
        const int need = bm * chain_pitch(d.hid[i]);
        ChainImg dst;
        if (cur_dies && cur.size() == 1 && cur[0].size >= need) {
          dst = cur[0];
          dst.K = d.hid[i]; dst.pitch = chain_pitch(d.hid[i]);
        } else {
          if (cur_dies) for (auto &c : cur) release(c);
          dst = image(d.hid[i], need);
        }
        const bool ride = rides(i);
        gemm(cur, m.W(i), d.in_of(i), d.hid[i], m.Bv(i), CHA_LRELU, dst, store_h ? m.h[i] : nullptr, d.hid[i], win,
             ride ? m.HW() + col : nullptr, ld_head, d.dout, ride && !begun);
        if (ride) begun = true;
        col += d.in_of(i);
        const bool last = i + 1 == nh;
        if (last || !rides(i + 1)) narrow({dst}, m.HW(), ld_head, col, d.dout, !begun, last, m.HB(), m.out, m.ldout, win, 0);
        begun = true;
        cur = {dst};
        cur_dies = true;
      }
      if (nh > 0) release(cur[0]);
      else if (in_dies) for (auto &c : ins) release(c);
      return none;
    }
    // wide head: every feature block stays in LDS until the head GEMM has read it
    std::vector<ChainImg> feats = ins, cur = ins;
    for (int i = 0; i < nh; ++i) {
      ChainImg dst = image(d.hid[i]);
      gemm(cur, m.W(i), d.in_of(i), d.hid[i], m.Bv(i), CHA_LRELU, dst, store_h ? m.h[i] : nullptr, d.hid[i], win);
      feats.push_back(dst);
      cur = {dst};
    }
    // the head's output image may reuse what dies here: the epilogue writes only after every wave has finished reading
    for (size_t i = in_dies ? 0 : ins.size(); i < feats.size(); ++i) release(feats[i]);
    ChainImg dst;
    if (want_out_image) dst = image(d.dout);
    gemm(feats, m.HW(), ld_head, d.dout, m.HB(), CHA_NONE, dst, m.out, m.ldout, win);
    return dst;
  }
};

MlpInst make_inst(fdql_agent *a, const MlpDesc &d, const std::string &p, const float *wbase, int64_t worigin, int rows,
                  bool bwd) {
  MlpInst m;
  m.d = &d;
  m.wbase = wbase;
  m.worigin = worigin;
  m.rows = rows;
  for (size_t i = 0; i < d.hid.size(); ++i) {
    m.h.push_back(a->buf(p + ".h" + std::to_string(i)));
    if (bwd) {
      m.dpre.push_back(a->buf(p + ".dpre" + std::to_string(i)));
      m.dpre_cs.push_back(a->buf(p + ".cs" + std::to_string(i)));
      m.dpre_cs_rows.push_back(0);
    }
    if (a->has_buf(p + ".gm" + std::to_string(i))) m.gm.push_back(reinterpret_cast<unsigned *>(a->buf(p + ".gm" + std::to_string(i))));
  }
  return m;
}

// Which row-block kernel takes a group of like problems (one launch): the weight-stationary one when it has the form,
// else the streamed-weights one, else none (the tile kernels).  Deterministic in (problems, environment): the plan
// builder asks the same question where the answer changes what other stages read (the dgrad form's column sums).
bool rows_launch_of(const fdql_agent *a, const std::vector<GemmProblem> &grp, RowsLaunch &rl) {
  const long long tiles = (long long)grp.size() * (grp[0].M / RG_BM);
  if (tiles < a->rows_min_tiles) return false;
  if (wstat_from_problems(grp.data(), (int)grp.size(), rl.wa)) { rl.ws = true; return true; }
  rl.ws = false;
  return rowgemm_from_problems(grp.data(), (int)grp.size(), rl.rg);
}

int upload_tables(fdql_agent *a) {
  auto pad = [](size_t b) { return (b + 255) / 256 * 256; };
  size_t total = 0;
  for (Stage &s : a->stages) {
    if (s.kind == ST_GEMM) {
      for (auto &sub : s.sub) sub.probs.clear();
      s.rows.clear();
      std::vector<char> taken(s.gemm.size(), 0);
      if (s.try_rows) {   // like problems (same segment list shape and epilogue) -> one row-block launch per group
        {   // ... or the whole stage as ONE weight-stationary launch (critic layer 0: two-output and plain instances mixed)
          RowsLaunch rl;
          if (s.gemm.size() > 1 && rows_launch_of(a, s.gemm, rl) && rl.ws) {
            s.rows.push_back(rl);
            std::fill(taken.begin(), taken.end(), 1);
          }
        }
        for (size_t i = 0; i < s.gemm.size(); ++i) {
          if (taken[i]) continue;
          std::vector<GemmProblem> grp;
          std::vector<size_t> idx;
          for (size_t j = i; j < s.gemm.size(); ++j) {
            const GemmProblem &p = s.gemm[j], &q = s.gemm[i];
            bool like = !taken[j] && p.nseg == q.nseg && p.emit_seg == q.emit_seg && p.epi == q.epi && (p.C2 != nullptr) == (q.C2 != nullptr);
            for (int sg = 0; like && sg < p.nseg; ++sg)
              like = p.seg[sg].K == q.seg[sg].K && p.seg[sg].lda == q.seg[sg].lda && p.seg[sg].ldb == q.seg[sg].ldb &&
                     p.seg[sg].a_kc == q.seg[sg].a_kc && p.seg[sg].b_kc == q.seg[sg].b_kc;
            if (like) {
              grp.push_back(p);
              idx.push_back(j);
            }
          }
          RowsLaunch rl;
          if (rows_launch_of(a, grp, rl)) {
            s.rows.push_back(rl);
            for (size_t j : idx) taken[j] = 1;
          } else {
            for (size_t j : idx) taken[j] = 2;   // looked at, stays on the tile kernels
          }
        }
      }
      // a whole stage of narrow-output dgrads (d pi: one problem per frozen critic) as one streaming launch
      if (!s.gemm.empty() && std::find(taken.begin(), taken.end(), (char)1) == taken.end() && s.gemm[0].M >= a->rowdot_min_rows) {
        RowsLaunch rl;
        if (rowdot_from_problems(s.gemm.data(), (int)s.gemm.size(), rl.rdot)) {
          rl.dot = true;
          s.rows.push_back(rl);
          std::fill(taken.begin(), taken.end(), 1);
        }
      }
      // single-network dgrads (256-wide K-strided segments, gate / column sums) with enough 64-row blocks to fill most of the chip
      for (size_t i = 0; i < s.gemm.size(); ++i) {
        RowsLaunch rl;
        if (taken[i] != 1 && s.gemm[i].M / RD_BM >= a->rowdgrad_min_blocks &&
            s.gemm[i].M / RD_BM <= (s.rd_max_blocks > 0 ? s.rd_max_blocks : a->rowdgrad_max_blocks) &&
            rowdgrad_from_problem(s.gemm[i], rl.rda)) {
          if (s.fold_sum && !rowdgrad_fold_sum(rl.rda, s.fold_parts, s.fold_n, s.fold_stride, s.fold_out, s.fold_cs)) {
            set_error("stage %s: the row-block dgrad kernel does not take the folded sum it was planned with", s.name.c_str());
            return FDQL_EINVAL;
          }
          rl.rd = true;
          s.rows.push_back(rl);
          taken[i] = 1;
        }
      }
      if (s.fold_sum && (s.rows.size() != 1 || !s.rows[0].rd)) {
        set_error("stage %s was planned with a folded sum but does not run on the row-block dgrad kernel", s.name.c_str());
        return FDQL_EINVAL;
      }
      // Small batches (temporal_len 2, a handful of env rows): a stage whose problems are too few tiles to fill the chip is as
      // long as one workgroup's serial K loop on the tile kernels; the small-batch kernel (smallgemm.hip) splits K over the 16
      // waves of a workgroup instead.  Per stage: every problem left for the tile kernels must have the form, and together they
      // are at most small_max_tiles 64x64 tiles (FDQL_SMALL_GEMM=0: never; FDQL_SMALL_GEMM_MAX_TILES).
      // A non-default tile shape / main-loop build (tests, experiments) keeps its kernels; dense shape GEMM_SMALL (test hook)
      // forces the small-batch kernel on every problem that has the form, whatever the size.
      const int dshape = gemm_dense_shape();
      const bool small_forced = dshape == GEMM_SMALL;
      bool small = a->small_max_tiles > 0 && dshape == GEMM_64x64 && gemm_variant() == GEMM_DEFAULT_VARIANT;
      long long tiles64 = 0;
      for (size_t i = 0; i < s.gemm.size() && small; ++i) {
        if (taken[i] == 1) continue;
        small = gemm_small_takes(s.gemm[i]);
        tiles64 += (long long)((s.gemm[i].M + 63) / 64) * ((s.gemm[i].N + 63) / 64);
      }
      // (short-K stages - the rank-Q outer product of a last hidden layer's gradient under a Q-wide head - are one load round per
      // workgroup whatever their tile count: up to 4x the tiles)
      int kmax = 0;
      for (size_t i = 0; i < s.gemm.size(); ++i)
        if (taken[i] != 1) { int k = 0; for (int sg = 0; sg < s.gemm[i].nseg; ++sg) k += s.gemm[i].seg[sg].K; kmax = std::max(kmax, k); }
      small = small && tiles64 > 0 && (tiles64 <= a->small_max_tiles || (kmax <= 32 && tiles64 <= 4LL * a->small_max_tiles));
      for (size_t i = 0; i < s.gemm.size(); ++i) {
        if (taken[i] == 1) continue;
        if (s.gemm[i].fz_h) { set_error("stage %s: a fused head-dgrad problem was not taken by the row-block kernel", s.name.c_str()); return FDQL_ESTATE; }
        const bool sm = small || (small_forced && gemm_small_takes(s.gemm[i]));
        s.sub[sm ? (int)GEMM_SMALL : gemm_pick_shape(s.gemm[i], small_forced ? (int)GEMM_64x64 : dshape)].probs.push_back(s.gemm[i]);
      }
      for (auto &sub : s.sub) total += pad(sub.probs.size() * sizeof(GemmProblem));
    }
    if (s.kind == ST_SKINNY_WGRAD) total += pad(s.swg.size() * sizeof(SkinnyWgradProblem));
    if (s.kind == ST_HEAD_DGRAD) total += pad(s.hdg.size() * sizeof(HeadDgradProblem));
    if (s.kind == ST_CHAIN) total += pad(s.cprobs.size() * sizeof(ChainProblem)) + pad(s.cops.size() * sizeof(ChainOp));
  }
  if (a->tables_dev) { FDQL_HIP(hipFree(a->tables_dev)); a->tables_dev = nullptr; }
  FDQL_HIP(hipMalloc(&a->tables_dev, total ? total : 256));
  std::vector<char> host(total);
  size_t off = 0;
  for (Stage &s : a->stages) {
    if (s.kind == ST_GEMM) {
      s.flops = 0; s.bytes = 0;
      for (auto &p : s.gemm) { s.flops += gemm_flops(p); s.bytes += gemm_bytes(p); }
      for (int sh = 0; sh < GEMM_NSHAPES; ++sh) {
        GemmSub &sub = s.sub[sh];
        if (sub.probs.empty()) { sub.blocks = 0; sub.dev = nullptr; continue; }
        sub.blocks = gemm_finalize(sub.probs.data(), (int)sub.probs.size(), sh);
        const size_t bytes = sub.probs.size() * sizeof(GemmProblem);
        memcpy(host.data() + off, sub.probs.data(), bytes);
        sub.dev = (char *)a->tables_dev + off;
        off += pad(bytes);
      }
    } else if (s.kind == ST_SKINNY_WGRAD) {
      s.blocks = s.stream ? stream_wgrad_finalize(s.swg.data(), (int)s.swg.size()) : skinny_wgrad_finalize(s.swg.data(), (int)s.swg.size());
      s.flops = 0; s.bytes = 0;
      for (auto &p : s.swg) {
        s.flops += 2.0 * p.M * (double)p.K * p.Nout;
        s.bytes += 4.0 * p.M * ((double)p.K + (p.dY ? p.Nout : 0));
      }
      const size_t bytes = s.swg.size() * sizeof(SkinnyWgradProblem);
      memcpy(host.data() + off, s.swg.data(), bytes);
      s.dev = (char *)a->tables_dev + off;
      off += pad(bytes);
    } else if (s.kind == ST_HEAD_DGRAD) {
      s.blocks = head_dgrad_finalize(s.hdg.data(), (int)s.hdg.size());
      s.flops = 0; s.bytes = 0;
      for (auto &p : s.hdg) {
        s.flops += 2.0 * p.M * (double)p.N * p.Q;
        s.bytes += 8.0 * p.M * (double)p.N;   // read h, write dpre
      }
      const size_t bytes = s.hdg.size() * sizeof(HeadDgradProblem);
      memcpy(host.data() + off, s.hdg.data(), bytes);
      s.dev = (char *)a->tables_dev + off;
      off += pad(bytes);
    } else if (s.kind == ST_CHAIN) {
      s.blocks = chain_finalize(s.cprobs.data(), (int)s.cprobs.size(), s.chain_bm);
      size_t bytes = s.cprobs.size() * sizeof(ChainProblem);
      memcpy(host.data() + off, s.cprobs.data(), bytes);
      s.dev = (char *)a->tables_dev + off;
      off += pad(bytes);
      bytes = s.cops.size() * sizeof(ChainOp);
      memcpy(host.data() + off, s.cops.data(), bytes);
      s.cops_dev = (char *)a->tables_dev + off;
      off += pad(bytes);
    }
  }
  if (total) FDQL_HIP(hipMemcpy(a->tables_dev, host.data(), total, hipMemcpyHostToDevice));
  // three dependent single-network dgrads in a row on the row-block dgrad kernel, the first with the folded sum of the d state
  // shares (a config-2-shaped plan: joiner.dpre0, d enc, enc_obs.dpre0): one launch with the 64-row activations resident in LDS
  for (Stage &s : a->stages)   // (decided anew with every table upload)
    if (s.chained) { s.off = false; s.chained = false; }
  for (size_t i = 0; i + 2 < a->stages.size(); ++i) {
    Stage &s1 = a->stages[i], &s2 = a->stages[i + 1], &s3 = a->stages[i + 2];
    auto lone_rd = [](const Stage &s) {
      if (s.kind != ST_GEMM || s.gemm.size() != 1 || s.rows.size() != 1 || !s.rows[0].rd) return false;
      for (const GemmSub &sub : s.sub) if (!sub.probs.empty()) return false;
      return true;
    };
    if (!s1.fold_sum || !lone_rd(s1) || !lone_rd(s2) || !lone_rd(s3) || s1.phase != s2.phase || s1.phase != s3.phase) continue;
    RowChainArgs c;
    if (!rowchain_from_launches(s1.rows[0].rda, s2.rows[0].rda, s3.rows[0].rda, c)) continue;
    s1.rows[0].chain3 = true;
    s1.rows[0].rch = c;
    s2.off = s3.off = true;
    s2.chained = s3.chained = true;
    // the launch does their work (fdql_agent_stats counts executed flops); s1's own figures were recomputed from its
    // problems at the top of this upload, so a second upload does not add them twice
    s1.flops += s2.flops + s3.flops;
    s1.bytes += s2.bytes + s3.bytes;
  }
  // head-fusion planes: with every hidden layer of the critics on weight-stationary launches, those launches sum a tile's column
  // planes themselves, the plane-sum stage is switched off and the finish adds one plane per layer (Stage::hf_role)
  {
    Stage *fin = nullptr, *sum = nullptr;
    int nfwd = 0;
    bool all = true;
    for (Stage &s : a->stages) {
      if (s.hf_role == 2) sum = &s;
      if (s.hf_role == 3) fin = &s;
      if (s.hf_role != 1) continue;
      ++nfwd;
      size_t n = 0;
      bool ok = !s.rows.empty();
      for (const RowsLaunch &rl : s.rows) { ok = ok && rl.ws && rl.wa.hf_q > 0; n += (size_t)rl.wa.ninst; }
      all = all && ok && n == s.gemm.size();
    }
    const bool presum = fin && fin->hfin_can_presum && nfwd > 0 && nfwd == fin->hfin_presum.planes && all;
    for (Stage &s : a->stages)
      if (s.hf_role == 1)
        for (RowsLaunch &rl : s.rows)
          if (rl.ws) rl.wa.hf_presum = presum ? 1 : 0;
    if (fin) *fin->hfin = presum ? fin->hfin_presum : fin->hfin_plain;
    if (sum) sum->off = presum;
  }
  // gate masks: written by the critics' forward launches and read by their backward launches only when every one of those
  // forward layers runs weight-stationary (the mask layout is that kernel's register layout); FDQL_NO_GATE_MASKS: never
  {
    int nfwd = 0;
    bool all = getenv("FDQL_NO_GATE_MASKS") == nullptr;
    for (Stage &s : a->stages) {
      if (s.gm_role != 1) continue;
      ++nfwd;
      size_t n = 0;
      bool ok = !s.rows.empty();
      for (const RowsLaunch &rl : s.rows) { ok = ok && rl.ws; n += (size_t)rl.wa.ninst; }
      all = all && ok && n == s.gemm.size();
    }
    const bool masks = nfwd > 0 && all;
    for (Stage &s : a->stages) {   // a stage planned on the masks and its GEMM stand-in: exactly one of them runs
      if (s.needs_masks) s.off = !masks;
      if (s.masks_fallback) s.off = masks;
    }
    for (Stage &s : a->stages) {
      for (RowsLaunch &rl : s.rows) {
        if (!rl.ws) continue;
        if (rl.wa.grad == 0) {   // forward launches: nobody reads masks written outside the scheme
          if (!masks || s.gm_role != 1)
            for (int i = 0; i < rl.wa.ninst; ++i) rl.wa.inst[i].gm_out = rl.wa.inst[i].gm_out2 = nullptr;
        } else if (s.gm_role == 2 && rl.wa.grad == 1) {   // gated dgrad forms: every instance must carry the masks it would read
          bool have = masks;
          for (int i = 0; i < rl.wa.ninst; ++i) have = have && rl.wa.inst[i].gm_ref && (!rl.wa.fz || rl.wa.inst[i].gm_fz);
          rl.wa.use_masks = have ? 1 : 0;
        }
      }
    }
  }
  return 0;
}

int build_plan(fdql_agent *a) {
  const fdql_agent_config_t &c = a->cfg;
  a->stages.clear();
  Builder b(a);
  const int N = a->N, M = a->M, B = a->B, L = c.latent, A = c.act_dim, C = c.n_critics, Q = c.n_quantiles, Nq = a->Nq;
  const fdql_batch_t &x = a->batch;
  float *params = a->params, *targets = a->targets;

  // ---- instances
  MlpInst eo = make_inst(a, a->enc_obs, "enc_obs", params, 0, N, true);
  if (c.obs_dim) eo.in.push_back({x.obs_1d, c.obs_dim, c.obs_dim});
  if (c.goal_dim) {
    eo.in.push_back({x.achieved_goal, c.goal_dim, c.goal_dim});
    eo.in.push_back({x.desired_goal, c.goal_dim, c.goal_dim});
  }
  const int nconv = (int)a->conv.size();
  const int conv_col0 = c.obs_dim + 2 * c.goal_dim;   // first column of the conv features in the obs MLP's input
  if (nconv) {
    const float *feat = a->buf("conv" + std::to_string(nconv - 1) + ".out");
    eo.in.push_back({feat, a->conv_feat, a->conv_feat});
  }
  eo.out = a->buf("enc_obs.out"); eo.ldout = c.enc_features;
  const bool gru = c.joiner_gru != 0;
  MlpInst jo;
  float *state = a->buf("state");
  if (!gru) {
    jo = make_inst(a, a->joiner, "joiner", params, 0, N, true);
    jo.in.push_back({eo.out, c.enc_features, c.enc_features});
    jo.out = state; jo.ldout = L;
  }
  const float *s_cur = state, *s_nxt = state + (int64_t)B * L;

  MlpInst at = make_inst(a, a->actor, "actor_t", targets, a->tgt_begin, M, false);
  at.in.push_back({s_nxt, L, L});
  at.out = a->buf("actor_t.out"); at.ldout = a->actor.dout;
  MlpInst ao = make_inst(a, a->actor, "actor", params, 0, M, true);
  ao.in.push_back({s_cur, L, L});
  ao.out = a->buf("actor.out"); ao.ldout = a->actor.dout;

  std::vector<MlpInst> ct, co, cf;
  for (int k = 0; k < C; ++k) {
    const std::string s = std::to_string(k);
    MlpInst t = make_inst(a, a->critic[k], "crit_t" + s, targets, a->tgt_begin, M, false);
    t.in.push_back({s_nxt, L, L});
    t.in.push_back({a->buf("next_action"), A, A});
    t.out = a->buf("next_z") + k * Q; t.ldout = Nq;
    ct.push_back(t);
    MlpInst o = make_inst(a, a->critic[k], "crit" + s, params, 0, M, true);
    o.in.push_back({s_cur, L, L});
    o.in.push_back({c.discrete ? a->buf("action_onehot") : x.action, A, A});
    o.out = a->buf("q_pred") + k * Q; o.ldout = Nq;
    co.push_back(o);
    MlpInst f = make_inst(a, a->critic[k], "crit_f" + s, params, 0, M, true);
    f.in.push_back({s_cur, L, L});
    f.in.push_back({a->buf("pi"), A, A});
    f.out = a->buf("q_frozen") + k * Q; f.ldout = Nq;
    cf.push_back(f);
  }

  DevState *dst = a->st();
  const float *log_alpha = params + a->log_alpha_off;

  // ---- stage 0: tick + prep
  bool fold_prep = false;
  PrepArgs prep_args;
  memset(&prep_args, 0, sizeof(prep_args));
  {
    const float inv_gb = 1.0f / (float)(B * (c.world_size > 0 ? c.world_size : 1));
    float *w = a->buf("w"), *ic = a->buf("is_contiguous");
    const float *td = x.task_done, *es = x.episode_step;
    const int T = a->T;
    const int burn = c.burn_in_steps;
    const int cumprod = gru ? 1 : 0;   // encoder.py:80
    // continuous policies: prep's workgroups ride in the policy-forward launch (nothing before the loss reads what it writes)
    fold_prep = !c.discrete && getenv("FDQL_NO_PREP_FOLD") == nullptr;
    prep_args = PrepArgs{td, es, T, B, burn, cumprod, inv_gb, w, ic, dst, log_alpha};
    if (!fold_prep)
      b.func_stage("prep", [=](hipStream_t s) { return prep_launch(td, es, T, B, burn, cumprod, inv_gb, w, ic, dst, log_alpha, s); });
    if (c.discrete) {   // stored action index -> one-hot critic input (deepQlearning.py:206-210)
      const float *act = x.action;
      float *oh = a->buf("action_onehot");
      b.func_stage("onehot", [=](hipStream_t s) { return onehot_launch(act, N, A, oh, s); });
    }
  }
  // ---- encoder forward (encoder.py:52-67)
  auto fwd_chain = [&](std::vector<MlpInst *> group, const std::string &name) {
    const size_t nh = group[0]->d->hid.size();
    for (size_t i = 0; i < nh; ++i) {
      Stage &gs = b.gemm_stage(name + ".fwd" + std::to_string(i));
      for (MlpInst *m : group) gs.gemm.push_back(b.fwd_layer(*m, (int)i));
    }
    Stage &hs = b.gemm_stage(name + ".head");
    for (MlpInst *m : group) hs.gemm.push_back(b.fwd_head(*m));
  };
  for (int i = 0; i < nconv; ++i) {   // pixel encoder forward: im2col + GEMM (bias, LeakyReLU) per layer, all N images
    const fdql_agent::ConvLayer &Lc = a->conv[i];
    const ConvGeom g = Lc.g;
    const int K = g.C * g.k * g.k;
    const long long rows = (long long)N * g.OH * g.OW;
    FDQL_REQUIRE(rows < (1LL << 31), "conv layer %d: %lld im2col rows exceed the GEMM's 32-bit row index", i, rows);
    float *out = a->buf("conv" + std::to_string(i) + ".out");
    if (Lc.fast_fwd) {   // implicit GEMM (conv.hip): the image groups resident in LDS, no column matrix
      ConvFwdArgs ca;
      ca.in.base = i == 0 ? (const void *)x.obs_2d_u8 : (const void *)a->buf("conv" + std::to_string(i - 1) + ".out");
      ca.in.u8 = i == 0; ca.in.slots = i == 0 ? x.obs_2d_slots : nullptr;
      ca.W = params + Lc.w_off; ca.bias = params + Lc.b_off; ca.out = out; ca.nimg = N; ca.g = g; ca.cout = Lc.cout;
      Stage &cs = b.func_stage("conv.fwd" + std::to_string(i), [=](hipStream_t s) { return conv_fwd_launch(ca, s); });
      cs.mfma = true;
      cs.flops = 2.0 * (double)rows * K * Lc.cout;
      cs.bytes = (i == 0 ? 1.0 : 4.0) * (double)N * g.C * g.H * g.W + 4.0 * (double)rows * Lc.cout;
      continue;
    }
    FDQL_REQUIRE(i > 0 || x.obs_2d, "conv layer 0 runs on the im2col path: it needs the float32 frames (batch.obs_2d)");
    float *col = a->buf("conv" + std::to_string(i) + ".col");
    const float *in = i == 0 ? x.obs_2d : a->buf("conv" + std::to_string(i - 1) + ".out");
    const int nhwc = i > 0;
    const float scale = i == 0 ? 1.0f / 255.0f : 1.0f;
    const long long nimg = N;
    b.func_stage("conv.im2col", [=](hipStream_t s) { return im2col_launch(in, nhwc, scale, nimg, g, col, s); });
    Stage &gs = b.gemm_stage("conv.fwd" + std::to_string(i));
    GemmProblem p = Builder::new_gemm((int)rows, Lc.cout, out, Lc.cout);
    Builder::add_seg(p, col, K, 1, params + Lc.w_off, K, 1, K);
    p.bias = params + Lc.b_off;
    p.epi = EPI_LRELU;
    gs.gemm.push_back(p);
  }
  // Row-block chain (chain.hip): encoder MLP -> joiner MLP -> online actor and target actor in ONE launch, the
  // activations of a 64-row block resident in LDS from the observation to the policy logits.  Falls back to the
  // per-layer launches when a layer does not fit the kernel (see ChainBuilder).
  // FDQL_CHAIN: "0" never, "1" (default) the encoder/actor chain when the batch fills at least half the chip with
  // 64-row blocks (fewer blocks leave most CUs idle for the length of a whole chain: the per-layer launches with their
  // K-splits are faster there), "all" every eligible program incl. the critics' (measured slower than the grouped
  // launches at config 2 so far: DESIGN.md section 5), regardless of size - the parity tests run all three.
  const char *chain_env = getenv("FDQL_CHAIN");
  const std::string chain_mode = chain_env ? chain_env : "1";
  const bool chain_all = chain_mode == "all";
  int chain_min_blocks = 96;   // (tuning hook FDQL_CHAIN_MIN_BLOCKS)
  if (const char *v = getenv("FDQL_CHAIN_MIN_BLOCKS")) chain_min_blocks = atoi(v);
  // Rows per workgroup: 64 when that many blocks fill the chip, else 32 (twice the workgroups - one rank's share of a
  // data-parallel batch - and images of half the size: 64 rows of a 376-column observation next to a hidden image do not fit
  // the LDS, 32 do).  FDQL_CHAIN_BM = 32 / 64 forces one.
  const bool chain_on = chain_mode != "0" && getenv("FDQL_NO_CHAIN") == nullptr;
  std::vector<int> bms;
  if (const char *v = getenv("FDQL_CHAIN_BM")) {
    bms.push_back(atoi(v) == 32 ? 32 : CH_BM);
  } else {
    if (chain_all || N >= (long long)chain_min_blocks * CH_BM) bms.push_back(CH_BM);
    // 32-row blocks only while they are one round of workgroups (one per CU): measured at config 4, 128 windows per GPU
    // (200 blocks) 1.172 -> 1.154 ms per step against the six per-layer launches; at 256 windows (400 blocks, 1.6 rounds) the
    // chain is the slower one (1.902 -> 1.987 ms)
    int ncu = 256, dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    if (chain_all || (N >= (long long)chain_min_blocks * 32 && (N + 31) / 32 <= ncu)) bms.push_back(32);
  }
  bool enc_chained = false;
  for (size_t t = 0; t < bms.size() && !enc_chained && (chain_all || chain_on) && !gru; ++t) {
    Stage cs;
    cs.kind = ST_CHAIN; cs.name = "enc_joiner_actors"; cs.chain_bm = bms[t];
    ChainBuilder cb(cs);
    cb.begin(N);
    ChainImg xi = cb.load(eo.in);
    ChainImg ei = cb.mlp(eo, {xi}, true, true, {0, N, 0}, true);
    ChainImg si = cb.mlp(jo, {ei}, true, true, {0, N, 0}, true);
    cb.mlp(ao, {si}, false, true, {0, M, 0}, false);
    cb.mlp(at, {si}, true, false, {B, N, B}, false);
    cb.end();
    if (cb.ok) { a->stages.push_back(cs); enc_chained = true; }
  }
  if (enc_chained) {
    // nothing left to launch for these three networks
  } else {
  fwd_chain({&eo}, "enc_obs");
  if (!gru) {
    fwd_chain({&jo}, "joiner");
  } else {
    // GRU joiner (encoder.py:40-42, 63-65): the input projection of all T*B rows is one GEMM; the scan over t is
    // T x (recurrent GEMM [B, L] x [L, 3L] + gate kernel) - sequential by nature and latency-bound at B = 256
    const int L3 = 3 * L, F = c.enc_features, T = a->T;
    float *gi = a->buf("gru.gi"), *gh = a->buf("gru.gh"), *hprev = a->buf("gru.hprev"), *h0 = a->buf("gru.h0");
    const float *wih = params + a->gru_wih, *whh = params + a->gru_whh, *bih = params + a->gru_bih, *bhh = params + a->gru_bhh;
    {
      Stage &gs = b.gemm_stage("gru.gi");
      GemmProblem p = Builder::new_gemm(N, L3, gi, L3);
      Builder::add_seg(p, eo.out, F, 1, wih, F, 1, F);
      p.bias = bih;
      gs.gemm.push_back(p);
    }
    {
      const int mode = c.gru_state_mode;
      const float *src = mode == 1 ? x.agent_state : (mode == 2 ? params + a->gru_h0 : nullptr);
      b.func_stage("gru.h0", [=](hipStream_t s) { return gru_h0_launch(mode, src, h0, B, L, s); });
    }
    const bool scan = gru_scan_takes(B, L);   // the whole scan as ONE persistent launch (gruscan.hip) instead of T x (GEMM + gate kernel)
    if (scan) {
      GruScanArgs ga;
      memset(&ga, 0, sizeof(ga));
      float *pf = a->buf("gru.wpack_f"), *pb = a->buf("gru.wpack_b");
      ga.T = T; ga.B = B; ga.L = L; ga.W = pf; ga.bhh = bhh; ga.gi = gi; ga.h0 = h0; ga.gh = gh; ga.state = state; ga.hprev = hprev;
      b.func_stage("gru.pack", [=](hipStream_t s) { return gru_pack_launch(whh, L, pf, pb, s); });
      b.func_stage("gru.scan", [=](hipStream_t s) { return gru_scan_fwd_launch(ga, s); });
    }
    for (int t = 0; t < T && !scan; ++t) {
      const float *hp = t == 0 ? h0 : state + (int64_t)(t - 1) * B * L;
      float *gh_t = gh + (int64_t)t * B * L3, *h_t = state + (int64_t)t * B * L, *hs_t = hprev + (int64_t)t * B * L;
      const float *gi_t = gi + (int64_t)t * B * L3;
      float *ghp = a->buf("gru.ghp");
      Stage &gs = b.gemm_stage("gru.gh");
      GemmProblem p = Builder::new_gemm(B, L3, ghp, L3);
      Builder::add_seg(p, hp, L, 1, whh, L, 1, L);
      p.ksplit = GRU_KSPLIT_FWD;
      p.split_stride = (long long)B * L3;
      gs.gemm.push_back(p);
      b.func_stage("gru.cell", [=](hipStream_t s) {
        return gru_cell_fwd_launch(gi_t, gh_t, ghp, GRU_KSPLIT_FWD, bhh, hp, h_t, hs_t, B, L, s);
      });
    }
  }
  fwd_chain({&at, &ao}, "actors");
  }
  // ---- policy sampling (gaussian_mlp.py:15-39)
  {
    PolicyFwdArgs p0{at.out, nullptr, nullptr, a->buf("next_action"), a->buf("next_log_pi"), 0u, nullptr, nullptr};
    PolicyFwdArgs p1{ao.out, nullptr, a->buf("noise_actor"), a->buf("pi"), a->buf("log_pi"), 1u,
                     c.discrete ? a->buf("action_onehot") : x.action, a->buf("pi_diff")};
    fdql_agent *ag = a;
    const PrepArgs pra = prep_args;
    const bool with_prep = fold_prep;
    b.func_stage("policy_fwd", [=](hipStream_t s) {
      PolicyFwdArgs q0 = p0, q1 = p1;
      q0.noise = ag->noise_t;
      q1.noise = ag->noise_a;
      return policy_fwd_launch(q0, q1, 2, M, A, dst, ag->seed, ag->cfg.discrete, s, with_prep ? &pra : nullptr);
    });
  }
  // ---- critics forward: target(next, a'), online(cur, a), frozen(cur, pi)
  {
    // critic_frozen is a copy of critic taken when the actor loss is formed (soft_actor_critic.py:142),
    // so both read the online weights and layer 0 of q(s, a) and q(s, pi) shares s.Ws: ONE problem per
    // critic accumulates cat(s, a), stores h0 of the online pass, continues with (pi - a).Wa and stores
    // h0 of the frozen pass (GemmProblem::emit_seg) - 10 layer-0 problems instead of 15.
    const size_t nh = a->critic[0].hid.size();
    // Head fusion: each hidden layer's launch also forms its part of the skip head's dot product (GemmProblem::hf_*),
    // so the head streams only cat(s, a) instead of every hidden activation again (584 -> 197 MB at config 2).
    // Needs the Q outputs of a critic to be 1, 2, 4 or 8 (the butterfly's group size).
    bool fuse = nh > 0 && getenv("FDQL_NO_HEAD_FUSE") == nullptr && (Q == 1 || Q == 2 || Q == 4 || Q == 8);
    float *hf_parts = a->buf("hf.parts"), *hf_sum = a->buf("hf.sum");
    const long long MQ = (long long)M * Q;
    auto inst_id = [&](int k, int which) { return 3 * k + which; };   // which: 0 target, 1 online, 2 frozen
    auto plane0 = [&](int layer) { int p = 0; for (int i = 0; i < layer; ++i) p += ((a->critic[0].hid[i] + 63) / 64) * 2; return p; };
    auto set_hf = [&](GemmProblem &p, const MlpInst &m, int layer, int inst, bool second) {
      if (!fuse) return;
      p.hf_w = m.HW() + b.head_col_of_hidden(*m.d, layer);
      p.hf_ldw = m.d->head_ld();
      p.hf_q = Q;
      float *out = hf_parts + ((long long)inst * a->hf_planes + plane0(layer)) * MQ;
      if (second) p.hf_out2 = out; else p.hf_out = out;
    };
    bool crit_chained = false;
    if (chain_all) {   // every critic instance as one chain program: cat(s, a) -> hidden layers -> skip head
      Stage cs;
      cs.kind = ST_CHAIN; cs.name = "critics.fwd";
      if (const char *v = getenv("FDQL_CHAIN_BM")) cs.chain_bm = atoi(v) == 32 ? 32 : CH_BM;
      ChainBuilder cb(cs);
      for (int k = 0; k < C && cb.ok; ++k) {
        int which = 0;
        for (MlpInst *m : {&ct[k], &co[k], &cf[k]}) {
          cb.begin(M);
          ChainImg xi = cb.load(m->in);
          cb.mlp(*m, {xi}, true, which != 0, {0, M, 0}, false);   // target activations are never needed again
          cb.end();
          ++which;
        }
      }
      if (cb.ok) { a->stages.push_back(cs); crit_chained = true; }
    }
    if (crit_chained) {
      // done
    } else if (nh > 0 && getenv("FDQL_NO_DUAL") == nullptr) {
      Stage &gs = b.gemm_stage("critics.fwd0");
      gs.try_rows = true;
      gs.hf_role = fuse ? 1 : 0;
      gs.gm_role = 1;
      for (int k = 0; k < C; ++k) {
        GemmProblem pt = b.fwd_layer(ct[k], 0);
        pt.emit_seg = pt.nseg - 1;   // no tail: rides in the same launch as the dual problems
        set_hf(pt, ct[k], 0, inst_id(k, 0), false);
        gs.gemm.push_back(pt);
        GemmProblem p = b.fwd_layer(co[k], 0);
        Builder::add_seg(p, a->buf("pi_diff"), A, 1, co[k].W(0) + L, a->critic[k].din, 1, A);
        p.emit_seg = p.nseg - 2;
        p.C2 = cf[k].h[0];
        p.ldc2 = a->critic[k].hid[0];
        if (!cf[k].gm.empty()) p.gm_out2 = cf[k].gm[0];
        set_hf(p, co[k], 0, inst_id(k, 1), false);
        set_hf(p, cf[k], 0, inst_id(k, 2), true);
        gs.gemm.push_back(p);
      }
      for (size_t i = 1; i < nh; ++i) {
        Stage &ls = b.gemm_stage("critics.fwd" + std::to_string(i));
        ls.try_rows = true;
        ls.hf_role = fuse ? 1 : 0;
        ls.gm_role = 1;
        for (int k = 0; k < C; ++k) {
          int which = 0;
          for (MlpInst *m : {&ct[k], &co[k], &cf[k]}) {
            GemmProblem p = b.fwd_layer(*m, (int)i);
            set_hf(p, *m, (int)i, inst_id(k, which++), false);
            ls.gemm.push_back(p);
          }
        }
      }
      // The head's finish.  With the hidden layers' parts already formed (head fusion), what is left per instance is
      // cat(s, a) . Wh[:, inputs] + the parts + the bias: a row-per-wave kernel for all instances (k_head_finish)
      // instead of a partial-sum reduction launch plus a head GEMM streaming cat(s, a) through padded tiles.
      bool finished = false;
      if (fuse && C * Q <= 16 && A <= 16 && C <= HEAD_FINISH_MAX_SETS && getenv("FDQL_NO_HEAD_FINISH") == nullptr) {
        HeadFinishArgs ha;
        memset(&ha, 0, sizeof(ha));
        ha.M = M; ha.L = L; ha.A = A; ha.Q = Q; ha.planes = a->hf_planes; ha.ngroups = 2;
        bool okf = true;
        for (int k = 0; k < C; ++k)
          for (MlpInst *m : {&ct[k], &co[k], &cf[k]}) okf = okf && m->in.size() == 2 && m->in[0].width == L && m->in[1].width == A;
        if (okf) {
          HeadFinishGroup &gc = ha.g[0], &gn = ha.g[1];   // group 0: s_cur (online + frozen, one weight set per critic); 1: s_next (targets)
          gc.s = co[0].in[0].ptr; gc.lds = co[0].in[0].ld; gc.nsets = C; gc.nvar = 2; gc.ldw = a->critic[0].head_ld();
          gc.a[0] = co[0].in[1].ptr; gc.lda[0] = co[0].in[1].ld; gc.out[0] = a->buf("q_pred"); gc.ldo[0] = Nq;
          gc.a[1] = cf[0].in[1].ptr; gc.lda[1] = cf[0].in[1].ld; gc.out[1] = a->buf("q_frozen"); gc.ldo[1] = Nq;
          gn.s = ct[0].in[0].ptr; gn.lds = ct[0].in[0].ld; gn.nsets = C; gn.nvar = 1; gn.ldw = a->critic[0].head_ld();
          gn.a[0] = ct[0].in[1].ptr; gn.lda[0] = ct[0].in[1].ld; gn.out[0] = a->buf("next_z"); gn.ldo[0] = Nq;
          for (int k = 0; k < C; ++k) {
            okf = okf && co[k].HW() == cf[k].HW() && a->critic[k].head_ld() == gc.ldw;   // frozen reads the online weights
            gc.Wh[k] = co[k].HW(); gc.bias[k] = co[k].HB();
            gn.Wh[k] = ct[k].HW(); gn.bias[k] = ct[k].HB();
            gc.parts[k][0] = hf_sum + (long long)inst_id(k, 1) * MQ;
            gc.parts[k][1] = hf_sum + (long long)inst_id(k, 2) * MQ;
            gn.parts[k][0] = hf_sum + (long long)inst_id(k, 0) * MQ;
          }
        }
        // few rows (temporal_len 2): the plane sum inside the finish (its 16-row waves add the planes as they read them) instead
        // of a reduction launch in front of it - one launch less; at many rows the pair of launches is the faster one (config 2:
        // 0.0105 + 0.0153 ms against 0.0278 ms folded).  FDQL_HEAD_SUM_LAUNCH=1 / =0 forces either form.
        const char *hsl = getenv("FDQL_HEAD_SUM_LAUNCH");
        const bool sum_in_finish = okf && (hsl ? hsl[0] == '0' : M <= 4096);
        if (sum_in_finish) {
          ha.sum_planes = 1;
          const long long inst_stride = (long long)a->hf_planes * MQ;
          for (int k = 0; k < C; ++k) {
            ha.g[0].parts[k][0] = hf_parts + inst_id(k, 1) * inst_stride;
            ha.g[0].parts[k][1] = hf_parts + inst_id(k, 2) * inst_stride;
            ha.g[1].parts[k][0] = hf_parts + inst_id(k, 0) * inst_stride;
          }
        }
        if (okf) {
          const int ninst = 3 * C, planes = a->hf_planes;
          if (!sum_in_finish)
            b.func_stage("critics.head_sum", [=](hipStream_t s) { return reduce_partials_batched_launch(hf_parts, ninst, planes, MQ, hf_sum, s); }).hf_role = 2;
          // the form the finish takes when every hidden layer's launch sums its own planes (decided in upload_tables): one plane
          // per layer, added while they are read
          HeadFinishArgs hp = ha;
          bool same = true;
          for (size_t i = 1; i < nh; ++i) same = same && a->critic[0].hid[i] == a->critic[0].hid[0];
          hp.sum_planes = 1;
          hp.planes = (int)nh;
          hp.plane_step = ((a->critic[0].hid[0] + 63) / 64) * 2;
          {
            const long long inst_stride = (long long)a->hf_planes * MQ;
            for (int k = 0; k < C; ++k) {
              hp.g[0].parts[k][0] = hf_parts + inst_id(k, 1) * inst_stride;
              hp.g[0].parts[k][1] = hf_parts + inst_id(k, 2) * inst_stride;
              hp.g[1].parts[k][0] = hf_parts + inst_id(k, 0) * inst_stride;
            }
          }
          auto hap = std::make_shared<HeadFinishArgs>(ha);
          Stage &fs = b.func_stage("critics.head", [=](hipStream_t s) { return head_finish_launch(*hap, s); });
          fs.hf_role = 3; fs.hfin = hap; fs.hfin_plain = ha; fs.hfin_presum = hp;
          fs.hfin_can_presum = same && getenv("FDQL_NO_HEAD_PRESUM") == nullptr;
          finished = true;
        }
      }
      if (!finished) {
      if (fuse) {
        const int ninst = 3 * C, planes = a->hf_planes;
        b.func_stage("critics.head_sum", [=](hipStream_t s) { return reduce_partials_batched_launch(hf_parts, ninst, planes, MQ, hf_sum, s); });
      }
      Stage &hs = b.gemm_stage("critics.head");
      for (int k = 0; k < C; ++k) {
        int which = 0;
        for (MlpInst *m : {&ct[k], &co[k], &cf[k]}) {
          GemmProblem p = fuse ? b.fwd_head_inputs_only(*m) : b.fwd_head(*m);
          if (fuse) { p.epi = EPI_ADD_REF; p.ref = hf_sum + (long long)inst_id(k, which) * MQ; p.ldref = Q; }
          ++which;
          hs.gemm.push_back(p);
        }
      }
      }
    } else {
      std::vector<MlpInst *> g;
      for (int k = 0; k < C; ++k) { g.push_back(&ct[k]); g.push_back(&co[k]); g.push_back(&cf[k]); }
      fwd_chain(g, "critics");
    }
  }
  // ---- loss
  bool ride_finish = false;
  LossFinishArgs finish_args;
  memset(&finish_args, 0, sizeof(finish_args));
  {
    LossArgs la;
    memset(&la, 0, sizeof(la));
    la.M = M; la.B = B; la.Nq = Nq; la.Nt = a->Nt;
    int G = 8;
    while (G < Nq) G <<= 1;
    if (loss_wave_form(c.distributional, Nq)) G = 64;   // one wave per row, four rows per workgroup (kernels.hip, k_loss_wave)
    la.G = G;
    la.distributional = c.distributional; la.lowerbound = c.use_lowerbound; la.max_entropy = c.use_max_entropy;
    la.gamma = (float)c.gamma; la.target_entropy = -(float)A; la.half_inv_nq = (float)(0.5 / (double)Nq);
    la.st = dst; la.log_alpha = log_alpha;
    la.z_target = a->buf("next_z"); la.q_pred = a->buf("q_pred"); la.z_frozen = a->buf("q_frozen");
    la.logp_next = a->buf("next_log_pi"); la.logp = a->buf("log_pi");
    la.reward = x.reward; la.task_done = x.task_done; la.mc_return = c.use_lowerbound ? x.mc_return : nullptr;
    la.w = a->buf("w"); la.dz = a->buf("dz"); la.dzf = a->buf("dzf"); la.td_target = a->buf("td_target");
    la.q_loss = a->buf("q_loss"); la.pi_loss = a->buf("pi_loss"); la.alpha_loss = a->buf("alpha_loss");
    la.partials = a->buf("loss_partials");
    int nblocks = loss_blocks(M, G);
    // few workgroups (temporal_len 2): the last one to finish also sums the partial rows (kernels.hip, LossFinishArgs) - one
    // launch less; many workgroups would serialise on the arrival counter (round 2: ~80 ns per atomic)
    const bool fuse_finish = nblocks <= 64 && !c.bootstrap_nstep && getenv("FDQL_NO_LOSS_FINISH_FUSE") == nullptr;
    if (fuse_finish) {
      LossFinishArgs fa;
      memset(&fa, 0, sizeof(fa));
      fa.nblocks = nblocks; fa.M = M; fa.Nq = Nq; fa.st = dst; fa.scalars = a->buf("scalars");
      fa.dlog_alpha = a->buf("slabs") + a->log_alpha_off; fa.lr = c.lr; fa.b1 = c.beta1; fa.b2 = c.beta2;
      LossFinishArgs *fdev = reinterpret_cast<LossFinishArgs *>(a->buf("loss_fin_args"));
      FDQL_HIP(hipMemcpy(fdev, &fa, sizeof(fa), hipMemcpyHostToDevice));
      la.fin = fdev;
    }
    b.func_stage("loss", [=](hipStream_t s) { return loss_launch(la, s); });
    if (c.bootstrap_nstep) {   // soft_actor_critic.py:102-132: its loss value rides as one more partial row
      BootArgs ba;
      memset(&ba, 0, sizeof(ba));
      ba.T = a->T; ba.B = B; ba.Nq = Nq; ba.gamma = (float)c.gamma;
      ba.scale = (float)(1.0 / ((double)B * Nq * (c.world_size > 0 ? c.world_size : 1) * a->T));
      ba.reward = x.reward; ba.task_done = x.task_done; ba.contig = a->buf("is_contiguous");
      ba.td_target = a->buf("td_target"); ba.q_pred = a->buf("q_pred"); ba.dz = a->buf("dz");
      ba.partial_row = a->buf("loss_partials") + (int64_t)nblocks * LOSS_NPART;
      b.func_stage("boot_lowerbound", [=](hipStream_t s) { return boot_lowerbound_launch(ba, s); });
      ++nblocks;
    }
    float *scal = a->buf("scalars");
    float *dla = a->buf("slabs") + a->log_alpha_off;
    const float *parts = la.partials;
    // also the Adam bias corrections of the step about to be applied (torch.optim.Adam's Python floats)
    const double lr = c.lr, b1 = c.beta1, b2 = c.beta2;
    // single-process plans: the finish rides in the policy-backward launch (its next consumer is the optimiser); a two-bucket
    // plan needs d log_alpha in the early bucket, before that launch
    ride_finish = !fuse_finish && !a->bucketed() && getenv("FDQL_NO_LOSS_FINISH_RIDE") == nullptr;
    finish_args.nblocks = nblocks; finish_args.M = M; finish_args.Nq = Nq; finish_args.st = dst; finish_args.scalars = scal;
    finish_args.dlog_alpha = dla; finish_args.lr = lr; finish_args.b1 = b1; finish_args.b2 = b2;
    if (!fuse_finish && !ride_finish)
      b.func_stage("loss_finish", [=](hipStream_t s) { return loss_finish_launch(parts, nblocks, M, Nq, dst, scal, dla, lr, b1, b2, s); });
  }
  // ---- critic backward (online: wgrad + d state; frozen: d pi only)
  {
    const size_t nh = a->critic[0].hid.size();
    // Two hidden layers under a narrow head: the last layer's gradient (k_head_dgrad: a rank-Q outer product gated by
    // LeakyReLU') can be formed inside the loader of the row-block launch that consumes it (rowgemm.hip, FUSE) instead of
    // making a 2 x 131 MB round trip through HBM in a launch of its own.  Only when that launch takes the problems.
    bool fused1 = false;
    if (nh == 2 && Builder::narrow_head_last(a->critic[0], 1) && getenv("FDQL_NO_FUSE_DPRE1") == nullptr) {
      std::vector<GemmProblem> cand;
      for (int k = 0; k < C; ++k) {
        int which = 0;
        for (MlpInst *m : {&co[k], &cf[k]}) {
          GemmProblem p = b.bwd_dpre(*m, 0, a->buf(which == 0 ? "dz" : "dzf") + k * Q, Nq);
          p.fz_h = m->h[1];
          if (m->gm.size() > 1) p.gm_fz = m->gm[1];
          p.fz_w = m->HW() + b.head_col_of_hidden(*m->d, 1);
          p.fz_ldw = m->d->head_ld();
          p.fz_out = m->dpre[1];
          p.fz_colsum = m->dpre_cs[1];
          p.fz_discard = which == 1 && getenv("FDQL_KEEP_FROZEN_DPRE1") == nullptr;   // frozen copy: its dpre1 feeds nothing but this GEMM
          cand.push_back(p);
          ++which;
        }
      }
      RowsLaunch rl;
      if (M % RG_BM == 0 && rows_launch_of(a, cand, rl)) {
        Stage &gs = b.gemm_stage("critics.dpre1+0");
        gs.try_rows = true;
        gs.gm_role = 2;
        gs.gemm = cand;
        fused1 = true;
        if (rl.ws)   // the weight-stationary launch leaves one partial row of column sums per workgroup
          for (int k = 0; k < C; ++k)
            for (MlpInst *m : {&co[k], &cf[k]}) m->dpre_cs_rows[0] = m->dpre_cs_rows[1] = wstat_colsum_rows(rl.wa);
      }
    }
    // few rows (temporal_len 2): the last layer's rank-Q product as GEMM problems on the small-batch kernel (one load round per
    // workgroup) instead of the streaming kernel's row loop: 0.022 -> 0.011 ms at 256 rows
    const bool head_dgrad_as_gemm = a->small_max_tiles > 0 && gemm_dense_shape() == GEMM_64x64 && gemm_variant() == GEMM_DEFAULT_VARIANT &&
                                    (long long)2 * C * ((M + 63) / 64) * ((a->critic[0].hid.empty() ? 0 : a->critic[0].hid.back()) + 63) / 64 <= 4LL * a->small_max_tiles;
    // Will the forward launches leave gate masks?  (the question upload_tables answers for good: every forward layer of the
    // critics as one weight-stationary launch; a stage that relies on the answer is checked there - Stage::needs_masks)
    bool masks_planned = getenv("FDQL_NO_GATE_MASKS") == nullptr;
    for (const Stage &fs : a->stages) {
      if (fs.gm_role != 1 || !masks_planned) continue;
      RowsLaunch rl;
      masks_planned = fs.gemm.size() > 1 && rows_launch_of(a, fs.gemm, rl) && rl.ws;
    }
    for (int i = (int)nh - 1; i >= 0 && !fused1; --i) {
      // the last hidden layer under a head of up to 32 outputs, gated by the masks (config 4: 25 quantiles; the rank-25 product was
      // a tile launch at 25 TF that read h again): FDQL_NO_HEAD_DGRAD_MASKED keeps the GEMM stage
      if (i == (int)nh - 1 && masks_planned && Q > HEAD_DGRAD_MAXQ && Q <= HDM_MAXQ && a->critic[0].hid[i] == 256 && M % 32 == 0 && 2 * C <= HDM_MAX_INST &&
          !co[0].gm.empty() && getenv("FDQL_NO_HEAD_DGRAD_MASKED") == nullptr) {
        HeadDgradMaskedArgs ha;
        memset(&ha, 0, sizeof(ha));
        ha.M = M; ha.Q = Q; ha.ninst = 2 * C; ha.lddy = Nq; ha.ldw = a->critic[0].head_ld();
        for (int k = 0; k < C; ++k) {
          int w = 0;
          for (MlpInst *m : {&co[k], &cf[k]}) {
            const int n = 2 * k + w;
            ha.dY[n] = a->buf(w == 0 ? "dz" : "dzf") + k * Q;
            ha.Wh[n] = m->HW() + b.head_col_of_hidden(*m->d, i);
            ha.gm[n] = m->gm[i];
            ha.dpre[n] = m->dpre[i];
            ha.colsum[n] = m->dpre_cs[i];
            ++w;
          }
        }
        Stage &fs = b.func_stage("critics.dpre" + std::to_string(i) + "(masked)", [=](hipStream_t s) { return head_dgrad_masked_launch(ha, s); });
        fs.needs_masks = true;
        // the same product as a GEMM stage gated by h itself (same outputs, same per-64-row column sums): switched on by
        // upload_tables instead of the masked launch when the forward launches of the final plan turn out not to write masks
        Stage &fb = b.gemm_stage("critics.dpre" + std::to_string(i) + "(unmasked)");
        fb.masks_fallback = true;
        fb.off = true;
        for (int k = 0; k < C; ++k) {
          fb.gemm.push_back(b.bwd_dpre(co[k], i, a->buf("dz") + k * Q, Nq));
          fb.gemm.push_back(b.bwd_dpre(cf[k], i, a->buf("dzf") + k * Q, Nq));
        }
        continue;
      }
      if (Builder::narrow_head_last(a->critic[0], i) && !head_dgrad_as_gemm) {
        Stage st;
        st.kind = ST_HEAD_DGRAD; st.name = "critics.dpre" + std::to_string(i);
        for (int k = 0; k < C; ++k) {
          st.hdg.push_back(b.bwd_dpre_head(co[k], i, a->buf("dz") + k * Q, Nq));
          st.hdg.push_back(b.bwd_dpre_head(cf[k], i, a->buf("dzf") + k * Q, Nq));
        }
        a->stages.push_back(st);
        continue;
      }
      Stage &gs = b.gemm_stage("critics.dpre" + std::to_string(i));
      gs.try_rows = true;
      gs.gm_role = 2;
      for (int k = 0; k < C; ++k) {
        gs.gemm.push_back(b.bwd_dpre(co[k], i, a->buf("dz") + k * Q, Nq));
        gs.gemm.push_back(b.bwd_dpre(cf[k], i, a->buf("dzf") + k * Q, Nq));
      }
      {
        RowsLaunch rl;
        std::vector<GemmProblem> grp = gs.gemm;
        if (rows_launch_of(a, grp, rl) && rl.ws)
          for (int k = 0; k < C; ++k)
            for (MlpInst *m : {&co[k], &cf[k]}) m->dpre_cs_rows[i] = wstat_colsum_rows(rl.wa);
      }
    }
    // d pi: input-grad of each frozen critic's action columns as its own narrow (128x32) problem
    // -> C partials [C][M][A], summed in fixed order by the policy backward kernel
    {
      Stage &gs = b.gemm_stage("dpi");
      for (int k = 0; k < C; ++k) {
        GemmProblem p = Builder::new_gemm(M, A, a->buf("dpi_part") + (int64_t)k * M * A, A);
        b.input_grad_segs(cf[k], a->buf("dzf") + k * Q, Nq, L, p);
        gs.gemm.push_back(p);
      }
    }
  }
  // ---- data-parallel plans: the critics' weight gradients now, and their slab sum, so that the all-reduce of the arena
  // range [crit_begin, n_train) (critics + log_alpha: 2/3 of the arena at config 2) can run beside everything below
  const bool bucketed = a->bucketed();
  a->grad_bucket = bucketed ? a->crit_begin : a->n_train;
  size_t first_rest_stage = 0;
  if (bucketed) {
    Stage cn, cws;
    cn.kind = ST_GEMM; cn.name = "wgrad.critics.narrow";
    cws.kind = ST_SKINNY_WGRAD; cws.name = "colsums.critics";
    for (int k = 0; k < C; ++k) b.wgrads(co[k], a->buf("dz") + k * Q, Nq, nullptr, cn, cn, cws);
    b.flush_wgrad_stat("wgrad.critics", cn);
    Stage cnw;
    cnw.kind = ST_SKINNY_WGRAD; cnw.stream = true; cnw.name = "wgrad.critics.stream";
    b.take_stream_wgrads(cn, cnw, cws);
    b.colsums_into_stream(cws, cnw);
    if (!cn.gemm.empty()) a->stages.push_back(cn);
    if (!cnw.swg.empty()) a->stages.push_back(cnw);
    if (!cws.swg.empty()) a->stages.push_back(cws);
    const float *slabs = a->buf("slabs");
    float *grads = a->grads;
    const int S = a->nsplit;
    const long long P = a->n_train, first = a->crit_begin;
    b.func_stage("reduce_slabs.critics", [=](hipStream_t s) { return reduce_slabs_range_launch(slabs, S, P, first, P - first, grads, s); }).when = 1;
    first_rest_stage = a->stages.size();
  }
  // ---- policy backward
  bool fuse_pd = false;
  {
    const float *lo = ao.out, *nz = a->buf("noise_actor"), *pi = a->buf("pi"), *dpi = a->buf("dpi_part"), *w = a->buf("w");
    float *dlo = a->buf("dlogits"), *dpi_sum = a->buf("dpi");
    const int disc = c.discrete;
    const float *lparts = a->buf("loss_partials");
    const LossFinishArgs fa = finish_args;
    const bool ride = ride_finish;
    // the actor's last hidden layer under its narrow head: its pre-activation gradient in the same launch (FDQL_NO_POLICY_DPRE_FUSE:
    // a GEMM stage of its own, as before round 4)
    const int last = (int)a->actor.hid.size() - 1;
    fuse_pd = last >= 0 && policy_bwd_dpre_takes(disc, A, a->actor.hid[last]) && a->actor.dout == 2 * A && ao.dpre[last] &&
              getenv("FDQL_NO_POLICY_DPRE_FUSE") == nullptr;
    if (fuse_pd) {
      const float *Wh = ao.HW() + b.head_col_of_hidden(*ao.d, last), *h = ao.h[last];
      const int ldw = ao.d->head_ld();
      float *dpre = ao.dpre[last], *cs = ao.dpre_cs[last];
      b.func_stage("policy_bwd+actor.dpre" + std::to_string(last), [=](hipStream_t s) {
        return policy_bwd_dpre_launch(lo, nz, pi, dpi, C, dpi_sum, w, dst, M, A, dlo, Wh, ldw, h, dpre, cs, s, lparts, ride ? &fa : nullptr);
      });
    } else {
      b.func_stage("policy_bwd", [=](hipStream_t s) {
        return policy_bwd_launch(lo, nz, pi, dpi, C, dpi_sum, w, dst, M, A, dlo, disc, s, lparts, ride ? &fa : nullptr);
      });
    }
  }
  // ---- actor backward
  // The weight gradients of a network only need that network's own dpre/dY, so instead of one big
  // wgrad stage at the end they ride along with the small single-network dgrad launches that follow
  // (same tile shape -> same launch): those launches have only ~400 workgroups of their own.
  std::vector<size_t> hosts;  // stage indices of the dense dgrad launches after the critics' backward
  for (int i = (int)a->actor.hid.size() - 1; i >= 0; --i) {
    if (fuse_pd && i == (int)a->actor.hid.size() - 1) continue;   // formed by the policy backward's launch
    // (the rank-2A product on the streaming kernel k_head_dgrad - 12 broadcast LDS reads per element - measured 0.050 ms against
    // 0.015 ms for this K = 12 problem on MFMA tiles at config 2: it stays a GEMM problem)
    Stage &gs = b.gemm_stage("actor.dpre" + std::to_string(i));
    gs.gemm.push_back(b.bwd_dpre(ao, i, a->buf("dlogits"), a->actor.dout));
    hosts.push_back(a->stages.size() - 1);
  }
  // ---- d state = sum over online critics and the actor
  {
    Stage &gs = b.gemm_stage("dstate");
    if (a->dstate_split) {
      float *parts = a->buf("dstate.parts");
      const long long ML = (long long)M * L;
      gs.try_rows = true;   // the critics' shares: weight-stationary plain dgrad form when there are enough rows
      for (int k = 0; k <= C; ++k) {
        GemmProblem p = Builder::new_gemm(M, L, parts + k * ML, L);
        if (k < C) b.input_grad_segs(co[k], a->buf("dz") + k * Q, Nq, 0, p);
        else b.input_grad_segs(ao, a->buf("dlogits"), a->actor.dout, 0, p);
        gs.gemm.push_back(p);
      }
    } else {
      GemmProblem p = Builder::new_gemm(M, L, a->buf("dstate"), L);
      for (int k = 0; k < C; ++k) b.input_grad_segs(co[k], a->buf("dz") + k * Q, Nq, 0, p);
      b.input_grad_segs(ao, a->buf("dlogits"), a->actor.dout, 0, p);
      p.colsum = a->buf("cs.dstate");
      gs.gemm.push_back(p);
    }
    hosts.push_back(a->stages.size() - 1);
  }
  const size_t idx_dstate = a->stages.size() - 1;
  // The sum of the shares inside the launch that consumes it first - the joiner's top hidden layer's dgrad on the row-block dgrad
  // kernel (rowdgrad.h, RowDgradArgs::sum_*): each 64-row workgroup adds its rows of the C + 1 shares while it stages them, writes
  // d state and its column sums; the summing launch (and one write + read of d state) goes away.  Asked here with the same
  // deterministic questions upload_tables asks, because the answer changes the column sums' row count.
  bool fold_dsum = false;
  // ... and with one hidden layer in the joiner and in the encoder the three launches behind d state become one (k_rowdgrad_chain):
  // its stages may have more blocks than one round of workgroups (784 at config 4, B = 1024: 0.379 ms against 0.397 for the summing
  // launch + three tile launches; a lone row-block dgrad launch of that size loses against the tile kernel)
  const bool chain_planned = a->joiner.hid.size() == 1 && a->enc_obs.hid.size() == 1 && c.enc_features == L && getenv("FDQL_NO_ROWDGRAD_CHAIN") == nullptr;
  const int rd_max_blocks = chain_planned ? std::max(a->rowdgrad_max_blocks, 1024) : a->rowdgrad_max_blocks;
  if (a->dstate_split && L % 4 == 0 && !gru && !a->joiner.hid.empty()) {
    MlpInst jq = jo;
    jq.rows = M;
    GemmProblem p = b.bwd_dpre(jq, (int)a->joiner.hid.size() - 1, a->buf("dstate"), L);
    RowDgradArgs tmp;
    fold_dsum = M / RD_BM >= a->rowdgrad_min_blocks && M / RD_BM <= rd_max_blocks && (M >= a->rowdot_min_rows ? p.N > 16 : true) &&
                rowdgrad_from_problem(p, tmp) &&
                rowdgrad_fold_sum(tmp, a->buf("dstate.parts"), C + 1, (long long)M * L, a->buf("dstate"), a->buf("cs.dstate"));
  }
  if (a->dstate_split && !fold_dsum) {
    const float *parts = a->buf("dstate.parts");
    float *dsum = a->buf("dstate");
    const long long ML = (long long)M * L;
    const int np = C + 1;
    if (L % 4 == 0) {   // the sum also leaves the column sums the joiner head's bias gradient is reduced from
      float *csd = a->buf("cs.dstate");
      b.func_stage("dstate.sum", [=](hipStream_t s) { return sum_parts_colsum_launch(parts, np, M, L, dsum, csd, s); });
    } else {
      b.func_stage("dstate.sum", [=](hipStream_t s) { return reduce_partials_launch(parts, np, ML, dsum, s); });
    }
  }
  // column sums of d state: per 64 rows from the one-problem GEMM or the folded sum, per 32 rows from the summing launch, else
  // straight from d state
  const float *cs_dstate = (a->dstate_split && L % 4) ? nullptr : a->buf("cs.dstate");
  const int cs_dstate_rows = a->dstate_split ? (fold_dsum ? (M + 63) / 64 : (M + 31) / 32) : 0;
  // ---- encoder backward over the M rows that carry gradient (next-only rows get none)
  MlpInst jb = jo, eb = eo;
  jb.rows = M; eb.rows = M;
  for (int i = (int)a->joiner.hid.size() - 1; i >= 0; --i) {
    Stage &gs = b.gemm_stage((fold_dsum && i == (int)a->joiner.hid.size() - 1 ? "dstate.sum+joiner.dpre" : "joiner.dpre") + std::to_string(i));
    gs.gemm.push_back(b.bwd_dpre(jb, i, a->buf("dstate"), L));
    if (fold_dsum && i == (int)a->joiner.hid.size() - 1) {
      gs.fold_sum = true; gs.fold_parts = a->buf("dstate.parts"); gs.fold_n = C + 1; gs.fold_stride = (long long)M * L;
      gs.fold_out = a->buf("dstate"); gs.fold_cs = a->buf("cs.dstate");
      gs.rd_max_blocks = rd_max_blocks;
    }
    hosts.push_back(a->stages.size() - 1);
  }
  if (gru) {
    // back-propagation through time over the T-1 rows that carry gradient (h_{T-1} only feeds no_grad targets):
    //   dh_t = d state[t] + dh_{t+1} * z_{t+1} + d gh_{t+1} W_hh
    const int L3 = 3 * L, T = a->T;
    float *dgi = a->buf("gru.dgi"), *dgh = a->buf("gru.dgh"), *dhw = a->buf("gru.dhw");
    float *dhz[2] = {a->buf("gru.dhz0"), a->buf("gru.dhz1")};
    const float *gi = a->buf("gru.gi"), *gh = a->buf("gru.gh"), *hprev = a->buf("gru.hprev"), *dstate = a->buf("dstate");
    const float *whh = params + a->gru_whh;
    const bool scan = gru_scan_takes(B, L);
    if (scan) {
      float *dh_init = a->buf("gru.dh_init");
      GruScanArgs ga;
      memset(&ga, 0, sizeof(ga));
      ga.T = T; ga.B = B; ga.L = L; ga.W = a->buf("gru.wpack_b"); ga.gi = gi; ga.gh = const_cast<float *>(gh); ga.hprev = const_cast<float *>(hprev);
      ga.dstate = dstate; ga.dgi = dgi; ga.dgh = dgh; ga.dh_init = c.gru_state_mode == 2 ? dh_init : nullptr;
      b.func_stage("gru.scan_bwd", [=](hipStream_t s) { return gru_scan_bwd_launch(ga, s); });
      if (c.gru_state_mode == 2) {
        float *out = a->buf("slabs") + a->gru_h0;
        b.func_stage("gru.dh0", [=](hipStream_t s) { return gru_dh0_launch(dh_init, nullptr, 0, B, L, out, s); });
      }
    }
    for (int t = T - 2; t >= 0 && !scan; --t) {
      const int64_t r3 = (int64_t)t * B * L3, r1 = (int64_t)t * B * L;
      const bool last = t == T - 2;
      const float *ca = last ? nullptr : dhz[(t + 1) & 1], *cb = last ? nullptr : dhw;
      float *out_z = dhz[t & 1];
      b.func_stage("gru.cell_bwd", [=](hipStream_t s) {
        return gru_cell_bwd_launch(dstate + r1, ca, cb, GRU_KSPLIT_BWD, gi + r3, gh + r3, hprev + r1, dgi + r3, dgh + r3, out_z,
                                   B, L, s);
      });
      Stage &gs = b.gemm_stage("gru.dh_prev");
      GemmProblem p = Builder::new_gemm(B, L, dhw, L);
      Builder::add_seg(p, dgh + r3, L3, 1, whh, L, 0, L3);
      p.ksplit = GRU_KSPLIT_BWD;
      p.split_stride = (long long)B * L;
      gs.gemm.push_back(p);
    }
    if (c.gru_state_mode == 2 && !scan) {   // learned start state: d hidden_state = sum_b d h_{-1}
      float *out = a->buf("slabs") + a->gru_h0;
      const float *za = dhz[0];
      b.func_stage("gru.dh0", [=](hipStream_t s) { return gru_dh0_launch(za, dhw, GRU_KSPLIT_BWD, B, L, out, s); });
    }
  }
  {
    Stage &gs = b.gemm_stage("denc");
    GemmProblem p = Builder::new_gemm(M, c.enc_features, a->buf("denc"), c.enc_features);
    if (gru) Builder::add_seg(p, a->buf("gru.dgi"), 3 * L, 1, params + a->gru_wih, c.enc_features, 0, 3 * L);
    else b.input_grad_segs(jb, a->buf("dstate"), L, 0, p);
    p.colsum = a->buf("cs.denc");
    gs.gemm.push_back(p);
    if (fold_dsum) gs.rd_max_blocks = rd_max_blocks;
    hosts.push_back(a->stages.size() - 1);
  }
  const size_t idx_denc = a->stages.size() - 1;
  for (int i = (int)a->enc_obs.hid.size() - 1; i >= 0; --i) {
    Stage &gs = b.gemm_stage("enc_obs.dpre" + std::to_string(i));
    gs.gemm.push_back(b.bwd_dpre(eb, i, a->buf("denc"), c.enc_features));
    if (fold_dsum) gs.rd_max_blocks = rd_max_blocks;
    hosts.push_back(a->stages.size() - 1);
  }
  // ---- pixel encoder backward (M images): d features -> per layer [dW, db], d col -> col2im -> previous layer
  if (nconv) {
    const int last = nconv - 1, F2 = a->conv_feat;
    {
      Stage &gs = b.gemm_stage("conv.dfeat");
      GemmProblem p = Builder::new_gemm(M, F2, a->buf("conv" + std::to_string(last) + ".dpre"), F2);
      b.input_grad_segs(eb, a->buf("denc"), c.enc_features, conv_col0, p);
      p.epi = EPI_LRELU_GRAD;
      p.ref = a->buf("conv" + std::to_string(last) + ".out");
      p.ldref = F2;
      gs.gemm.push_back(p);
    }
    for (int i = last; i > 0; --i) {
      const fdql_agent::ConvLayer &Lc = a->conv[i];
      const ConvGeom g = Lc.g;
      const int K = g.C * g.k * g.k;
      const long long rows = (long long)M * g.OH * g.OW;
      if (Lc.fast_dgrad) {   // gather-form implicit GEMM: no d col matrix, no col2im
        ConvDgradArgs da;
        da.dpre = a->buf("conv" + std::to_string(i) + ".dpre"); da.W = params + Lc.w_off;
        da.act_prev = a->buf("conv" + std::to_string(i - 1) + ".out"); da.dprev = a->buf("conv" + std::to_string(i - 1) + ".dpre");
        da.nimg = M; da.g = g; da.cout = Lc.cout;
        Stage &ds = b.func_stage("conv.dgrad" + std::to_string(i), [=](hipStream_t s) { return conv_dgrad_launch(da, s); });
        ds.mfma = true;
        ds.flops = 2.0 * (double)rows * K * Lc.cout;
        ds.bytes = 4.0 * ((double)rows * Lc.cout + 2.0 * (double)M * g.C * g.H * g.W);
        continue;
      }
      float *dcol = a->buf("conv" + std::to_string(i) + ".dcol");
      Stage &gs = b.gemm_stage("conv.dcol" + std::to_string(i));
      GemmProblem p = Builder::new_gemm((int)rows, K, dcol, K);
      Builder::add_seg(p, a->buf("conv" + std::to_string(i) + ".dpre"), Lc.cout, 1, params + Lc.w_off, K, 0, Lc.cout);
      gs.gemm.push_back(p);
      const float *act_prev = a->buf("conv" + std::to_string(i - 1) + ".out");
      float *dprev = a->buf("conv" + std::to_string(i - 1) + ".dpre");
      const long long nimg = M;
      b.func_stage("conv.col2im", [=](hipStream_t s) { return col2im_mask_launch(dcol, act_prev, nimg, g, dprev, s); });
    }
  }
  // ---- weight gradients (K-split slabs) + column sums
  {
    Stage tail, ws;
    std::vector<std::function<hipError_t(hipStream_t)>> conv_post;   // conv weight / bias partials -> slab 0
    tail.kind = ST_GEMM; tail.name = "wgrad.enc";
    ws.kind = ST_SKINNY_WGRAD; ws.name = "colsums";
    // critics: spread over the dgrad launches that follow their backward (they are ready by then)
    if (!bucketed)
      for (int k = 0; k < C; ++k) b.wgrads(co[k], a->buf("dz") + k * Q, Nq, nullptr, a->stages[hosts[k % hosts.size()]], tail, ws);
    // actor: needs d logits / its dpre -> from the d state launch on
    b.wgrads(ao, a->buf("dlogits"), a->actor.dout, nullptr, a->stages[idx_dstate], tail, ws);
    // joiner: needs d state and its dpre -> the d enc launch; encoder MLP: needs d enc and its dpre -> the tail
    if (!gru) {
      b.wgrads(jb, a->buf("dstate"), L, cs_dstate, a->stages[idx_denc], tail, ws, cs_dstate_rows);
    } else {   // GRU weights: dW_hh = d gh^T h_prev, dW_ih = d gi^T e, biases = column sums (all over the M rows)
      const int L3 = 3 * L, F = c.enc_features;
      float *slab = a->buf("slabs");
      Stage &host = a->stages[idx_denc];
      b.wgrad_gemm(M, a->buf("gru.dgh"), L3, L3, a->buf("gru.hprev"), L, L, slab + a->gru_whh, L, host, tail);
      b.wgrad_gemm(M, a->buf("gru.dgi"), L3, L3, eo.out, F, F, slab + a->gru_wih, F, host, tail);
      b.wgrad_bias(M, a->buf("gru.dgh"), L3, L3, nullptr, slab + a->gru_bhh, ws);
      b.wgrad_bias(M, a->buf("gru.dgi"), L3, L3, nullptr, slab + a->gru_bih, ws);
    }
    b.wgrads(eb, a->buf("denc"), c.enc_features, a->buf("cs.denc"), tail, tail, ws);
    for (int i = 0; i < nconv; ++i) {   // conv weights: dW = d pre^T col over the M*OH*OW rows, bias = column sums
      const fdql_agent::ConvLayer &Lc = a->conv[i];
      const int K = Lc.g.C * Lc.g.k * Lc.g.k;
      const int R = (int)((long long)M * Lc.g.OH * Lc.g.OW);
      float *slab = a->buf("slabs");
      const float *dpre = a->buf("conv" + std::to_string(i) + ".dpre");
      if (Lc.fast_wgrad) {   // output-stationary implicit GEMM: (dW, db) partials per slab, one reduction into slab 0
        ConvWgradArgs wa;
        wa.in.base = i == 0 ? (const void *)x.obs_2d_u8 : (const void *)a->buf("conv" + std::to_string(i - 1) + ".out");
        wa.in.u8 = i == 0; wa.in.slots = i == 0 ? x.obs_2d_slots : nullptr;
        wa.dpre = dpre; wa.wpart = a->buf("conv" + std::to_string(i) + ".wpart"); wa.nimg = M; wa.g = Lc.g; wa.cout = Lc.cout;
        const int nslab = conv_wgrad_slabs(Lc.g, Lc.cout, i == 0, M);
        const long long nw = (long long)Lc.cout * K + Lc.cout;
        FDQL_REQUIRE(nslab > 0 && a->named.at("conv" + std::to_string(i) + ".wpart").second >= nslab * nw, "conv layer %d: weight-gradient slabs", i);
        Stage &wst = b.func_stage("conv.wgrad" + std::to_string(i), [=](hipStream_t s) { return conv_wgrad_launch(wa, s); });
        wst.mfma = true;
        wst.flops = 2.0 * (double)R * K * Lc.cout;
        wst.bytes = (i == 0 ? 1.0 : 4.0) * (double)M * Lc.g.C * Lc.g.H * Lc.g.W + 4.0 * (double)R * Lc.cout;
        const float *wpart = wa.wpart;
        float *wdst = slab + Lc.w_off;   // (the bias follows its weights in the arena: checked in layout)
        conv_post.push_back([=](hipStream_t s) { return reduce_partials_launch(wpart, nslab, nw, wdst, s); });
        continue;
      }
      float *wpart = a->buf("conv" + std::to_string(i) + ".wpart"), *bpart = a->buf("conv" + std::to_string(i) + ".bpart");
      const int S2 = conv_wsplit(R);
      {
        GemmProblem p = Builder::new_gemm(Lc.cout, K, wpart, K);
        Builder::add_seg(p, dpre, Lc.cout, 0, a->buf("conv" + std::to_string(i) + ".col"), K, 0, R);
        p.ksplit = S2;
        p.split_stride = (long long)Lc.cout * K;
        tail.gemm.push_back(p);
      }
      float *wdst = slab + Lc.w_off, *bdst = slab + Lc.b_off;
      const long long nw = (long long)Lc.cout * K;
      const int cout = Lc.cout, nblk = colsum_tall_blocks(R);
      conv_post.push_back([=](hipStream_t s) {
        hipError_t e = reduce_partials_launch(wpart, S2, nw, wdst, s);
        if (e == hipSuccess) e = colsum_tall_launch(dpre, R, cout, cout, bpart, s);
        const int nblk2 = colsum_tall_blocks(nblk);     // second level: the [nblk, cout] partials are tall again
        float *bpart2 = bpart + (long long)nblk * cout;
        if (e == hipSuccess) e = colsum_tall_launch(bpart, nblk, cout, cout, bpart2, s);
        if (e == hipSuccess) e = reduce_partials_launch(bpart2, nblk2, cout, bdst, s);
        return e;
      });
    }
    b.flush_wgrad_stat("wgrad.dense", tail);
    Stage nws;
    nws.kind = ST_SKINNY_WGRAD; nws.stream = true; nws.name = "wgrad.stream";
    b.take_stream_wgrads(tail, nws, ws);
    b.colsums_into_stream(ws, nws);
    if (!tail.gemm.empty()) a->stages.push_back(tail);
    if (!nws.swg.empty()) a->stages.push_back(nws);
    if (!ws.swg.empty()) a->stages.push_back(ws);
    if (!conv_post.empty())
      b.func_stage("conv.wgrad_reduce", [=](hipStream_t s) {
        for (const auto &f : conv_post) { hipError_t e = f(s); if (e != hipSuccess) return e; }
        return hipSuccess;
      });
  }
  {
    const float *slabs = a->buf("slabs");
    float *grads = a->grads;
    const int S = a->nsplit;
    const long long P = a->n_train;
    // a split call (data-parallel: the all-reduce sits between the phases) sums the slabs into grads here; the
    // single-process step forms the sum inside k_adam_polyak
    const long long count = a->grad_bucket;   // a bucketed plan has summed [grad_bucket, P) already
    b.func_stage("reduce_slabs", [=](hipStream_t s) { return reduce_slabs_range_launch(slabs, S, P, 0, count, grads, s); }).when = 1;
  }
  // ---- Adam + polyak (+ frozen copy)
  {
    AdamArgs ad;
    memset(&ad, 0, sizeof(ad));
    ad.n = a->n_train; ad.params = a->params; ad.m = a->adam_m; ad.v = a->adam_v; ad.grads = a->grads;
    ad.grad_scale = 1.0f;
    ad.one_minus_b1 = (float)(1.0 - c.beta1); ad.b2 = (float)c.beta2; ad.one_minus_b2 = (float)(1.0 - c.beta2);
    ad.eps = (float)c.adam_eps; ad.st = dst; ad.targets = a->targets; ad.tgt_begin = a->tgt_begin; ad.tgt_end = a->tgt_end;
    ad.tau = (float)c.tau; ad.one_minus_tau = (float)(1.0 - c.tau); ad.hard = c.hard_updates;
    ad.frozen = c.keep_frozen_copy ? a->frozen : nullptr; ad.frozen_begin = a->crit_begin; ad.frozen_end = a->crit_end;
    b.func_stage("adam_polyak", [=](hipStream_t s) { return adam_launch(ad, s); }, FDQL_PHASE_APPLY).when = 1;
    AdamArgs af = ad;
    af.slabs = a->buf("slabs"); af.nslab = a->nsplit; af.grads_out = a->grads;
    b.func_stage("adam_polyak", [=](hipStream_t s) { return adam_launch(af, s); }, FDQL_PHASE_APPLY).when = 2;
  }
  for (size_t i = 0; i < a->stages.size(); ++i) a->stages[i].gpart = (bucketed && i < first_rest_stage) ? 0 : 1;
  int rc = upload_tables(a);
  if (rc) return rc;
  a->plan_ready = true;
  return 0;
}

hipError_t run_stage(fdql_agent *a, Stage &s, hipStream_t stream) {
  switch (s.kind) {
    case ST_GEMM:
      for (const RowsLaunch &rl : s.rows) {
        hipError_t e = rl.launch(stream);
        if (e != hipSuccess) return e;
      }
      for (int sh = 0; sh < GEMM_NSHAPES; ++sh) {
        const GemmSub &sub = s.sub[sh];
        if (sub.blocks <= 0) continue;
        hipError_t e = gemm_launch((const GemmProblem *)sub.dev, (int)sub.probs.size(), sub.blocks, sh, stream);
        if (e != hipSuccess) return e;
      }
      return hipSuccess;
    case ST_SKINNY_WGRAD:
      if (s.stream) return stream_wgrad_launch(s.swg.data(), (const SkinnyWgradProblem *)s.dev, (int)s.swg.size(), s.blocks, stream);
      return skinny_wgrad_launch_host(s.swg.data(), (const SkinnyWgradProblem *)s.dev, (int)s.swg.size(), s.blocks, stream);
    case ST_FUNC: return s.fn(stream);
    case ST_HEAD_DGRAD: return head_dgrad_launch((const HeadDgradProblem *)s.dev, (int)s.hdg.size(), s.blocks, stream);
    case ST_WGRAD_STAT: return wgrad_stat_launch(s.wga, stream);
    case ST_CHAIN:
      return chain_launch((const ChainProblem *)s.dev, (int)s.cprobs.size(), (const ChainOp *)s.cops_dev, s.blocks, s.lds_floats, s.chain_bm, stream);
  }
  return hipSuccess;
}

// Records the FDQL_PHASE_ALL launch list of the current plan into a graph (nothing executes during the capture).
int capture_update(fdql_agent *a) {
  if (!a->cap_stream) FDQL_HIP(hipStreamCreateWithFlags(&a->cap_stream, hipStreamNonBlocking));
  FDQL_HIP(hipStreamBeginCapture(a->cap_stream, hipStreamCaptureModeThreadLocal));
  hipError_t bad = hipSuccess;
  const char *where = "";
  for (Stage &st : a->stages) {
    if (!st.runs_in(FDQL_PHASE_ALL)) continue;
    bad = run_stage(a, st, a->cap_stream);
    if (bad != hipSuccess) { where = st.name.c_str(); break; }
  }
  hipGraph_t graph = nullptr;
  hipError_t e = hipStreamEndCapture(a->cap_stream, &graph);
  if (bad != hipSuccess || e != hipSuccess) {
    if (graph) (void)hipGraphDestroy(graph);
    set_error("graph capture%s%s: %s", where[0] ? " at stage " : "", where, hipGetErrorString(bad != hipSuccess ? bad : e));
    return FDQL_EHIP;
  }
  e = hipGraphInstantiate(&a->graph.exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e != hipSuccess) { a->graph.exec = nullptr; set_error("hipGraphInstantiate: %s", hipGetErrorString(e)); return FDQL_EHIP; }
  return 0;
}

int prepare_update(fdql_agent *a, const fdql_batch_t *batch, const float *noise_target, const float *noise_actor,
                   uint64_t seed) {
  if (!a || !a->bound) { set_error("fdql_agent_update: agent not bound"); return FDQL_ESTATE; }
  FDQL_REQUIRE(batch && batch->action && batch->reward && batch->task_done && batch->episode_step,
               "fdql_agent_update: batch needs action, reward, task_done, episode_step");
  FDQL_REQUIRE(!a->cfg.obs_dim || batch->obs_1d, "obs_dim > 0 needs obs_1d");
  FDQL_REQUIRE(!a->cfg.img_c || (a->cfg.obs_2d_u8 ? batch->obs_2d_u8 != nullptr : batch->obs_2d != nullptr),
               "img_c > 0 needs the frames: batch.obs_2d (float32), or batch.obs_2d_u8 for an agent created with obs_2d_u8");
  FDQL_REQUIRE(!batch->obs_2d_slots || (a->cfg.obs_2d_u8 && !a->conv.empty() && a->conv[0].fast_fwd && a->conv[0].fast_wgrad),
               "batch.obs_2d_slots (frames read from the ring in place) needs fdql_agent_conv_reads_ring()");
  FDQL_REQUIRE(!a->cfg.goal_dim || (batch->achieved_goal && batch->desired_goal), "goal_dim > 0 needs achieved/desired goal");
  FDQL_REQUIRE(!a->cfg.use_lowerbound || batch->mc_return, "use_lowerbound needs mc_return");
  FDQL_REQUIRE(!(a->cfg.joiner_gru && a->cfg.gru_state_mode == 1) || batch->agent_state,
               "GRU joiner in store mode needs batch.agent_state");
  if (!a->plan_ready || memcmp(&a->batch, batch, sizeof(*batch)) != 0) {
    if (a->plan_ready) {   // keep the plan being replaced
      a->plan_cache.push_back({a->batch, std::move(a->stages), a->tables_dev, a->graph});
      a->graph = fdql_agent::PlanGraph();
      a->tables_dev = nullptr;
      a->plan_ready = false;
      if (a->plan_cache.size() > a->plan_cache_max) {
        (void)hipFree(a->plan_cache.front().tables_dev);   // synchronises: nothing still reads the evicted tables
        a->plan_cache.front().graph.reset();
        a->plan_cache.erase(a->plan_cache.begin());
      }
    }
    a->stages.clear();
    for (size_t i = 0; i < a->plan_cache.size(); ++i) {
      if (memcmp(&a->plan_cache[i].batch, batch, sizeof(*batch)) != 0) continue;
      a->batch = *batch;
      a->stages = std::move(a->plan_cache[i].stages);
      a->tables_dev = a->plan_cache[i].tables_dev;
      a->graph = a->plan_cache[i].graph;
      a->plan_cache.erase(a->plan_cache.begin() + i);
      a->plan_ready = true;
      break;
    }
    if (!a->plan_ready) {
      a->batch = *batch;
      a->graph = fdql_agent::PlanGraph();
      int rc = build_plan(a);
      if (rc) return rc;
      ++a->plans_built;
    }
  }
  a->noise_t = noise_target;
  a->noise_a = noise_actor;
  a->seed = seed;
  return 0;
}

}  // namespace

// ======================================================================================= C ABI
extern "C" {

const char *fdql_last_error(void) { return g_err; }
int fdql_version(void) { return 1; }
void fdql_abi_sizes(int32_t *out6) {
  out6[0] = (int32_t)sizeof(fdql_agent_config_t);
  out6[1] = (int32_t)sizeof(fdql_batch_t);
  out6[2] = (int32_t)sizeof(fdql_agent_stats_t);
  out6[3] = (int32_t)sizeof(fdql_kernel_time_t);
  out6[4] = (int32_t)sizeof(fdql_reward_fn_t);
  out6[5] = (int32_t)sizeof(fdql_episode_spec_t);
}

int fdql_agent_create(fdql_agent_t **out, const fdql_agent_config_t *cfg) {
  FDQL_REQUIRE(out && cfg, "null argument");
  const fdql_agent_config_t &c = *cfg;
  FDQL_REQUIRE(c.obs_dim >= 0 && c.act_dim > 0 && c.goal_dim >= 0 && c.img_c >= 0 && (c.obs_dim > 0 || c.img_c > 0),
               "bad obs/act/goal dims");
  if (c.img_c > 0) {
    FDQL_REQUIRE(c.n_conv >= 1 && c.n_conv <= FDQL_MAX_CONV && c.img_h > 0 && c.img_w > 0, "pixel input needs 1..%d conv layers", FDQL_MAX_CONV);
    int h = c.img_h, w = c.img_w;
    for (int i = 0; i < c.n_conv; ++i) {
      FDQL_REQUIRE(c.conv_out[i] > 0 && c.conv_k[i] > 0 && c.conv_s[i] > 0 && c.conv_k[i] <= h && c.conv_k[i] <= w,
                   "conv layer %d: bad channels / kernel / stride for a %dx%d map", i, h, w);
      FDQL_REQUIRE(i + 1 == c.n_conv || c.conv_out[i] % 4 == 0, "conv layer %d: out_channels must be a multiple of 4", i);
      h = (h - c.conv_k[i]) / c.conv_s[i] + 1; w = (w - c.conv_k[i]) / c.conv_s[i] + 1;
    }
  }
  FDQL_REQUIRE(!c.discrete || c.act_dim <= 32, "discrete actor: at most 32 actions");
  FDQL_REQUIRE(c.act_dim <= 64, "at most 64 action dimensions (the policy kernels give a row's actions the lanes of one wave)");
  FDQL_REQUIRE(c.n_critics > 0 && c.n_quantiles > 0 && c.n_critics * c.n_quantiles <= 256, "need 0 < C*Q <= 256");
  FDQL_REQUIRE(2 * c.n_critics + 2 <= GEMM_MAX_SEG, "too many critics for one d(state) GEMM");
  FDQL_REQUIRE(c.T >= 2 && c.B >= 1, "need T >= 2, B >= 1");
  FDQL_REQUIRE(c.n_enc_hidden >= 0 && c.n_enc_hidden <= FDQL_MAX_HIDDEN && c.n_joint_hidden >= 0 &&
                   c.n_joint_hidden <= FDQL_MAX_HIDDEN && c.n_pi_hidden >= 0 && c.n_pi_hidden <= FDQL_MAX_HIDDEN &&
                   c.n_critic_hidden >= 0 && c.n_critic_hidden <= FDQL_MAX_HIDDEN, "bad hidden layer counts");
  FDQL_REQUIRE(c.latent > 0 && c.enc_features > 0, "bad latent dims");
  FDQL_REQUIRE(!c.joiner_gru || c.n_joint_hidden == 1,
               "GRU joiner: one layer only (len(joint_hidden_dims) == 1) - the reference's own hidden-state plumbing "
               "(encoder.py:83-87, 117) carries a [latent] vector per row");
  FDQL_REQUIRE(!c.joiner_gru || (c.gru_state_mode >= 0 && c.gru_state_mode <= 2), "gru_state_mode must be 0, 1 or 2");
  FDQL_REQUIRE(c.burn_in_steps >= 0 && c.burn_in_steps <= c.T - 1, "burn_in_steps outside [0, T-1]");
  FDQL_REQUIRE(!c.bootstrap_nstep || (!c.distributional && c.use_lowerbound),
               "bootstrap_nstep needs distributional == 0 and use_lowerbound == 1 (the reference forms the term only in "
               "SoftActorCritic.q_loss under use_nStep_lowerbounds)");
  fdql_agent *a = new fdql_agent();
  a->cfg = c;
  {
    // measured on ROCm 7.2 / MI355X: the replay is 0.5-1.5 % SLOWER than the eager launch list at
    // T=50 and at T=2 - the step is bound by its kernels' own latency, the host stays ahead of the queue - so it is opt-in
    const char *e = getenv("FDQL_GRAPH");
    a->use_graph = e && e[0] == '1';
    a->no_buckets = getenv("FDQL_NO_BUCKETS") != nullptr;
    a->force_buckets = getenv("FDQL_FORCE_BUCKETS") != nullptr;   // test hook: the two-bucket plan at world_size 1
    const char *r = getenv("FDQL_ROWGEMM");   // "0": off; "all": every eligible group whatever its size (tests)
    if (const char *pc = getenv("FDQL_PLAN_CACHE")) {
      const int v = atoi(pc);
      if (v >= 0 && v <= 256) a->plan_cache_max = (size_t)v;
    }
    if (r && r[0] == '0') a->rows_min_tiles = 1LL << 60;
    else if (r && !strcmp(r, "all")) a->rows_min_tiles = 1;
    else if (r && atoi(r) > 1) a->rows_min_tiles = atoi(r);
    if (const char *f = getenv("FDQL_WGRAD_STAT_FACTOR")) { if (atoi(f) >= 1) a->wgrad_stat_factor = atoi(f); }
    if (const char *f = getenv("FDQL_SMALL_GEMM")) { if (f[0] == '0') a->small_max_tiles = 0; }
    if (const char *f = getenv("FDQL_SMALL_GEMM_MAX_TILES")) { if (atoi(f) >= 0) a->small_max_tiles = atoi(f); }
    if (const char *f = getenv("FDQL_ROWDGRAD_MIN_BLOCKS")) { if (atoi(f) >= 1) a->rowdgrad_min_blocks = atoi(f); }
    if (const char *f = getenv("FDQL_ROWDGRAD_MAX_BLOCKS")) { if (atoi(f) >= 1) a->rowdgrad_max_blocks = atoi(f); }
  }
  a->T = c.T; a->B = c.B; a->N = c.T * c.B; a->M = (c.T - 1) * c.B; a->A = c.act_dim; a->L = c.latent;
  a->Nq = c.n_critics * c.n_quantiles;
  if (c.distributional) {
    const int drop = (int)(c.drop_frac * a->Nq);
    a->Nt = drop > 0 ? a->Nq - drop : 0;  // quirk q6: [:-0] is empty
    if (a->Nt <= 0) { delete a; set_error("top_quantiles_to_drop leaves no target atoms (reference slice [:-0] is empty)"); return FDQL_EINVAL; }
  } else {
    a->Nt = 1;
  }
  // K-split of the weight-gradient GEMMs: enough slabs to fill 256 CUs (measured on config 2:
  // 8 -> 0.57 ms, 16 -> 0.45 ms, 32 -> 0.41 ms for the wgrad stage; the slab reduction grows by 0.02 ms)
  int s = a->M / 384;
  a->nsplit = s < 1 ? 1 : (s > 32 ? 32 : s);
  layout(a);
  if (c.obs_2d_u8 && (a->conv.empty() || !(a->conv[0].fast_fwd && a->conv[0].fast_wgrad))) {
    delete a;
    set_error("obs_2d_u8: the first conv layer must be one the implicit-GEMM kernels take (csrc/conv.hip: 32 x 8 / 4 on 4 x 84 x 84 "
              "frame stacks); other stacks take float32 frames (batch.obs_2d)");
    return FDQL_EINVAL;
  }
  {
    // With the dense 256 x 256 blocks on the output-stationary kernel (wgrad.h: one slab per workgroup of a block, ncu /
    // blocks of them - 16 to 18 at config 2) and the narrow ones on single-wave streaming workgroups (k_stream_wgrad), slabs
    // beyond that count are only cleared and summed: the split is sized to it (config 2: 32 -> 20 slabs, the slab sum in
    // k_adam_polyak 0.031 -> 0.023 ms, the cleared slabs' 50 MB of stores gone from the weight-gradient launch).
    const char *e1 = getenv("FDQL_WGRAD_STAT"), *e2 = getenv("FDQL_STREAM_WGRAD");
    int nblk = 0;
    auto count = [&](const MlpDesc &d, int copies) {
      int feat = d.din;
      for (size_t i = 0; i < d.hid.size(); ++i) {
        if (d.hid[i] == WG_N) nblk += copies * ((i == 0 ? d.din : (d.hid[i - 1] == WG_N ? WG_N : 0)) / WG_N);
        feat += d.hid[i];
      }
      if (d.dout == WG_N) nblk += copies * (feat / WG_N);
    };
    for (const MlpDesc &d : a->critic) count(d, 1);
    const int nblk_critics = nblk;
    count(a->actor, 1); count(a->joiner, 1); count(a->enc_obs, 1);
    hipDeviceProp_t pr;
    int dev = 0;
    if (!(e1 && e1[0] == '0') && !(e2 && e2[0] == '0') && nblk > 0 && nblk <= WG_MAX_INST && a->M % WG_BM == 0 &&
        a->M / a->nsplit <= STREAM_WGRAD_MAX_SLAB_ROWS &&   // (long slabs keep the tile kernels' narrow launches, which want the splits)
        (long long)nblk * (a->M / WG_BM) >= a->wgrad_stat_factor * a->rows_min_tiles && hipGetDevice(&dev) == hipSuccess &&
        hipGetDeviceProperties(&pr, dev) == hipSuccess) {
      // a data-parallel (two-bucket) plan launches the critics' blocks and the rest separately (at most one workgroup per slab
      // and block): sized to the larger launch, the critics' (config 2: 10 blocks -> 28 slabs; the 5-block launch of the
      // rest then runs 140 workgroups instead of 100)
      const int per_launch = a->bucketed() && nblk_critics > 0 ? std::max(nblk_critics, nblk - nblk_critics) : nblk;
      const int want = std::max(8, pr.multiProcessorCount / per_launch + 3);
      if (want < a->nsplit) a->nsplit = want;
    }
  }
  if (const char *e = getenv("FDQL_NSPLIT")) {  // tuning hook: K-split of the weight-gradient GEMMs
    const int v = atoi(e);
    if (v >= 1 && v <= 64) a->nsplit = v;
  }
  {
    const char *we = getenv("FDQL_WSTAT");
    bool ws_ok = !(we && (we[0] == '0' || !strcmp(we, "fwd"))) && getenv("FDQL_NO_DSTATE_SPLIT") == nullptr;
    for (const MlpDesc &d : a->critic) ws_ok = ws_ok && !d.hid.empty() && d.hid[0] == WS_N;
    ws_ok = ws_ok && c.latent == WS_N && a->M % WS_BM == 0 && c.n_critics <= WS_MAX_INST && c.n_quantiles <= 32 &&
            (long long)c.n_critics * (a->M / RG_BM) >= a->rows_min_tiles;
    // (few rows: one problem per network + a summing pass also beats the one-problem form on the small-batch kernel, whose
    // slices would walk 2 (C + 1) K-segments in four serial load passes: 0.344 against 0.325 ms per temporal_len-2 step)
    a->dstate_split = a->M <= DSTATE_SPLIT_MAX_ROWS || ws_ok;
  }
  carve(a);
  a->ws_need = a->carve_top;
  *out = a;
  return 0;
}

int fdql_agent_destroy(fdql_agent_t *a) {
  if (!a) return 0;
  if (a->tables_dev) (void)hipFree(a->tables_dev);
  a->graph.reset();
  for (auto &c : a->plan_cache) { (void)hipFree(c.tables_dev); c.graph.reset(); }
  if (a->cap_stream) (void)hipStreamDestroy(a->cap_stream);
  delete a;
  return 0;
}

int64_t fdql_agent_arena_floats(const fdql_agent_t *a, int32_t which) {
  if (!a) return -1;
  if (which == 0) return a->n_train;
  if (which == 1) return a->tgt_end - a->tgt_begin;
  if (which == 2) return a->crit_end - a->crit_begin;
  return -1;
}

int32_t fdql_agent_tensor_info(const fdql_agent_t *a, int32_t index, char *name, int32_t name_cap, int32_t *arena,
                               int64_t *offset_floats, int32_t *shape2) {
  if (!a) return -1;
  if (index < 0) return (int32_t)a->tensors.size();
  if (index >= (int32_t)a->tensors.size()) return -1;
  const TensorInfo &t = a->tensors[index];
  if (name && name_cap > 0) { strncpy(name, t.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
  if (arena) *arena = t.arena;
  if (offset_floats) *offset_floats = t.off;
  if (shape2) { shape2[0] = t.rows; shape2[1] = t.cols; }
  return 0;
}

int64_t fdql_agent_workspace_bytes(const fdql_agent_t *a) { return a ? a->ws_need : -1; }

int fdql_agent_bind(fdql_agent_t *a, float *params, float *grads, float *adam_m, float *adam_v, float *targets,
                    float *frozen, void *workspace, int64_t workspace_bytes) {
  FDQL_REQUIRE(a, "null agent");
  std::lock_guard<std::mutex> lk(a->mu);
  FDQL_REQUIRE(params && grads && adam_m && adam_v && targets && workspace, "null pointer in bind");
  FDQL_REQUIRE(workspace_bytes >= a->ws_need, "workspace too small: %lld < %lld", (long long)workspace_bytes, (long long)a->ws_need);
  FDQL_REQUIRE(!a->cfg.keep_frozen_copy || frozen, "keep_frozen_copy needs a frozen arena");
  FDQL_REQUIRE((reinterpret_cast<uintptr_t>(params) & 15) == 0 && (reinterpret_cast<uintptr_t>(grads) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(targets) & 15) == 0 && (reinterpret_cast<uintptr_t>(workspace) & 255) == 0,
               "arenas must be 16-byte aligned and the workspace 256-byte aligned");
  a->params = params; a->grads = grads; a->adam_m = adam_m; a->adam_v = adam_v; a->targets = targets; a->frozen = frozen;
  a->ws = (char *)workspace;
  a->ws_bytes = workspace_bytes;
  carve(a);
  FDQL_HIP(hipMemset(workspace, 0, a->ws_need));
  FDQL_HIP(hipMemset(grads, 0, a->n_train * 4));
  FDQL_HIP(hipMemset(adam_m, 0, a->n_train * 4));
  FDQL_HIP(hipMemset(adam_v, 0, a->n_train * 4));
  DevState st;
  memset(&st, 0, sizeof(st));
  st.alpha_next = (float)exp(a->cfg.init_log_alpha);  // soft_actor_critic.py:41 (float64 exp)
  st.alpha_cur = st.alpha_next;
  FDQL_HIP(hipMemcpy(a->st(), &st, sizeof(st), hipMemcpyHostToDevice));
  FDQL_HIP(hipDeviceSynchronize());
  a->bound = true;
  a->plan_ready = false;
  a->graph.reset();
  for (auto &c : a->plan_cache) { (void)hipFree(c.tables_dev); c.graph.reset(); }
  a->plan_cache.clear();
  return 0;
}

int fdql_agent_update(fdql_agent_t *a, const fdql_batch_t *batch, const float *noise_target, const float *noise_actor,
                      uint64_t seed, int32_t phase, void *stream) {
  if (!a) { set_error("null agent"); return FDQL_EINVAL; }
  std::lock_guard<std::mutex> lk(a->mu);
  if (phase < FDQL_PHASE_ALL || phase > FDQL_PHASE_GRAD_REST) { set_error("unknown phase %d", (int)phase); return FDQL_EINVAL; }
  if (phase != FDQL_PHASE_APPLY && phase != FDQL_PHASE_GRAD_REST) {
    int rc = prepare_update(a, batch, noise_target, noise_actor, seed);
    if (rc) return rc;
  } else if (!a->plan_ready) {
    set_error("FDQL_PHASE_APPLY / FDQL_PHASE_GRAD_REST before the phase that starts the step");
    return FDQL_ESTATE;
  }
  // split-phase protocol: GRAD_CRITICS -> GRAD_REST -> APPLY, or GRAD -> APPLY; GRAD / GRAD_CRITICS / ALL (re)start a step
  if (phase == FDQL_PHASE_GRAD_REST && a->step_state != fdql_agent::STEP_CRITICS_DONE) {
    set_error("FDQL_PHASE_GRAD_REST needs FDQL_PHASE_GRAD_CRITICS of the same step right before it");
    return FDQL_ESTATE;
  }
  if (phase == FDQL_PHASE_APPLY && a->step_state != fdql_agent::STEP_GRAD_DONE) {
    set_error("FDQL_PHASE_APPLY needs a finished gradient (FDQL_PHASE_GRAD, or FDQL_PHASE_GRAD_CRITICS + FDQL_PHASE_GRAD_REST) of the same step");
    return FDQL_ESTATE;
  }
  // the state this call leaves behind when every stage has been launched; until then (and after a failed launch) there is
  // no step an APPLY / GRAD_REST could continue
  const fdql_agent::StepState done_state = phase == FDQL_PHASE_GRAD_CRITICS ? fdql_agent::STEP_CRITICS_DONE
                  : (phase == FDQL_PHASE_GRAD || phase == FDQL_PHASE_GRAD_REST) ? fdql_agent::STEP_GRAD_DONE : fdql_agent::STEP_NONE;
  a->step_state = fdql_agent::STEP_NONE;
  hipStream_t s = (hipStream_t)stream;
  if (phase == FDQL_PHASE_ALL && a->use_graph) {
    fdql_agent::PlanGraph &g = a->graph;
    if (g.seed != a->seed || g.noise_t != a->noise_t || g.noise_a != a->noise_a) {
      g.reset();
      g.seed = a->seed; g.noise_t = a->noise_t; g.noise_a = a->noise_a;
    }
    if (!g.exec && g.eager_runs >= 1) {   // the first run stays eager: one-time set-up inside the launchers happens there
      int rc = capture_update(a);
      if (rc) return rc;
    }
    if (g.exec) {
      FDQL_HIP(hipGraphLaunch(g.exec, s));
      ++a->graph_launches;
      a->step_state = done_state;
      return 0;
    }
    ++g.eager_runs;
  }
  for (Stage &st : a->stages) {
    if (!st.runs_in(phase)) continue;
    hipError_t e = run_stage(a, st, s);
    if (e != hipSuccess) { set_error("stage %s: %s", st.name.c_str(), hipGetErrorString(e)); return FDQL_EHIP; }
  }
  a->step_state = done_state;
  return 0;
}

int32_t fdql_agent_conv_reads_ring(const fdql_agent_t *a) {
  return a && a->cfg.obs_2d_u8 && !a->conv.empty() && a->conv[0].fast_fwd && a->conv[0].fast_wgrad ? 1 : 0;
}

int fdql_agent_grad_bucket(fdql_agent_t *a, int64_t *first_early_float) {
  FDQL_REQUIRE(a && first_early_float, "null argument");
  std::lock_guard<std::mutex> lk(a->mu);
  *first_early_float = a->bucketed() ? a->crit_begin : a->n_train;
  return 0;
}

int32_t fdql_agent_profile_update(fdql_agent_t *a, const fdql_batch_t *batch, const float *noise_target,
                                  const float *noise_actor, uint64_t seed, fdql_kernel_time_t *out, int32_t cap,
                                  void *stream) {
  if (!a) { set_error("null agent"); return FDQL_EINVAL; }
  std::lock_guard<std::mutex> lk(a->mu);
  int rc = prepare_update(a, batch, noise_target, noise_actor, seed);
  if (rc) return rc;
  a->step_state = fdql_agent::STEP_NONE;   // a whole update of its own: no split-phase step survives it
  hipStream_t s = (hipStream_t)stream;
  // one entry per kernel launch: GEMM stages launch one kernel per tile shape
  struct Part { Stage *st; int shape; };   // shape < -1: row-block group -(shape + 2)
  std::vector<Part> parts;
  for (Stage &st : a->stages) {
    if (!st.runs_in(FDQL_PHASE_ALL)) continue;
    if (st.kind == ST_GEMM) {
      for (size_t r = 0; r < st.rows.size(); ++r) parts.push_back({&st, -2 - (int)r});
      for (int sh = 0; sh < GEMM_NSHAPES; ++sh)
        if (st.sub[sh].blocks > 0) parts.push_back({&st, sh});
    } else {
      parts.push_back({&st, -1});
    }
  }
  const size_t n = parts.size();
  std::vector<hipEvent_t> ev(n + 1);
  for (auto &e : ev) FDQL_HIP(hipEventCreate(&e));
  FDQL_HIP(hipEventRecord(ev[0], s));
  for (size_t i = 0; i < n; ++i) {
    hipError_t e;
    if (parts[i].shape >= 0) {
      const GemmSub &sub = parts[i].st->sub[parts[i].shape];
      e = gemm_launch((const GemmProblem *)sub.dev, (int)sub.probs.size(), sub.blocks, parts[i].shape, s);
    } else if (parts[i].shape < -1) {
      e = parts[i].st->rows[-2 - parts[i].shape].launch(s);
    } else {
      e = run_stage(a, *parts[i].st, s);
    }
    if (e != hipSuccess) { set_error("stage %s: %s", parts[i].st->name.c_str(), hipGetErrorString(e)); return FDQL_EHIP; }
    FDQL_HIP(hipEventRecord(ev[i + 1], s));
  }
  FDQL_HIP(hipStreamSynchronize(s));
  static const char *shape_names[GEMM_NSHAPES] = {"128x128", "128x32", "32x128", "64x128", "64x64dual", "64x64", "64x64hf", "dma128x128", "dma128x64", "dma64x64", "small"};
  int32_t cnt = 0;
  for (size_t i = 0; i < n && cnt < cap; ++i, ++cnt) {
    float ms = 0;
    FDQL_HIP(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
    const Stage &st = *parts[i].st;
    double flops = st.flops, bytes = st.bytes;
    if (parts[i].shape >= 0) {
      flops = 0; bytes = 0;
      for (const auto &p : st.sub[parts[i].shape].probs) { flops += gemm_flops(p); bytes += gemm_bytes(p); }
      snprintf(out[cnt].name, sizeof(out[cnt].name), "gemm%s:%s", shape_names[parts[i].shape], st.name.c_str());
    } else if (parts[i].shape < -1) {
      const RowsLaunch &rl = st.rows[-2 - parts[i].shape];
      if (rl.dot) {
        flops = rowdot_flops(rl.rdot);
        bytes = 4.0 * rl.rdot.M * rl.rdot.nprob * (double)(RD_K + rl.rdot.Q + rl.rdot.A);
        snprintf(out[cnt].name, sizeof(out[cnt].name), "rowdot<%d>:%s", rl.rdot.A, st.name.c_str());
      } else if (rl.rd && rl.chain3) {
        flops = rowchain_flops(rl.rch);
        bytes = 4.0 * rl.rch.M * (double)RD_N * (rl.rch.nsum + 4 + 2);
        snprintf(out[cnt].name, sizeof(out[cnt].name), "rowdchain:%s+denc+enc_obs.dpre0", st.name.c_str());
      } else if (rl.rd) {
        flops = rowdgrad_flops(rl.rda);
        bytes = 4.0 * rl.rda.M * (double)RD_N * (rl.rda.nseg + 1 + (rl.rda.gate ? 1 : 0));
        snprintf(out[cnt].name, sizeof(out[cnt].name), "rowd<%d,%d>:%s", rl.rda.nseg, rl.rda.gate, st.name.c_str());
      } else if (rl.ws) {
        flops = wstat_flops(rl.wa);
        bytes = 4.0 * rl.wa.M * rl.wa.ninst * (double)(WS_KMAIN + WS_N * (rl.wa.dual ? 2 : 1));
        // (the name carries the kernel's template arguments: bench.py / tools map it to the rocprofv3 kernel name)
        if (rl.wa.grad) {
          bytes = 4.0 * rl.wa.M * rl.wa.ninst * (double)(WS_KMAIN * (rl.wa.fz ? 2 : 1) + 2 * WS_N);
          snprintf(out[cnt].name, sizeof(out[cnt].name), "wstatg<%d,%d,%d>:%s", rl.wa.fz ? 1 : 0, rl.wa.grad == 2 ? 1 : 0, rl.wa.nslot_loop, st.name.c_str());
        } else {
          snprintf(out[cnt].name, sizeof(out[cnt].name), "wstat<%d,%d,%d>:%s", rl.wa.nslot_loop, rl.wa.nslot_tail, rl.wa.hf_q, st.name.c_str());
        }
      } else {
      const RowGemmArgs &ra = rl.rg;
      flops = rowgemm_flops(ra);
      bytes = 4.0 * ra.M * ra.ninst * (double)(RG_KMAIN + RG_N * (ra.dual ? 2 : 1) + (ra.grad ? RG_N : 0));
      snprintf(out[cnt].name, sizeof(out[cnt].name), "rows%s%s:%s", ra.grad ? "KS" : "", ra.dual ? "dual" : "", st.name.c_str());
      }
    } else {
      snprintf(out[cnt].name, sizeof(out[cnt].name), "%s%s",
               st.kind == ST_SKINNY_WGRAD ? (st.stream ? "nwgrad:" : "colsum:")
                                          : (st.kind == ST_CHAIN ? "chain:" : (st.kind == ST_WGRAD_STAT ? "wgstat:" : (st.mfma ? "conv:" : "k:"))),
               st.name.c_str());
    }
    out[cnt].ms = ms;
    out[cnt].flops = flops;
    out[cnt].bytes = bytes;
  }
  for (auto &e : ev) (void)hipEventDestroy(e);
  return cnt;
}

// ---- act(): encoder -> joiner -> actor on a few rows (deepQlearning.py:155-187)
static int64_t act_mlp_floats(const MlpDesc &d, int64_t rows) {
  int64_t n = 0;
  for (int h : d.hid) n += pad4(rows * h);
  return n + pad4(rows * d.dout);
}

int64_t fdql_agent_act_workspace_bytes(const fdql_agent_t *a, int32_t rows) {
  if (!a || rows < 0) return -1;
  int64_t joiner = act_mlp_floats(a->joiner, rows);
  if (a->cfg.joiner_gru)   // gi, gh [rows, 3L], zero start state and the new state [rows, L]
    joiner = 2 * pad4((int64_t)rows * 3 * a->cfg.latent) + 2 * pad4((int64_t)rows * a->cfg.latent);
  int64_t conv = 0;   // im2col matrix + NHWC output of every conv layer
  for (const auto &Lc : a->conv) {
    const int64_t pos = (int64_t)rows * Lc.g.OH * Lc.g.OW;
    conv += pad4(pos * Lc.g.C * Lc.g.k * Lc.g.k) + pad4(pos * Lc.cout);
  }
  return 4 * (conv + act_mlp_floats(a->enc_obs, rows) + joiner + act_mlp_floats(a->actor, rows));
}

namespace {
// hidden layers then the skip head over cat(in, h_0..h_{n-1}) (mlp.py:88-94); returns the output buffer
// pol: the policy head follows this MLP - when its last layer is narrow enough both run as one launch (*fused_policy = true)
hipError_t act_mlp(const fdql_agent *a, const MlpDesc &d, const ActSeg *in, int nin, int rows, float *&top, float **out,
                   hipStream_t s, const ActPolicyArgs *pol = nullptr, bool *fused_policy = nullptr) {
  ActSeg feats[ACT_MAX_SEG];
  int nf = 0;
  for (int i = 0; i < nin; ++i) feats[nf++] = in[i];
  const bool no_fuse = getenv("FDQL_ACT_NO_FUSE") != nullptr;   // (tuning / test hook: every layer its own launch, policy its own)
  // One hidden layer over a few input columns (the observation encoder at config 2 / 3: 17 / 48 columns): the skip head's
  // launch recomputes it per workgroup (ActLayerArgs::pre_*) instead of waiting for a launch of its own
  int kin = 0;
  for (int i = 0; i < nin; ++i) kin += in[i].width;
  if (!no_fuse && d.hid.size() == 1 && kin <= 64 && d.hid[0] <= 256 && nin + 1 <= ACT_MAX_SEG) {
    ActLayerArgs l;
    memset(&l, 0, sizeof(l));
    for (int j = 0; j < nin; ++j) l.in[j] = in[j];
    l.in[nin] = {nullptr, d.hid[0], d.hid[0]};
    l.nseg = nin + 1;
    l.pre_W = a->params + d.w_off[0]; l.pre_ldw = d.in_of(0); l.pre_bias = a->params + d.b_off[0]; l.pre_N = d.hid[0]; l.pre_nseg = nin;
    l.W = a->params + d.hw_off; l.ldw = d.head_ld(); l.bias = a->params + d.hb_off;
    l.out = top; l.ldo = d.dout; l.N = d.dout; l.rows = rows; l.leaky = 0;
    *out = top;
    top += pad4((int64_t)rows * d.dout);
    return act_layer_launch(l, s);
  }
  for (size_t i = 0; i < d.hid.size(); ++i) {
    ActLayerArgs l;
    memset(&l, 0, sizeof(l));
    if (i == 0) { for (int j = 0; j < nin; ++j) l.in[j] = in[j]; l.nseg = nin; }
    else { l.in[0] = feats[nf - 1]; l.nseg = 1; }
    l.W = a->params + d.w_off[i]; l.ldw = d.in_of((int)i); l.bias = a->params + d.b_off[i];
    l.out = top; l.ldo = d.hid[i]; l.N = d.hid[i]; l.rows = rows; l.leaky = 1;
    hipError_t e = act_layer_launch(l, s);
    if (e != hipSuccess) return e;
    feats[nf++] = {top, d.hid[i], d.hid[i]};
    top += pad4((int64_t)rows * d.hid[i]);
  }
  ActLayerArgs l;
  memset(&l, 0, sizeof(l));
  for (int j = 0; j < nf; ++j) l.in[j] = feats[j];
  l.nseg = nf;
  l.W = a->params + d.hw_off; l.ldw = d.head_ld(); l.bias = a->params + d.hb_off;
  l.out = top; l.ldo = d.dout; l.N = d.dout; l.rows = rows; l.leaky = 0;
  *out = top;
  top += pad4((int64_t)rows * d.dout);
  if (pol && fused_policy && !no_fuse && act_head_policy_takes(l, *pol)) {
    *fused_policy = true;
    return act_head_policy_launch(l, *pol, s);
  }
  return act_layer_launch(l, s);
}
// The one-launch form (k_act_fused): feed-forward encoder / joiner / actor, at most ACTF_ROWS rows.  Feature rows of an MLP
// in LDS: [input blocks | h_0 | h_1 ...], every block starting on a multiple of 4 floats; a layer reads blocks of its own
// MLP's rows, the skip head all of them (mlp.py:88-94) and writes the next MLP's first input block.
bool act_fused_build(const fdql_agent *a, const ActSeg *in, int nin, int rows, ActFusedArgs &fa) {
  // Opt-in (FDQL_ACT_FUSED=1): measured SLOWER than the launch-per-layer form - 60.8 us against 39.7 us for one row at
  // config-2 widths - one CU pulling a layer's 256 KB of weights takes ~10 us per layer, more than the launch it saves
  // (DESIGN.md section 5).  Kept for the parity test and as the starting point of a multi-workgroup form.
  const char *env = getenv("FDQL_ACT_FUSED");
  if (!env || env[0] != '1') return false;
  if (rows > ACTF_ROWS || a->cfg.joiner_gru || !a->conv.empty() || nin > ACTF_MAX_IN) return false;
  memset(&fa, 0, sizeof(fa));
  fa.rows = rows;
  const MlpDesc *mlps[3] = {&a->enc_obs, &a->joiner, &a->actor};
  int top = 0;
  struct Rows { int base, pitch; std::vector<int> off, width; } fr[3];
  for (int m = 0; m < 3; ++m) {
    const MlpDesc &d = *mlps[m];
    int at = 0;
    auto block = [&](int w) { fr[m].off.push_back(at); fr[m].width.push_back(w); at += (int)pad4(w); };
    if (m == 0) for (int j = 0; j < nin; ++j) block(in[j].width);
    else block(mlps[m - 1]->dout);
    for (int h : d.hid) block(h);
    fr[m].base = top; fr[m].pitch = at;
    top += ACTF_ROWS * at;
  }
  fa.logits_off = top; fa.logits_pitch = (int)pad4(a->actor.dout);
  top += ACTF_ROWS * fa.logits_pitch;
  if (top > ACTF_LDS_FLOATS) return false;
  for (int j = 0; j < nin; ++j) fa.in[j] = {in[j].ptr, in[j].ld, in[j].width, fr[0].base + fr[0].off[j], fr[0].pitch};
  fa.nin = nin;
  for (int m = 0; m < 3; ++m) {
    const MlpDesc &d = *mlps[m];
    const int nfirst = m == 0 ? nin : 1, nh = (int)d.hid.size();
    if (nfirst + nh > ACTF_MAX_SEG) return false;
    for (int i = 0; i <= nh; ++i) {   // i == nh: the skip head
      if (fa.nlayers >= ACTF_MAX_LAYERS) return false;
      ActFusedLayer &L = fa.L[fa.nlayers++];
      L.x_pitch = fr[m].pitch;
      int wcol = 0;
      auto seg = [&](int blk) { L.seg[L.nseg++] = {fr[m].base + fr[m].off[blk], wcol, fr[m].width[blk]}; wcol += fr[m].width[blk]; };
      if (i == 0 || i == nh) for (int j = 0; j < nfirst; ++j) seg(j);
      if (i == nh) for (int j = 0; j < nh; ++j) seg(nfirst + j);
      else if (i > 0) seg(nfirst + i - 1);
      if (i < nh) {
        L.W = a->params + d.w_off[i]; L.ldw = d.in_of(i); L.bias = a->params + d.b_off[i]; L.N = d.hid[i]; L.leaky = 1;
        L.out_off = fr[m].base + fr[m].off[nfirst + i]; L.out_pitch = fr[m].pitch;
      } else {
        L.W = a->params + d.hw_off; L.ldw = d.head_ld(); L.bias = a->params + d.hb_off; L.N = d.dout; L.leaky = 0;
        if (m < 2) { L.out_off = fr[m + 1].base; L.out_pitch = fr[m + 1].pitch; }
        else { L.out_off = fa.logits_off; L.out_pitch = fa.logits_pitch; }
      }
    }
  }
  return true;
}
}  // namespace

int fdql_agent_act(fdql_agent_t *a, const float *obs_1d, const float *achieved_goal, const float *desired_goal,
                   const float *obs_2d, const float *agent_state, const uint8_t *exploit_mask, const float *noise,
                   uint64_t seed, uint64_t counter, int32_t rows, float *action, float *log_prob, float *explore_action,
                   float *exploit_action, float *hidden_state, void *workspace, int64_t workspace_bytes, void *stream) {
  if (!a) { set_error("null agent"); return FDQL_EINVAL; }
  std::lock_guard<std::mutex> lk(a->mu);
  if (!a->bound) { set_error("fdql_agent_act: agent not bound"); return FDQL_ESTATE; }
  FDQL_REQUIRE(rows >= 0, "fdql_agent_act: rows < 0");
  if (rows == 0) return 0;
  const fdql_agent_config_t &c = a->cfg;
  FDQL_REQUIRE(action && (obs_1d || !c.obs_dim) && (obs_2d || !c.img_c), "fdql_agent_act: action and the observation inputs are required");
  FDQL_REQUIRE(!c.goal_dim || (achieved_goal && desired_goal), "goal_dim > 0 needs achieved/desired goal");
  FDQL_REQUIRE(workspace && workspace_bytes >= fdql_agent_act_workspace_bytes(a, rows) &&
                   (reinterpret_cast<uintptr_t>(workspace) & 15) == 0,
               "fdql_agent_act: workspace too small or misaligned (need %lld bytes)",
               (long long)fdql_agent_act_workspace_bytes(a, rows));
  FDQL_REQUIRE((int)a->enc_obs.hid.size() + 3 <= ACT_MAX_SEG, "too many hidden layers for act()");
  hipStream_t s = (hipStream_t)stream;
  float *top = (float *)workspace;
  ActSeg in[4];
  int nin = 0;
  if (c.obs_dim) in[nin++] = {obs_1d, c.obs_dim, c.obs_dim};
  if (c.goal_dim) {  // encoder.py:54-58: cat(obs_1d, achieved_goal, desired_goal) as K-segments
    in[nin++] = {achieved_goal, c.goal_dim, c.goal_dim};
    in[nin++] = {desired_goal, c.goal_dim, c.goal_dim};
  }
  float *enc = nullptr, *state = nullptr, *logits = nullptr;
  hipError_t e = hipSuccess;
  {   // a handful of rows through feed-forward networks: the whole of act() in one launch
    ActFusedArgs fa;
    if (act_fused_build(a, in, nin, rows, fa)) {
      ActPolicyArgs &p = fa.pol;
      p.ld = a->actor.dout; p.rows = rows; p.A = c.act_dim; p.discrete = c.discrete;
      p.exploit_mask = exploit_mask; p.noise = noise; p.seed = seed; p.counter = counter;
      p.action = action; p.log_prob = log_prob; p.explore = explore_action; p.exploit = exploit_action;
      e = act_fused_launch(fa, s);
      if (e != hipSuccess) { set_error("fdql_agent_act: %s", hipGetErrorString(e)); return FDQL_EHIP; }
      return 0;
    }
  }
  {   // pixel encoder: im2col + one skinny layer launch per conv layer (rows * OH * OW "batch rows")
    const float *cin = obs_2d;
    for (size_t i = 0; i < a->conv.size() && e == hipSuccess; ++i) {
      const fdql_agent::ConvLayer &Lc = a->conv[i];
      const int K = Lc.g.C * Lc.g.k * Lc.g.k;
      const int64_t pos = (int64_t)rows * Lc.g.OH * Lc.g.OW;
      float *col = top; top += pad4(pos * K);
      float *out = top; top += pad4(pos * Lc.cout);
      e = im2col_launch(cin, i > 0, i == 0 ? 1.0f / 255.0f : 1.0f, rows, Lc.g, col, s);
      ActLayerArgs l;
      memset(&l, 0, sizeof(l));
      l.in[0] = {col, K, K}; l.nseg = 1; l.W = a->params + Lc.w_off; l.ldw = K; l.bias = a->params + Lc.b_off;
      l.out = out; l.ldo = Lc.cout; l.N = Lc.cout; l.rows = (int)pos; l.leaky = 1;
      if (e == hipSuccess) e = act_layer_launch(l, s);
      cin = out;
    }
    if (!a->conv.empty()) in[nin++] = {cin, a->conv_feat, a->conv_feat};
  }
  if (e == hipSuccess) e = act_mlp(a, a->enc_obs, in, nin, rows, top, &enc, s);
  if (e == hipSuccess && !c.joiner_gru) {
    ActSeg x = {enc, a->enc_obs.dout, a->enc_obs.dout};
    e = act_mlp(a, a->joiner, &x, 1, rows, top, &state, s);
  } else if (e == hipSuccess) {
    // one GRU step from the carried hidden state (encoder.py:63-65, 72-76; NULL = zeros like nn.GRU's default)
    const int L = c.latent, L3 = 3 * c.latent, F = c.enc_features;
    float *gi = top; top += pad4((int64_t)rows * L3);
    float *gh = top; top += pad4((int64_t)rows * L3);
    float *hz = top; top += pad4((int64_t)rows * L);
    float *hn = top; top += pad4((int64_t)rows * L);
    const float *hp = agent_state;
    if (!hp) { e = gru_h0_launch(0, nullptr, hz, rows, L, s); hp = hz; }
    ActLayerArgs l;
    memset(&l, 0, sizeof(l));
    l.in[0] = {enc, F, F}; l.nseg = 1; l.W = a->params + a->gru_wih; l.ldw = F; l.bias = a->params + a->gru_bih;
    l.out = gi; l.ldo = L3; l.N = L3; l.rows = rows; l.leaky = 0;
    if (e == hipSuccess) e = act_layer_launch(l, s);
    l.in[0] = {hp, L, L}; l.W = a->params + a->gru_whh; l.ldw = L; l.bias = a->params + a->gru_bhh; l.out = gh;
    if (e == hipSuccess) e = act_layer_launch(l, s);
    float *hout = hidden_state ? hidden_state : hn;
    if (e == hipSuccess) e = gru_cell_fwd_launch(gi, gh, nullptr, 0, nullptr, hp, hout, nullptr, rows, L, s);
    state = hout;
  }
  ActPolicyArgs p;
  memset(&p, 0, sizeof(p));
  p.ld = a->actor.dout; p.rows = rows; p.A = c.act_dim; p.discrete = c.discrete;
  p.exploit_mask = exploit_mask; p.noise = noise; p.seed = seed; p.counter = counter;
  p.action = action; p.log_prob = log_prob; p.explore = explore_action; p.exploit = exploit_action;
  bool fused_policy = false;
  if (e == hipSuccess) { ActSeg x = {state, a->joiner.dout, a->joiner.dout}; e = act_mlp(a, a->actor, &x, 1, rows, top, &logits, s, &p, &fused_policy); }
  if (e == hipSuccess && !fused_policy) {
    p.logits = logits;
    e = act_policy_launch(p, s);
  }
  if (e != hipSuccess) { set_error("fdql_agent_act: %s", hipGetErrorString(e)); return FDQL_EHIP; }
  return 0;
}

int fdql_agent_scalars(fdql_agent_t *a, float *host_out8, void *stream) {
  if (!a) { set_error("null agent"); return FDQL_EINVAL; }
  std::lock_guard<std::mutex> lk(a->mu);
  if (!a->bound) { set_error("agent not bound"); return FDQL_ESTATE; }
  DevState st;
  FDQL_HIP(hipMemcpyAsync(host_out8, a->buf("scalars"), 8 * sizeof(float), hipMemcpyDeviceToHost, (hipStream_t)stream));
  FDQL_HIP(hipMemcpyAsync(&st, a->st(), sizeof(st), hipMemcpyDeviceToHost, (hipStream_t)stream));
  FDQL_HIP(hipStreamSynchronize((hipStream_t)stream));
  host_out8[7] = (float)st.step;  // optimiser steps applied so far
  return 0;
}

int fdql_agent_summaries(fdql_agent_t *a, float *host_out, int32_t cap, int32_t with_grad_norms, void *stream) {
  if (!a || !host_out) { set_error("null argument"); return FDQL_EINVAL; }
  std::lock_guard<std::mutex> lk(a->mu);
  if (!a->bound || !a->plan_ready) { set_error("no update has run yet"); return FDQL_ESTATE; }
  // gradient-norm ranges: one per trainable tensor, in fdql_agent_tensor_info order (arena 0)
  std::vector<long long> ranges;
  if (with_grad_norms)
    for (const TensorInfo &t : a->tensors)
      if (t.arena == 0) { ranges.push_back(t.off); ranges.push_back(t.cols > 0 ? (long long)t.rows * t.cols : (t.rows > 0 ? t.rows : 1)); }
  const int nr = (int)ranges.size() / 2;
  FDQL_REQUIRE(cap >= 4 + nr, "summaries need room for %d floats", 4 + nr);
  hipStream_t s = (hipStream_t)stream;
  float *out_dev = a->buf("summaries");
  FDQL_REQUIRE(a->named.at("summaries").second >= (int64_t)((4 + nr + 1) & ~1) + (int64_t)ranges.size() * 2, "scratch too small for the summaries");
  long long *ranges_dev = reinterpret_cast<long long *>(out_dev + ((4 + nr + 1) & ~1));
  if (nr) FDQL_HIP(hipMemcpyAsync(ranges_dev, ranges.data(), ranges.size() * sizeof(long long), hipMemcpyHostToDevice, s));
  hipError_t e = summaries_launch(a->buf("q_pred"), a->M, a->Nq, a->buf("is_contiguous"), a->T - 1, a->B, a->T, a->grads, ranges_dev, nr, out_dev, s);
  if (e != hipSuccess) { set_error("summaries: %s", hipGetErrorString(e)); return FDQL_EHIP; }
  FDQL_HIP(hipMemcpyAsync(host_out, out_dev, (4 + nr) * sizeof(float), hipMemcpyDeviceToHost, s));
  FDQL_HIP(hipStreamSynchronize(s));
  return 4 + nr;
}

int fdql_agent_set_alpha(fdql_agent_t *a, float alpha, void *stream) {
  if (!a) { set_error("null agent"); return FDQL_EINVAL; }
  std::lock_guard<std::mutex> lk(a->mu);
  if (!a->bound) { set_error("agent not bound"); return FDQL_ESTATE; }
  DevState st;
  FDQL_HIP(hipStreamSynchronize((hipStream_t)stream));
  FDQL_HIP(hipMemcpy(&st, a->st(), sizeof(st), hipMemcpyDeviceToHost));
  st.alpha_next = alpha;
  st.alpha_cur = alpha;
  FDQL_HIP(hipMemcpy(a->st(), &st, sizeof(st), hipMemcpyHostToDevice));
  return 0;
}

int fdql_agent_set_step(fdql_agent_t *a, int32_t step, void *stream) {
  if (!a) { set_error("null agent"); return FDQL_EINVAL; }
  std::lock_guard<std::mutex> lk(a->mu);
  if (!a->bound) { set_error("agent not bound"); return FDQL_ESTATE; }
  DevState st;
  FDQL_HIP(hipStreamSynchronize((hipStream_t)stream));
  FDQL_HIP(hipMemcpy(&st, a->st(), sizeof(st), hipMemcpyDeviceToHost));
  st.step = step;
  FDQL_HIP(hipMemcpy(a->st(), &st, sizeof(st), hipMemcpyHostToDevice));
  return 0;
}

int fdql_agent_debug_ptr(fdql_agent_t *a, const char *name, const float **dev_ptr, int64_t *count) {
  if (!a || !a->bound) { set_error("agent not bound"); return FDQL_ESTATE; }
  auto it = a->named.find(name);
  FDQL_REQUIRE(it != a->named.end(), "unknown buffer '%s'", name);
  *dev_ptr = reinterpret_cast<const float *>(a->ws + it->second.first);
  if (count) *count = it->second.second;
  return 0;
}

int fdql_agent_stats(const fdql_agent_t *a, fdql_agent_stats_t *out) {
  FDQL_REQUIRE(a && out, "null argument");
  memset(out, 0, sizeof(*out));
  out->params = a->n_train;
  out->plans_built = a->plans_built;
  out->graph_launches = a->graph_launches;
  for (const Stage &s : a->stages) {
    if (!s.runs_in(FDQL_PHASE_ALL)) continue;
    out->n_launches++;
    if (s.kind == ST_GEMM) {
      out->gemm_flops += s.flops;
      for (const auto &sub : s.sub) if (sub.blocks > 0) out->n_gemm_launches++;
      out->n_gemm_launches += (int)s.rows.size();
    }
    if (s.kind == ST_SKINNY_WGRAD) out->skinny_flops += s.flops;
    if (s.kind == ST_CHAIN || s.kind == ST_WGRAD_STAT || (s.kind == ST_FUNC && s.mfma)) { out->gemm_flops += s.flops; out->n_gemm_launches++; }
  }
  return 0;
}

int fdql_debug_set_gemm_variant(int32_t variant) {
  FDQL_REQUIRE(variant >= 0 && variant <= 6, "variant must be 0..6");
  gemm_set_variant(variant);
  return 0;
}

int fdql_debug_set_gemm_dense_shape(int32_t shape) {
  FDQL_REQUIRE(gemm_shape_is_dense(shape) || shape == GEMM_SMALL,
               "dense shape must be 0 (128x128), 3 (64x128), 5 (64x64), an LDS-DMA shape 7 (128x128), 8 (128x64), 9 (64x64), or 10 (the small-batch kernel wherever it has the form)");
  gemm_set_dense_shape(shape);
  return 0;
}

int fdql_debug_chain_stamps(uint64_t *out, int32_t cap) {
  return chain_read_stamps(reinterpret_cast<unsigned long long *>(out), cap);
}

int fdql_test_chain_mlp(const float *x, int32_t rows, int32_t din, const int32_t *hid, int32_t nh, int32_t dout,
                        const float *weights, float *const *h_out, float *out, void *stream) {
  FDQL_REQUIRE(x && hid && weights && out && rows > 0 && din > 0 && dout > 0 && nh >= 0 && nh <= FDQL_MAX_HIDDEN, "bad arguments");
  fdql_agent tmp;
  MlpDesc d;
  int64_t top = 0;
  add_mlp(&tmp, d, "test", din, hid, nh, dout, top);
  MlpInst m;
  m.d = &d; m.wbase = weights; m.worigin = 0; m.rows = rows;
  m.in.push_back({x, din, din});
  for (int i = 0; i < nh; ++i) m.h.push_back(h_out ? h_out[i] : nullptr);
  m.out = out; m.ldout = dout;
  Stage cs;
  cs.kind = ST_CHAIN; cs.name = "test";
  if (const char *v = getenv("FDQL_CHAIN_BM")) cs.chain_bm = atoi(v) == 32 ? 32 : CH_BM;   // (test hook: 32-row blocks)
  ChainBuilder cb(cs);
  cb.begin(rows);
  ChainImg xi = cb.load(m.in);
  cb.mlp(m, {xi}, true, h_out != nullptr, {0, rows, 0}, false);
  cb.end();
  FDQL_REQUIRE(cb.ok, "this MLP does not fit the chain kernel");
  const int blocks = chain_finalize(cs.cprobs.data(), (int)cs.cprobs.size(), cs.chain_bm);
  void *dev = nullptr;
  const size_t pb = cs.cprobs.size() * sizeof(ChainProblem), ob = cs.cops.size() * sizeof(ChainOp);
  FDQL_HIP(hipMalloc(&dev, pb + ob + 256));
  FDQL_HIP(hipMemcpy(dev, cs.cprobs.data(), pb, hipMemcpyHostToDevice));
  void *odev = (char *)dev + (pb + 255) / 256 * 256;
  FDQL_HIP(hipMemcpy(odev, cs.cops.data(), ob, hipMemcpyHostToDevice));
  hipError_t e = chain_launch((const ChainProblem *)dev, (int)cs.cprobs.size(), (const ChainOp *)odev, blocks, cs.lds_floats, cs.chain_bm, (hipStream_t)stream);
  if (e != hipSuccess) { set_error("chain launch: %s", hipGetErrorString(e)); (void)hipFree(dev); return FDQL_EHIP; }
  FDQL_HIP(hipStreamSynchronize((hipStream_t)stream));
  FDQL_HIP(hipFree(dev));
  return 0;
}

int fdql_test_gemm(const float *A, int32_t lda, int32_t a_kc, const float *B, int32_t ldb, int32_t b_kc,
                   const float *bias, float *C, int32_t ldc, int32_t M, int32_t N, int32_t K, int32_t epilogue,
                   const float *ref, int32_t ldref, int32_t ksplit, void *stream) {
  GemmProblem p;
  memset(&p, 0, sizeof(p));
  p.emit_seg = -1;
  p.M = M; p.N = N; p.C = C; p.ldc = ldc; p.ksplit = ksplit < 1 ? 1 : ksplit; p.split_stride = (long long)M * ldc;
  p.bias = bias; p.epi = epilogue; p.ref = ref; p.ldref = ldref;
  p.nseg = 1;
  p.seg[0].A = A; p.seg[0].lda = lda; p.seg[0].a_kc = a_kc; p.seg[0].B = B; p.seg[0].ldb = ldb; p.seg[0].b_kc = b_kc; p.seg[0].K = K;
  const int shape = gemm_pick_shape(p, gemm_dense_shape());
  const int blocks = gemm_finalize(&p, 1, shape);
  GemmProblem *dev = nullptr;
  FDQL_HIP(hipMalloc(&dev, sizeof(p)));
  FDQL_HIP(hipMemcpy(dev, &p, sizeof(p), hipMemcpyHostToDevice));
  hipError_t e = gemm_launch(dev, 1, blocks, shape, (hipStream_t)stream);
  if (e != hipSuccess) { set_error("gemm launch: %s", hipGetErrorString(e)); (void)hipFree(dev); return FDQL_EHIP; }
  FDQL_HIP(hipStreamSynchronize((hipStream_t)stream));
  FDQL_HIP(hipFree(dev));
  return 0;
}

/* Test hook for the output-stationary weight-gradient kernel (wgrad.hip): nprob blocks dW[i] [256, ldw] (slab 0 at dW + i *
 * 256 * ldw, slabs slab_stride floats apart, nslab of them) = G[i]^T X[i] over M rows each (G, X: [nprob * M, 256]).  Every
 * slab of every block is written (partials or zeros): their sum is the gradient.  FDQL_EINVAL: the kernel does not take the form. */
int fdql_test_conv(int32_t mode, const void *in, int32_t u8, const int32_t *slots,
                   const float *W, const float *bias, const float *dpre, const float *act_prev, float *out, float *scratch,
                   int64_t scratch_floats, int64_t nimg, int32_t C, int32_t H, int32_t Wd, int32_t k, int32_t s, int32_t cout,
                   void *stream) {
  ConvGeom g;
  g.C = C; g.H = H; g.W = Wd; g.k = k; g.s = s;
  FDQL_REQUIRE(k > 0 && s > 0 && H >= k && Wd >= k && nimg > 0, "fdql_test_conv: bad geometry");
  g.OH = (H - k) / s + 1; g.OW = (Wd - k) / s + 1;
  ConvSrc src;
  src.base = in; src.u8 = u8; src.slots = slots;
  hipStream_t st = (hipStream_t)stream;
  if (mode == 0) {
    FDQL_REQUIRE(conv_fwd_takes(g, cout, u8 != 0), "fdql_test_conv: no forward kernel for this layer");
    ConvFwdArgs a;
    a.in = src; a.W = W; a.bias = bias; a.out = out; a.nimg = nimg; a.g = g; a.cout = cout;
    FDQL_HIP(conv_fwd_launch(a, st));
  } else if (mode == 1) {
    FDQL_REQUIRE(!u8 && conv_dgrad_takes(g, cout), "fdql_test_conv: no data-gradient kernel for this layer");
    ConvDgradArgs a;
    a.dpre = dpre; a.W = W; a.act_prev = act_prev; a.dprev = out; a.nimg = nimg; a.g = g; a.cout = cout;
    FDQL_HIP(conv_dgrad_launch(a, st));
  } else if (mode == 2) {
    const int nslab = conv_wgrad_slabs(g, cout, u8 != 0, nimg);
    const long long n = (long long)cout * C * k * k + cout;
    FDQL_REQUIRE(nslab > 0, "fdql_test_conv: no weight-gradient kernel for this layer");
    FDQL_REQUIRE(scratch_floats >= nslab * n, "fdql_test_conv: scratch of %lld floats, %lld needed", (long long)scratch_floats, nslab * n);
    ConvWgradArgs a;
    a.in = src; a.dpre = dpre; a.wpart = scratch; a.nimg = nimg; a.g = g; a.cout = cout;
    FDQL_HIP(conv_wgrad_launch(a, st));
    FDQL_HIP(reduce_partials_launch(scratch, nslab, n, out, st));
  } else {
    FDQL_REQUIRE(false, "fdql_test_conv: unknown mode %d", (int)mode);
  }
  return 0;
}

int fdql_test_wgrad_stat(const float *G, const float *X, float *dW, int32_t M, int32_t nprob, int32_t ldw, int32_t nslab,
                         int64_t slab_stride, void *stream) {
  return fdql_test_wgrad_stat_riders(G, X, dW, M, nprob, ldw, nslab, slab_stride, nullptr, 0, 0, nullptr, 0, 1, nullptr, 0, 0, nullptr, 0, stream);
}

int fdql_test_wgrad_stat_riders(const float *G, const float *X, float *dW, int32_t M, int32_t nprob, int32_t ldw, int32_t nslab,
                                int64_t slab_stride, const float *X2, int32_t nx2, int32_t ldx2, float *dW2, int32_t ldw2, int32_t x2_every,
                                const float *G2, int32_t ng2, int32_t ldg2, float *dW3, int32_t ldw3, void *stream) {
  std::vector<GemmProblem> probs;
  auto wgrad_problem = [&](int nout, int width, float *dst, int ldd, const float *dOut, int ldo, const float *Xp, int ldx) {
    GemmProblem p;
    memset(&p, 0, sizeof(p));
    p.M = nout; p.N = width; p.C = dst; p.ldc = ldd; p.ksplit = nslab; p.split_stride = slab_stride; p.emit_seg = -1;
    GemmSeg &sg = p.seg[p.nseg++];
    sg.A = dOut; sg.lda = ldo; sg.a_kc = 0; sg.B = Xp; sg.ldb = ldx; sg.b_kc = 0; sg.K = M;
    return p;
  };
  for (int i = 0; i < nprob; ++i)
    probs.push_back(wgrad_problem(WG_N, WG_N, dW + (long long)i * WG_N * ldw, ldw, G + (long long)i * M * WG_N, WG_N, X + (long long)i * M * WG_N, WG_N));
  WgArgs wa;
  FDQL_REQUIRE(wgrad_stat_from_problems(probs.data(), nprob, nslab, slab_stride, wa), "the output-stationary kernel does not take this form");
  for (int i = 0; i < nprob; ++i) {
    if (X2 && x2_every > 0 && i % x2_every == 0)
      FDQL_REQUIRE(wgrad_stat_add_rider(wa, i, wgrad_problem(WG_N, nx2, dW2 + (long long)i * WG_N * ldw2, ldw2, G + (long long)i * M * WG_N, WG_N,
                                                            X2 + (long long)i * M * ldx2, ldx2)), "narrow-input rider refused");
    if (G2)
      FDQL_REQUIRE(wgrad_stat_add_rider(wa, i, wgrad_problem(ng2, WG_N, dW3 + (long long)i * ng2 * ldw3, ldw3, G2 + (long long)i * M * ldg2, ldg2,
                                                            X + (long long)i * M * WG_N, WG_N)), "narrow-output rider refused");
  }
  FDQL_REQUIRE(wgrad_stat_balance(wa), "no workgroups to deal");
  hipError_t e = wgrad_stat_launch(wa, (hipStream_t)stream);
  if (e != hipSuccess) { set_error("wgrad_stat launch: %s", hipGetErrorString(e)); return FDQL_EHIP; }
  return 0;
}

int fdql_debug_rowgemm_life(uint64_t *out, int32_t cap) { return rowgemm_read_life((unsigned long long *)out, cap); }

/* Test hook for the persistent row-block kernel (rowgemm.hip): `ninst` instances of one layer, instance i using rows
 * [i*M, (i+1)*M) of every activation / output array and its own weights W0[i] [256 x 256] (ks: [k][n], else [n][k] with
 * row stride ldw0), W1[i] / W2[i] ([256 x k1] rows of stride k1, or K-strided [k1 x 256]).  fz_h: fused head dgrad
 * (GemmProblem::fz_*): A0's rows are OUTPUT, formed from fz_h, A1 (= dY, k1 = 2) and fz_w[i] [2 x fz_ldw].  Returns
 * FDQL_EINVAL when the kernel does not take the form (the caller's fallback is the tile kernels). */
int fdql_test_rowgemm(const float *A0, const float *A1, int32_t k1, const float *A2, int32_t k2, const float *W0, int32_t ldw0,
                      const float *W1, const float *W2, const float *bias, float *C, float *C2, const float *ref, float *colsum,
                      const float *hf_w, int32_t hf_ldw, int32_t hf_q, float *hf_out, float *hf_out2, int32_t M, int32_t ninst,
                      int32_t ks, int32_t grad, int32_t dual, int32_t planes, const float *fz_h, const float *fz_w, int32_t fz_ldw,
                      float *fz_colsum, void *stream) {
  std::vector<GemmProblem> probs;
  for (int i = 0; i < ninst; ++i) {
    GemmProblem p;
    memset(&p, 0, sizeof(p));
    p.M = M; p.N = RG_N; p.ksplit = 1; p.emit_seg = -1;
    const long long r0 = (long long)i * M;
    p.C = C + r0 * RG_N; p.ldc = RG_N;
    auto seg = [&](const float *Aptr, int lda, const float *W, int ldw, int K) {
      GemmSeg &sg = p.seg[p.nseg++];
      sg.A = Aptr; sg.lda = lda; sg.a_kc = 1; sg.B = W; sg.ldb = ldw; sg.b_kc = ks ? 0 : 1; sg.K = K;
    };
    seg(A0 + r0 * RG_KMAIN, RG_KMAIN, W0 + (long long)i * RG_KMAIN * ldw0 * (ks ? 1 : 1), ks ? RG_N : ldw0, RG_KMAIN);
    if (A1) seg(A1 + r0 * k1, k1, W1 + (long long)i * RG_N * k1, ks ? RG_N : k1, k1);
    if (A2) seg(A2 + r0 * k2, k2, W2 + (long long)i * RG_N * k2, ks ? RG_N : k2, k2);
    if (grad) {
      p.epi = EPI_LRELU_GRAD; p.ref = ref + r0 * RG_N; p.ldref = RG_N;
      p.colsum = colsum + (long long)i * (M / 64) * RG_N;
    } else {
      p.epi = EPI_LRELU; p.bias = bias + (long long)i * RG_N;
    }
    if (dual) { p.emit_seg = p.nseg - 2; p.C2 = C2 + r0 * RG_N; p.ldc2 = RG_N; }
    if (fz_h) {   // the main segment's A block is formed from (fz_h, A1 = dY, fz_w) and lands in A0's rows
      p.fz_h = fz_h + r0 * RG_N; p.fz_w = fz_w + (long long)i * 2 * fz_ldw; p.fz_ldw = fz_ldw;
      p.fz_out = const_cast<float *>(A0) + r0 * RG_KMAIN; p.fz_colsum = fz_colsum + (long long)i * (M / 64) * RG_N;
    }
    if (hf_w) {
      p.hf_w = hf_w + (long long)i * hf_q * hf_ldw; p.hf_ldw = hf_ldw; p.hf_q = hf_q;
      p.hf_out = hf_out + (long long)i * planes * M * hf_q;
      if (dual) p.hf_out2 = hf_out2 + (long long)i * planes * M * hf_q;
    }
    probs.push_back(p);
  }
  // the forward forms go to the weight-stationary kernel (wstat.hip) unless FDQL_WSTAT=0, as in the update's plan
  RowsLaunch rl;
  if (wstat_from_problems(probs.data(), (int)probs.size(), rl.wa)) rl.ws = true;
  else FDQL_REQUIRE(rowgemm_from_problems(probs.data(), (int)probs.size(), rl.rg), "the row-block kernels do not take this form");
  if (rl.ws && rl.wa.grad) {
    // the weight-stationary dgrad form writes wstat_colsum_rows() partial rows per instance, not one per 64 rows: clear the
    // caller's [ninst, M / 64, 256] buffers so that their sum over the row axis is the total either way
    const size_t bytes = (size_t)ninst * (M / 64) * RG_N * sizeof(float);
    FDQL_HIP(hipMemsetAsync(colsum, 0, bytes, (hipStream_t)stream));
    if (fz_colsum) FDQL_HIP(hipMemsetAsync(fz_colsum, 0, bytes, (hipStream_t)stream));
  }
  if (rl.ws && rl.wa.hf_q && getenv("FDQL_TEST_HF_PRESUM")) {
    // the planes summed inside the kernel (WsArgs::hf_presum): plane 0 of each instance holds the total, the caller's other
    // planes are cleared so that its sum over the plane axis is the total either way
    const size_t bytes = (size_t)ninst * planes * M * hf_q * sizeof(float);
    FDQL_HIP(hipMemsetAsync(hf_out, 0, bytes, (hipStream_t)stream));
    if (dual) FDQL_HIP(hipMemsetAsync(hf_out2, 0, bytes, (hipStream_t)stream));
    rl.wa.hf_presum = 1;
  }
  hipError_t e = rl.launch((hipStream_t)stream);
  if (e != hipSuccess) { set_error("row-block launch: %s", hipGetErrorString(e)); return FDQL_EHIP; }
  return 0;
}

}  // extern "C"
