// Host-side plan builder and runner of one SAC/TQC gradient step.
//
// fdql_agent_create() lays out the parameter arenas with the reference's state_dict names;
// the first fdql_agent_update() (or a change of batch pointers) builds a fixed list of
// launch stages — grouped GEMM problem tables, skinny-head tables, fused loss/policy
// kernels — uploads the tables once, and every later update just replays the launches.
// All per-step varying scalars (Adam step, bias corrections, lagged alpha) live in device
// memory (DevState), so the stage list is replayable without host-side changes.
#include "agent_internal.h"

namespace fdql {

static thread_local char g_err[1024] = "";
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static PlanSwitches g_switches;
const PlanSwitches &plan_switches() { return g_switches; }
// every environment variable of the library (common.h, PlanSwitches)
void plan_switches_refresh() {
  PlanSwitches w;
  auto is = [](const char *name, const char *val) { const char *e = getenv(name); return e && !strcmp(e, val); };
  auto set = [](const char *name) { return getenv(name) != nullptr; };
  if (const char *e = getenv("FDQL_CHAIN")) w.chain = !strcmp(e, "all") ? 3 : (!strcmp(e, "enc") ? 2 : (e[0] == '0' ? 0 : 1));
  if (const char *r = getenv("FDQL_ROWGEMM")) {
    if (r[0] == '0') w.rows_min_tiles = 1LL << 60;
    else if (!strcmp(r, "all")) { w.rows_min_tiles = 1; w.rows_all = true; }
    else if (atoi(r) > 1) w.rows_min_tiles = atoi(r);
  }
  w.rowdgrad = !is("FDQL_ROWDGRAD", "0");
  w.rowdgrad_chain = !set("FDQL_NO_ROWDGRAD_CHAIN");
  w.wgrad_stat = !is("FDQL_WGRAD_STAT", "0");
  w.wgrad_riders = !is("FDQL_WGRAD_RIDERS", "0");
  w.stream_wgrad = is("FDQL_STREAM_WGRAD", "0") ? 0 : (is("FDQL_STREAM_WGRAD", "2") ? 2 : 1);
  w.small_gemm = !is("FDQL_SMALL_GEMM", "0");
  w.colsum_stream = !set("FDQL_NO_COLSUM_STREAM");
  w.gate_masks = !set("FDQL_NO_GATE_MASKS");
  w.head_dgrad_masked = !set("FDQL_NO_HEAD_DGRAD_MASKED");
  w.head_fuse = !set("FDQL_NO_HEAD_FUSE");
  w.head_presum = !set("FDQL_NO_HEAD_PRESUM");
  w.dual = !set("FDQL_NO_DUAL");
  w.fuse_dpre1 = !set("FDQL_NO_FUSE_DPRE1");
  w.policy_dpre_fuse = !set("FDQL_NO_POLICY_DPRE_FUSE");
  w.small_folds = !set("FDQL_NO_SMALL_FOLDS");
  w.loss_wave = !is("FDQL_LOSS_WAVE", "0");
  w.gru_scan = !is("FDQL_GRU_SCAN", "0");
  w.act_fuse = !set("FDQL_ACT_NO_FUSE");
  w.implicit_conv = !set("FDQL_NO_IMPLICIT_CONV");
  w.graph = is("FDQL_GRAPH", "1");
  w.no_buckets = set("FDQL_NO_BUCKETS");
  w.force_buckets = set("FDQL_FORCE_BUCKETS");
  if (const char *pc = getenv("FDQL_PLAN_CACHE")) { const int v = atoi(pc); if (v >= 0 && v <= 256) w.plan_cache = v; }
  g_switches = w;
}

}  // namespace fdql

namespace {

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
    plan_switches_refresh();   // the environment, as it stands now (common.h, PlanSwitches)
    const PlanSwitches &w = plan_switches();
    // hipGraph replay: measured on ROCm 7.2 / MI355X 0.5-1.5 % SLOWER than the eager launch list at
    // T=50 and at T=2 - the step is bound by its kernels' own latency, the host stays ahead of the queue - so it is opt-in
    a->use_graph = w.graph;
    a->no_buckets = w.no_buckets;
    a->force_buckets = w.force_buckets;   // test hook: the two-bucket plan at world_size 1
    if (w.plan_cache >= 0) a->plan_cache_max = (size_t)w.plan_cache;
    a->rows_min_tiles = w.rows_min_tiles;
    if (w.rows_all) { a->wgrad_stat_min_tiles = 1; a->rowdgrad_min_blocks = 1; }   // "all": every persistent kernel whatever the size (tests)
    else if (w.rows_min_tiles >= (1LL << 40)) a->wgrad_stat_min_tiles = 1LL << 60;                                                  // FDQL_ROWGEMM=0: tile kernels only
    else if (w.rows_min_tiles != PlanSwitches().rows_min_tiles) a->wgrad_stat_min_tiles = 6 * w.rows_min_tiles;                       // FDQL_ROWGEMM=n
    if (!w.small_gemm) a->small_max_tiles = 0;
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
    if (plan_switches().wgrad_stat && plan_switches().stream_wgrad != 0 && nblk > 0 && nblk <= WG_MAX_INST && a->M % WG_BM == 0 &&
        a->M / a->nsplit <= STREAM_WGRAD_MAX_SLAB_ROWS &&   // (long slabs keep the tile kernels' narrow launches, which want the splits)
        hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) {
      // a data-parallel (two-bucket) plan launches the critics' blocks and the rest separately (at most one workgroup per slab
      // and block): sized to the larger launch, the critics' (config 2: 10 blocks -> 28 slabs; the 5-block launch of the
      // rest then runs 140 workgroups instead of 100)
      const int per_launch = a->bucketed() && nblk_critics > 0 ? std::max(nblk_critics, nblk - nblk_critics) : nblk;
      const int want = std::max(8, pr.multiProcessorCount / per_launch + 3);
      if (want < a->nsplit && a->wgrad_stat_pays(per_launch)) a->nsplit = want;
    }
  }
  {
    bool ws_ok = true;
    for (const MlpDesc &d : a->critic) ws_ok = ws_ok && !d.hid.empty() && d.hid[0] == WS_N;
    ws_ok = ws_ok && c.latent == WS_N && a->M % WS_BM == 0 && c.n_critics <= WS_MAX_INST && c.n_quantiles <= 32 &&
            (long long)c.n_critics * (a->M / ROWS_BM) >= a->rows_min_tiles;
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

int fdql_agent_set_launch_mode(fdql_agent_t *a, int32_t graph) {
  FDQL_REQUIRE(a && (graph == 0 || graph == 1), "fdql_agent_set_launch_mode: graph must be 0 or 1");
  std::lock_guard<std::mutex> lk(a->mu);
  a->use_graph = graph != 0;   // captured graphs are kept: switching back and forth costs nothing
  return 0;
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
  static const char *shape_names[GEMM_NSHAPES] = {"128x128", "128x32", "32x128", "64x128", "64x64dual", "64x64", "64x64hf", "small"};
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
      }
    } else {
      snprintf(out[cnt].name, sizeof(out[cnt].name), "%s%s",
               st.kind == ST_SKINNY_WGRAD ? (st.stream ? "nwgrad:" : "colsum:")
                                          : (st.kind == ST_CHAIN ? "chain:" : (st.kind == ST_WGRAD_STAT ? "wgstat:" : (st.prof ? st.prof : (st.mfma ? "conv:" : "k:")))),
               st.name.c_str());
    }
    out[cnt].ms = ms;
    out[cnt].flops = flops;
    out[cnt].bytes = bytes;
  }
  for (auto &e : ev) (void)hipEventDestroy(e);
  return cnt;
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

}  // extern "C"
