// act() (franQ/Agent/deepQlearning.py:155-187): encoder -> joiner -> actor on a handful of rows, reading the trainer's arena
// in place (agent_internal.h).
#include "agent_internal.h"

extern "C" {

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
  const bool no_fuse = !plan_switches().act_fuse;   // (FDQL_ACT_NO_FUSE: every layer its own launch, policy its own)
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

}  // extern "C"
