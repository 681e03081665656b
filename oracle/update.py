"""CPU oracle (PyTorch-CPU, fp32) for the SAC/TQC update.  TEST INFRASTRUCTURE ONLY.

A functional restatement of the reference's trainer step written from its text; weights
live in a flat ``dict`` keyed by the reference's ``state_dict`` names so golden vectors can
be loaded directly.  Citations are path:line under /root/reference.

    train_step            franQ/Agent/deepQlearning.py:105-127
    losses                franQ/Agent/deepQlearning.py:198-258
    encoder               franQ/Agent/components/encoder.py:52-67
    skip_head_mlp         franQ/Agent/models/mlp.py:64-94          (quirk q8: always LeakyReLU(0.01))
    gaussian_policy       franQ/Agent/models/gaussian_mlp.py:15-39 (quirk q9: eps 1e-4)
    gumbel_policy         franQ/Agent/models/gumbel_mlp.py:7-54 + torch RelaxedOneHotCategorical
    tqc target / loss     franQ/Agent/components/distributional_soft_actor_critic.py:40-103
    sac target / loss     franQ/Agent/components/soft_actor_critic.py:63-134
    actor / alpha loss    franQ/Agent/components/soft_actor_critic.py:136-154
    polyak                franQ/Agent/utils/common.py:10-19
    adam                  torch.optim.Adam defaults (deepQlearning.py:100-103), restated explicitly

Backward uses torch autograd on this restated forward (it is the independent check of the
hand-derived backward in the HIP kernels); Adam and polyak are written out.
"""
import math
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


@dataclass
class Spec:
    obs: int
    act: int                      # action dim, or number of actions when discrete
    goal: int = 0
    discrete: bool = False
    C: int = 2                    # num_critics            (conf.py:66)
    Q: int = 10                   # num_q_predictions      (conf.py:67)
    latent: int = 256             # latent_state_dim       (conf.py:68)
    enc_features: int = 256       # EncoderConf.hidden_features (conf.py:81)
    enc_hidden: Tuple[int, ...] = (256,)
    joint_hidden: Tuple[int, ...] = (256,)
    pi_hidden: Tuple[int, ...] = (256,)
    critic_hidden: Tuple[int, ...] = (256, 256)
    distributional: bool = True
    lowerbound: bool = True       # use_nStep_lowerbounds
    max_entropy: bool = True      # use_max_entropy_q
    hard_updates: bool = False
    gamma: float = 0.99
    tau: float = 5e-2
    lr: float = 3e-4
    init_log_alpha: float = -2
    drop: float = 0.2             # top_quantiles_to_drop
    T: int = 50
    B: int = 256
    world_size: int = 1           # data-parallel ranks: loss is normalised by B * world_size
    bootstrap: bool = False       # use_bootstrap_minibatch_nstep (SAC-min with lower bounds only)
    burn_in: int = 0              # int(T * burn_in_portion) when EncoderConf.use_burn_in, else 0
    # Pixel observations (BASELINE config 5).  NO REFERENCE EXISTS for this encoder (encoder.py:16-23 is dead code):
    # a small strided conv stack of this build's own design, LeakyReLU(0.01) like every other hidden layer (quirk q8),
    # input scaled by 1/255, output flattened in (y, x, channel) order and fed to the obs MLP next to obs_1d.
    img: Tuple[int, ...] = ()     # (C, H, W) of xp["obs_2d"], or () = none
    conv: Tuple[Tuple[int, int, int], ...] = ()   # per layer (out_channels, kernel, stride)
    gru: str = ""                 # "" = feed-forward joiner; else EncoderConf.JoinerModeEnum.gru with the latent-state
                                  # training mode "zero" | "learned" | "store" (encoder.py:40-42, 78-94)

    @property
    def conv_shapes(self):        # per layer (Cin, H_in, W_in, Cout, k, s, H_out, W_out)
        out, (c, h, w) = [], self.img if self.img else (0, 0, 0)
        for co, k, st in self.conv:
            ho, wo = (h - k) // st + 1, (w - k) // st + 1
            out.append((c, h, w, co, k, st, ho, wo))
            c, h, w = co, ho, wo
        return out

    @property
    def conv_features(self):
        if not self.img:
            return 0
        cs = self.conv_shapes
        return cs[-1][3] * cs[-1][6] * cs[-1][7] if cs else self.img[0] * self.img[1] * self.img[2]

    @property
    def enc_in(self):             # encoder.py:26-32 (+ the flattened conv features)
        return self.obs + 2 * self.goal + self.conv_features

    @property
    def act_feat(self):           # width of the action block fed to the critic
        return self.act

    @property
    def pi_out(self):             # gaussian_mlp.py:10 (2A) / gumbel_mlp.py:10 (n)
        return self.act if self.discrete else 2 * self.act

    @property
    def Nq(self):
        return self.C * self.Q

    @property
    def n_drop(self):
        return int(self.drop * self.Nq)

    @property
    def target_entropy(self):     # soft_actor_critic.py:42
        return -float(self.act)


# ---------------------------------------------------------------------------------------
# parameter naming (reference state_dict layout)
# ---------------------------------------------------------------------------------------
def mlp_shapes(prefix, din, hidden, dout):
    """SkipHeadMLP parameter shapes: mlp.py:76-86."""
    shapes = []
    prev = din
    for i, h in enumerate(hidden):
        shapes.append((f"{prefix}.feature_extractor.{i}.0.weight", (h, prev)))
        shapes.append((f"{prefix}.feature_extractor.{i}.0.bias", (h,)))
        prev = h
    shapes.append((f"{prefix}.head.weight", (dout, din + sum(hidden))))
    shapes.append((f"{prefix}.head.bias", (dout,)))
    return shapes


def net_shapes(spec: Spec, which: str):
    if which == "encoder.obs":
        conv = []
        for i, (ci, _, _, co, k, _, _, _) in enumerate(spec.conv_shapes):
            conv += [(f"encoder.visible_layer_encoders.obs_2d.conv.{i}.weight", (co, ci * k * k)),
                     (f"encoder.visible_layer_encoders.obs_2d.conv.{i}.bias", (co,))]
        return conv + mlp_shapes("encoder.visible_layer_encoders.obs_1d", spec.enc_in, spec.enc_hidden, spec.enc_features)
    if which == "encoder.joiner":
        if spec.gru:    # nn.GRU(hidden_features, latent, num_layers=1) + the learnable initial state (encoder.py:41-42)
            L, F_ = spec.latent, spec.enc_features
            return [("encoder.joiner.weight_ih_l0", (3 * L, F_)), ("encoder.joiner.weight_hh_l0", (3 * L, L)),
                    ("encoder.joiner.bias_ih_l0", (3 * L,)), ("encoder.joiner.bias_hh_l0", (3 * L,)),
                    ("encoder.hidden_state", (L,))]
        return mlp_shapes("encoder.joiner", spec.enc_features, spec.joint_hidden, spec.latent)
    if which in ("actor", "actor_target"):
        return mlp_shapes(f"actor_critic.{which}", spec.latent, spec.pi_hidden, spec.pi_out)
    if which in ("critic", "critic_target", "critic_frozen"):
        out = []
        for k in range(spec.C):
            out += mlp_shapes(f"actor_critic.{which}.nets.{k}", spec.latent + spec.act_feat, spec.critic_hidden, spec.Q)
        return out
    raise KeyError(which)


def trainable_names(spec: Spec) -> List[str]:
    """fast_params: encoder + actor + critic + log_alpha (deepQlearning.py:47-62,
    soft_actor_critic.py:44-52)."""
    names = [n for n, _ in net_shapes(spec, "encoder.obs") + net_shapes(spec, "encoder.joiner")]
    names += [n for n, _ in net_shapes(spec, "actor")]
    names += [n for n, _ in net_shapes(spec, "critic")]
    names.append("actor_critic.log_alpha")
    return names


def all_shapes(spec: Spec):
    s = net_shapes(spec, "encoder.obs") + net_shapes(spec, "encoder.joiner")
    s += [("actor_critic.log_alpha", ())]
    for w in ("critic", "critic_target", "critic_frozen", "actor", "actor_target"):
        s += net_shapes(spec, w)
    return s


def init_params(spec: Spec, seed=0) -> Dict[str, Tensor]:
    """xavier_uniform(gain 1) weights, zero biases (mlp.py:5-8,86); targets = hard copies
    (soft_actor_critic.py:34,38); critic_frozen has its own init (never copied until the
    first actor_loss).  Not RNG-compatible with the reference (module construction order
    differs); parity runs load the golden ``init`` instead."""
    g = torch.Generator().manual_seed(seed)
    p = {}
    for name, shape in all_shapes(spec):
        if name.startswith("encoder.joiner.") and spec.gru:       # nn.GRU.reset_parameters: U(-1/sqrt(L), 1/sqrt(L))
            k = 1.0 / math.sqrt(spec.latent)
            p[name] = (torch.rand(shape, generator=g, dtype=torch.float32) * 2 - 1) * k
        elif name == "encoder.hidden_state":                       # encoder.py:42, 117: torch.rand(out_features)
            p[name] = torch.rand(shape, generator=g, dtype=torch.float32)
        elif name.endswith("weight"):
            fan_out, fan_in = shape
            a = math.sqrt(6.0 / (fan_in + fan_out))
            p[name] = (torch.rand(shape, generator=g, dtype=torch.float32) * 2 - 1) * a
        elif name.endswith("bias"):
            p[name] = torch.zeros(shape, dtype=torch.float32)
        else:
            p[name] = torch.tensor(float(spec.init_log_alpha), dtype=torch.float32)
    for src, dst in (("critic", "critic_target"), ("actor", "actor_target")):
        for n, _ in net_shapes(spec, src):
            p[n.replace(f".{src}.", f".{dst}.")] = p[n].clone()
    return p


# ---------------------------------------------------------------------------------------
# networks
# ---------------------------------------------------------------------------------------
# Kink bookkeeping for the fp64 three-way gradient comparison (tests/test_gpu_parity.py): LeakyReLU and relu(mc - q)
# make the gradient discontinuous in the activations, so two evaluations that agree to 1e-7 in the forward pass can
# pick different branches for a unit that sits on its kink.  KINKS["record"] (a dict, when set) receives every
# pre-activation under "<prefix>.<layer>"; KINKS["force"] (a dict of bool tensors, when set) replaces the branch
# choice x > 0 of the units it names by the given pattern (e.g. the one the GPU took).
KINKS = {"record": None, "force": None}


def _leaky(pre, key):
    if KINKS["record"] is not None:
        KINKS["record"][key] = pre.detach()
    force = KINKS["force"]
    if force is not None and key in force:
        return torch.where(force[key], pre, 0.01 * pre)
    return F.leaky_relu(pre, 0.01)


def _relu_kink(x, key):
    if KINKS["record"] is not None:
        KINKS["record"][key] = x.detach()
    force = KINKS["force"]
    if force is not None and key in force:
        return torch.where(force[key], x, torch.zeros_like(x))
    return x.relu()


def skip_head_mlp(p, prefix, x, n_hidden):
    """mlp.py:88-94: h_i = LeakyReLU_0.01(W_i h_{i-1} + b_i); out = W_head cat(x, h_1..h_n) + b."""
    feats = [x]
    h = x
    for i in range(n_hidden):
        h = _leaky(F.linear(h, p[f"{prefix}.feature_extractor.{i}.0.weight"],
                            p[f"{prefix}.feature_extractor.{i}.0.bias"]), f"{prefix}.{i}")
        feats.append(h)
    return F.linear(torch.cat(feats, dim=-1), p[f"{prefix}.head.weight"], p[f"{prefix}.head.bias"])


def conv_features(p, spec: Spec, frames):
    """Pixel encoder of this build (no reference): frames [..., C, H, W] with values 0..255 -> x/255 ->
    conv(k, stride) + LeakyReLU(0.01) per layer (weights kept as [Cout, Cin*k*k]) -> flattened in (y, x, channel) order [..., Ho*Wo*Cout]."""
    lead = frames.shape[:-3]
    x = frames.reshape((-1,) + tuple(spec.img)) * (1.0 / 255.0)
    for i, (ci, _, _, co, k, st, _, _) in enumerate(spec.conv_shapes):
        w = p[f"encoder.visible_layer_encoders.obs_2d.conv.{i}.weight"]
        # K of the stored [Cout, K] weight runs (c, ky, kx) for the first layer (NCHW frames) and (ky, kx, c) after it
        # (NHWC feature maps): the order in which the device's im2col lays a window out
        w = w.view(co, ci, k, k) if i == 0 else w.view(co, k, k, ci).permute(0, 3, 1, 2)
        x = F.leaky_relu(F.conv2d(x, w, p[f"encoder.visible_layer_encoders.obs_2d.conv.{i}.bias"], stride=st), 0.01)
    return x.permute(0, 2, 3, 1).reshape(lead + (-1,))


def gru_cell(p, x_t, h):
    """One step of torch.nn.GRU (gate order r, z, n):
    r = s(W_ir x + b_ir + W_hr h + b_hr); z = s(W_iz x + b_iz + W_hz h + b_hz);
    n = tanh(W_in x + b_in + r * (W_hn h + b_hn)); h' = (1 - z) * n + z * h."""
    gi = F.linear(x_t, p["encoder.joiner.weight_ih_l0"], p["encoder.joiner.bias_ih_l0"])
    gh = F.linear(h, p["encoder.joiner.weight_hh_l0"], p["encoder.joiner.bias_hh_l0"])
    i_r, i_z, i_n = gi.chunk(3, -1)
    h_r, h_z, h_n = gh.chunk(3, -1)
    r = torch.sigmoid(i_r + h_r)
    z = torch.sigmoid(i_z + h_z)
    n = torch.tanh(i_n + r * h_n)
    return (1 - z) * n + z * h


def encoder(p, spec: Spec, xp, h0=None, return_hidden=False):
    """encoder.py:52-67: obs MLP, then the feed-forward joiner or (spec.gru) one GRU layer scanned over the
    leading (time) axis from ``h0`` [B, L] (zeros when None)."""
    parts = [xp["obs_1d"]] if spec.obs else []
    if spec.goal:
        parts += [xp["achieved_goal"], xp["desired_goal"]]
    if spec.img:
        parts.append(conv_features(p, spec, xp["obs_2d"]))
    x = torch.cat(parts, dim=-1)
    e = skip_head_mlp(p, "encoder.visible_layer_encoders.obs_1d", x, len(spec.enc_hidden))
    if not spec.gru:
        return skip_head_mlp(p, "encoder.joiner", e, len(spec.joint_hidden))
    h = h0 if h0 is not None else torch.zeros(e.shape[1:-1] + (spec.latent,), dtype=e.dtype)
    ys = []
    for t in range(e.shape[0]):
        h = gru_cell(p, e[t], h)
        ys.append(h)
    y = torch.stack(ys, 0)
    return (y, h) if return_hidden else y


def ensemble(p, spec: Spec, which, x):
    """mlp.py:105-108: concat of the C critics' outputs on the last dim."""
    return torch.cat([skip_head_mlp(p, f"actor_critic.{which}.nets.{k}", x, len(spec.critic_hidden))
                      for k in range(spec.C)], dim=-1)


LOG_SQRT_2PI = math.log(math.sqrt(2 * math.pi))


def gaussian_policy(p, spec: Spec, which, s, eps):
    """gaussian_mlp.py:15-39 with the N(0,1) draw ``eps`` supplied by the caller."""
    logits = skip_head_mlp(p, f"actor_critic.{which}", s, len(spec.pi_hidden))
    mean, log_std = torch.chunk(logits, 2, dim=-1)
    log_std = torch.clamp(log_std, min=-20.0, max=2.0)
    std = log_std.exp()
    x = mean + eps * std                                    # Normal.rsample
    logp = -((x - mean) ** 2) / (2 * std ** 2) - std.log() - LOG_SQRT_2PI   # Normal.log_prob
    a = torch.tanh(x)
    logp = logp - torch.log((1 - a.pow(2)) + 1e-4)
    return a, logp.sum(-1, keepdim=True)


def gumbel_policy(p, spec: Spec, which, s, u):
    """gumbel_mlp.py:13-21,40-54 on top of torch's ExpRelaxedCategorical.rsample at
    temperature 1, with the U(0,1) draw ``u`` supplied by the caller."""
    logits = skip_head_mlp(p, f"actor_critic.{which}", s, len(spec.pi_hidden))
    norm = logits - logits.logsumexp(dim=-1, keepdim=True)      # Categorical normalisation
    tiny = torch.finfo(torch.float32).eps                       # clamp_probs of the reference's float32 draw
    uc = u.clamp(min=tiny, max=1 - tiny)
    gumbels = -((-(uc.log())).log())
    scores = (norm + gumbels) / 1.0
    relaxed = (scores - scores.logsumexp(dim=-1, keepdim=True)).exp()
    hard = F.one_hot(torch.argmax(relaxed, dim=-1), logits.shape[-1]).to(relaxed.dtype)
    st = (hard - relaxed).detach() + relaxed                    # straight-through
    logp = -torch.sum(-st * F.log_softmax(norm, -1), -1, keepdim=True)
    return st, logp


def policy(p, spec, which, s, noise):
    return gumbel_policy(p, spec, which, s, noise) if spec.discrete else gaussian_policy(p, spec, which, s, noise)


def act(p, spec: Spec, xp, noise):
    """deepQlearning.py:155-187: encoder.forward_eval (encoder.py:52-76) -> online actor ->
    explore/exploit select by ``xp["exploit_mask"]`` ([rows, 1] bool), with the policy's noise draw
    supplied by the caller.  Returns (action, log_prob, explore_action, exploit_action)."""
    with torch.no_grad():
        hidden = None
        if spec.gru:   # forward_eval: one GRU step from xp["agent_state"] (encoder.py:72-76; runner.py:103-106)
            s, hidden = encoder(p, spec, {k: v.unsqueeze(0) for k, v in xp.items()}, h0=xp.get("agent_state"),
                                return_hidden=True)
            s = s[0]
        else:
            s = encoder(p, spec, xp)
        explore, logp = policy(p, spec, "actor", s, noise)
        logits = skip_head_mlp(p, "actor_critic.actor", s, len(spec.pi_hidden))
        if spec.discrete:                                       # deepQlearning.py:175-178
            explore = explore.argmax(-1, True)
            exploit = logits.argmax(-1, True)                   # gumbel_mlp.py:21 returns the raw logits
        else:
            exploit = torch.tanh(torch.chunk(logits, 2, dim=-1)[0])   # gaussian_mlp.py:38 tanh(mean)
        mask = xp["exploit_mask"]
        action = (exploit * mask) + (explore * torch.logical_not(mask))
        return (action, logp, explore, exploit, hidden) if spec.gru else (action, logp, explore, exploit)


# ---------------------------------------------------------------------------------------
# losses
# ---------------------------------------------------------------------------------------
def quantile_huber(q, y):
    """distributional_soft_actor_critic.py:90-103 (quirk q4: tau over the POOLED atoms)."""
    delta = y[..., None, :] - q[..., None]
    ad = delta.abs()
    huber = torch.where(ad > 1, ad - 0.5, delta ** 2 * 0.5)
    n = q.shape[-1]
    tau = torch.arange(n, dtype=q.dtype) / n + 1 / 2 / n
    tau = tau.view(*([1] * (q.dim() - 1)), n, 1)
    return (torch.abs(tau - (delta < 0).to(q.dtype)) * huber).mean((-1, -2))


def losses(p, spec: Spec, xp, noise_target, noise_actor, alpha):
    """deepQlearning.py:198-249.  ``alpha`` is last step's exp(log_alpha) (quirk q5).
    Returns the scalar loss and a dict of the intermediates the goldens hold."""
    aux = {}
    mask = (xp["task_done"] == 0)                                          # :201
    contig = (xp["episode_step"][1:] == xp["episode_step"][:-1] + 1) & mask[:-1]   # :202-203
    action = xp["action"]
    if spec.discrete:                                                      # :206-210
        action = torch.eye(spec.act, dtype=xp["reward"].dtype)[action.view(action.shape[:-1]).long()]
    h0 = None
    if spec.gru:                                                           # encoder.py:78-94 (forward_train)
        contig = torch.cumprod(contig.to(torch.int64), dim=0).bool()       # a window is valid up to its first break
        if spec.gru == "store":
            h0 = xp["agent_state"][0]
        elif spec.gru == "learned":
            h0 = p["encoder.hidden_state"].view(1, -1).repeat(contig.shape[1], 1)
    state = encoder(p, spec, xp, h0=h0)                                    # :213
    s_cur, s_nxt = state[:-1], state[1:]                                   # :251-258
    with torch.no_grad():                                                  # q_loss target
        a_n, logp_n = policy(p, spec, "actor_target", s_nxt, noise_target)
        z = ensemble(p, spec, "critic_target", torch.cat((s_nxt, a_n), -1))
        if spec.distributional:                                            # distributional…:50-58
            zs, _ = torch.sort(z, dim=-1)
            tq = zs[..., :-spec.n_drop]                                    # quirk q6: empty when n_drop == 0
            if spec.max_entropy:
                tq = tq + alpha * (-logp_n)
        else:                                                              # soft_actor_critic.py:73-78
            tq = z + alpha * (-logp_n) if spec.max_entropy else z
            tq, _ = torch.min(tq, dim=-1, keepdim=True)
        td = xp["reward"][1:] + mask[1:] * spec.gamma * tq
    q = ensemble(p, spec, "critic", torch.cat((s_cur, action[:-1]), -1))
    mc = xp["mc_return"][1:] if "mc_return" in xp else None
    if spec.distributional:
        q_loss = quantile_huber(q, td).unsqueeze(-1)                       # distributional…:70
        if spec.lowerbound:
            q_loss = q_loss + _relu_kink(mc - q, "lowerbound").mean(-1, keepdim=True)       # :76-79
    else:
        ql = F.smooth_l1_loss(q, td.expand_as(q), reduction="none")       # soft_actor_critic.py:88
        if spec.lowerbound:                                                # :93-97
            lb = (mc - q).relu()
            ql = ql * (lb == 0) + lb
        q_loss = ql.mean(-1, keepdim=True)                                 # :134
    # actor / alpha, soft_actor_critic.py:136-154 (critic_frozen == critic, :142)
    pi, logp = policy(p, spec, "actor", s_cur, noise_actor)
    frozen = {k.replace(".critic.", ".critic_frozen."): v.detach() for k, v in p.items() if ".critic.nets." in k}
    qpi = ensemble(frozen, spec, "critic_frozen", torch.cat((s_cur.detach(), pi), -1)).mean(-1, keepdim=True)
    pi_loss = -(alpha * (-logp)) - qpi
    alpha_loss = -(p["actor_critic.log_alpha"] * (spec.target_entropy - (-logp)).detach())
    w = contig.to(state.dtype)
    if spec.burn_in:                                                       # deepQlearning.py:219-220
        w = w.clone()
        w[:spec.burn_in] = 0
    loss = ((q_loss + pi_loss + alpha_loss) * w).sum(0) / (w.sum(0) + 1e-4)   # :222-224
    loss = loss.sum() / (loss.numel() * spec.world_size)                   # :225 (.mean over B)
    if spec.bootstrap:       # soft_actor_critic.py:102-132, deepQlearning.py:226-228
        # discounted reward over the whole window + bootstrap from the last TD target: lower bound on q(t=0)
        assert not spec.distributional and spec.lowerbound, "the reference only forms this term in SoftActorCritic.q_loss"
        g_pow = spec.gamma ** torch.arange(spec.T - 1, dtype=q.dtype).view(-1, 1, 1)
        mb_return = (xp["reward"][1:] * g_pow).sum(0)
        mb_mask = mask[1:].to(q.dtype).prod(0)
        boot = mb_mask * ((mb_return + (spec.gamma ** (spec.T - 1)) * td[-1]) - q[0]).relu()      # [B, Nq]
        loss = loss + (boot * w.prod(0)).mean() / spec.world_size
        aux["bootstrap_lowerbound"] = boot
    loss = loss / spec.T                                                   # :249
    aux.update(state=state, next_action=a_n, next_log_pi=logp_n, next_z=z, td_target=td, q_pred=q,
               q_loss=q_loss, pi=pi, log_pi=logp, q_frozen=None, pi_loss=pi_loss, alpha_loss=alpha_loss,
               is_contiguous=w, qpi=qpi)
    return loss, aux


# ---------------------------------------------------------------------------------------
# optimiser / targets
# ---------------------------------------------------------------------------------------
def adam_update(param, grad, m, v, step, lr, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam (defaults; single-tensor path), restated: returns new (param, m, v)."""
    m = m + (grad - m) * (1 - b1)                      # exp_avg.lerp_(grad, 1 - beta1)
    v = v * b2 + (grad * grad) * (1 - b2)              # exp_avg_sq.mul_(b2).addcmul_(g, g, 1 - b2)
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    step_size = lr / bc1
    denom = v.sqrt() / math.sqrt(bc2) + eps
    return param - step_size * (m / denom), m, v


@dataclass
class TrainState:
    params: Dict[str, Tensor]
    adam_m: Dict[str, Tensor] = field(default_factory=dict)
    adam_v: Dict[str, Tensor] = field(default_factory=dict)
    step: int = 0
    alpha: float = None           # curr_alpha carried between steps (soft_actor_critic.py:41,152)


def new_state(spec: Spec, params) -> TrainState:
    st = TrainState(params={k: v.clone() for k, v in params.items()})
    for n in trainable_names(spec):
        st.adam_m[n] = torch.zeros_like(st.params[n])
        st.adam_v[n] = torch.zeros_like(st.params[n])
    st.alpha = float(math.exp(float(st.params["actor_critic.log_alpha"])))   # soft_actor_critic.py:41 (float64 exp)
    return st


def train_step(st: TrainState, spec: Spec, xp, noise_target, noise_actor):
    """deepQlearning.py:105-127 for one shard: loss -> backward -> Adam -> polyak."""
    names = trainable_names(spec)
    leaves = {}
    for k, v in st.params.items():
        leaves[k] = v.detach().clone().requires_grad_(k in names)
    alpha = torch.tensor(st.alpha, dtype=torch.float32)
    loss, aux = losses(leaves, spec, xp, noise_target, noise_actor, alpha)
    grads = torch.autograd.grad(loss, [leaves[n] for n in names] + [aux["q_pred"]], allow_unused=True)
    aux["dq_pred"] = grads[-1]
    grads = {n: (g if g is not None else torch.zeros_like(leaves[n])) for n, g in zip(names, grads[:-1])}
    # critic_frozen <- critic BEFORE the optimiser step (soft_actor_critic.py:142)
    for k in list(st.params):
        if ".critic.nets." in k:
            st.params[k.replace(".critic.", ".critic_frozen.")] = st.params[k].clone()
    # curr_alpha for the NEXT step uses log_alpha before this step's Adam (:152)
    st.alpha = float(torch.exp(st.params["actor_critic.log_alpha"]))
    st.step += 1
    for n in names:
        st.params[n], st.adam_m[n], st.adam_v[n] = adam_update(st.params[n], grads[n], st.adam_m[n], st.adam_v[n],
                                                                st.step, spec.lr)
    for src, dst in (("actor", "actor_target"), ("critic", "critic_target")):   # soft_actor_critic.py:54-60
        for k in list(st.params):
            if f".{src}." in k:
                kt = k.replace(f".{src}.", f".{dst}.")
                if spec.hard_updates:
                    st.params[kt] = st.params[k].clone()
                else:
                    st.params[kt] = st.params[kt] * (1.0 - spec.tau) + st.params[k] * spec.tau
    aux["grad"] = grads
    return loss.detach(), aux


def grads_in(dtype, spec: Spec, params, xp, noise_target, noise_actor, alpha, force=None):
    """loss, d loss / d theta, d loss / d q_pred and every pre-activation of ONE evaluation of ``losses`` in ``dtype``
    (torch.float64: the arbiter of the three-way comparison oracle-f32 / GPU / f64).  ``force``: optional
    {"<prefix>.<layer>" | "lowerbound": bool tensor} branch pattern to impose on the kinks (see KINKS)."""
    names = trainable_names(spec)
    cast = lambda v: v.detach().to(dtype) if v.is_floating_point() else v.detach()
    leaves = {k: cast(v).clone().requires_grad_(k in names) for k, v in params.items()}
    xpd = {k: cast(v) for k, v in xp.items()}
    rec = {}
    KINKS["record"], KINKS["force"] = rec, force
    try:
        loss, aux = losses(leaves, spec, xpd, cast(noise_target), cast(noise_actor), torch.tensor(float(alpha), dtype=dtype))
        g = torch.autograd.grad(loss, [leaves[n] for n in names] + [aux["q_pred"]], allow_unused=True)
    finally:
        KINKS["record"], KINKS["force"] = None, None
    grads = {n: (x if x is not None else torch.zeros_like(leaves[n])) for n, x in zip(names, g[:-1])}
    return loss.detach(), grads, g[-1], rec, aux
