#!/usr/bin/env python3
"""Generate golden input/output vectors by RUNNING THE REFERENCE (build container only).

Usage (from the repo root, in the build container where /root/reference is mounted):

    python tests/golden/make_golden.py

Writes tests/golden/*.npz.  The fixtures are DATA ONLY (inputs, externally captured noise,
and the reference's outputs); no reference source travels.  Every array is produced by
calling the reference's own classes:

  * franQ.Replay.replay_memory.ReplayMemory               (ring add / wrap / windows)
  * franQ.Replay.wrappers.nstep_return.NStepReturn        (discounted return at write)
  * franQ.Replay.wrappers.her.HindsightNStepReplay        (write-time hindsight relabel)
  * franQ.Replay.wrappers.her_vmap.HindsightVmapWrite/Read + nstep_return_vmap.NStepReturnVmap
    (K virtual goals per step, one picked at read time) - "shim-pinned": these files need jax, which the
    image lacks; they run on the numpy-backed stand-in of tests/golden/_refimport.py
  * franQ.Agent.deepQlearning.DeepQLearning.train_step    (loss, backward, Adam, polyak)

Noise is captured by wrapping the torch RNG entry points the reference's distributions
use (torch.distributions.normal._standard_normal, torch.rand inside
torch.distributions.relaxed_categorical) so the HIP kernels can replay it.
"""
import os
import sys
import random
from collections import OrderedDict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refimport  # noqa: E402

_refimport.install()
from _refimport import Space  # noqa: E402

import franQ  # noqa: E402,F401
from franQ.Replay.replay_memory import ReplayMemory  # noqa: E402
from franQ.Replay.wrappers.nstep_return import NStepReturn  # noqa: E402
from franQ.Replay.wrappers.her import HindsightNStepReplay  # noqa: E402
from franQ.Replay.wrappers.her_vmap import HindsightVmapWrite, HindsightVmapRead  # noqa: E402  (on the jax stand-in)
from franQ.Replay.wrappers.nstep_return_vmap import NStepReturnVmap  # noqa: E402
from franQ.Agent.deepQlearning import DeepQLearning  # noqa: E402
from franQ.Agent.conf import AgentConf  # noqa: E402

torch.set_num_threads(1)


def save(name, d):
    flat = {}

    def rec(prefix, v):
        if isinstance(v, dict):
            for k, x in v.items():
                rec(f"{prefix}/{k}" if prefix else str(k), x)
        else:
            if isinstance(v, torch.Tensor):
                v = v.detach().cpu().numpy()
            flat[prefix] = np.asarray(v)

    rec("", d)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **flat)
    print(f"wrote {path}: {len(flat)} arrays, {os.path.getsize(path) / 1024:.1f} KiB")


# --------------------------------------------------------------------------------------
# Replay ring
# --------------------------------------------------------------------------------------
def golden_ring():
    out = {}
    rng = np.random.RandomState(7)
    for case, (maxlen, n_add, T, B) in {"wrap8": (8, 11, 3, 4), "nowrap50": (50, 37, 5, 7),
                                       "wrap50": (50, 120, 5, 7)}.items():
        r = ReplayMemory(maxlen, B, T)
        rows = []
        for i in range(n_add):
            row = {
                "obs_1d": rng.standard_normal(3),  # float64 on purpose (gym obs are often f64)
                "action": rng.uniform(-1, 1, size=2).astype(np.float32),
                "reward": float(np.float32(rng.standard_normal())),
                "task_done": bool(rng.rand() < 0.2),
                "episode_done": bool(i % 5 == 4),
                "episode_step": int(i % 5),
                "idx": 0,
            }
            rows.append(row)
            r.add(row)
        starts = rng.randint(0, len(r) - T, B)
        win = r._temporal_sample_idxes(starts, len(r))
        flat_idx = rng.randint(0, len(r), B)
        flat = r[flat_idx]
        out[case] = {
            "maxlen": maxlen, "T": T, "B": B,
            "rows": {k: np.stack([np.asarray(x[k]) for x in rows]) for k in rows[0]},
            "len": len(r), "top": r._top,
            "memory": {k: v for k, v in r.memory.items()},
            "starts": starts, "window": win,
            "flat_idx": flat_idx, "flat": flat,
        }
    # OversampleError thresholds (replay_memory.py:57-58)
    r = ReplayMemory(100, 4, 5)
    thr = []
    for i in range(12):
        r.add({"x": np.zeros(1, np.float32)})
        try:
            r.temporal_sample()
            thr.append(1)
        except Exception as e:  # OversampleError
            assert type(e).__name__ == "OversampleError"
            thr.append(0)
    out["oversample"] = {"T": 5, "B": 4, "ok_after_n_adds": np.asarray(thr)}
    save("ring", out)


class Sink:
    """Stands where the ReplayMemory would: records what the write wrappers emit."""

    def __init__(self):
        self.rows = []

    def add(self, d):
        self.rows.append({k: np.array(v) for k, v in d.items()})

    def __len__(self):
        return len(self.rows)

    def stacked(self):
        keys = sorted(self.rows[0].keys())
        return {k: np.stack([np.asarray(r[k]).reshape(-1) for r in self.rows]) for k in keys}


def golden_nstep():
    out = {}
    rng = np.random.RandomState(11)
    cases = {
        "sparse_1000": (1000, 0.99, [1000]),        # the reference's own test (tests/test_replays.py:16-33)
        "dense_two_eps": (1000, 0.97, [7, 5]),
        "pop_quirk": (4, 0.9, [9, 3, 4]),           # q3: _pop fires once at len == n_step
        "single_step": (1000, 0.99, [1, 2]),
    }
    for name, (n_step, gamma, ep_lens) in cases.items():
        sink = Sink()
        w = NStepReturn(sink, n_step, gamma)
        inputs = []
        for e, L in enumerate(ep_lens):
            for i in range(L):
                if name == "sparse_1000":
                    rew = float(i == L - 1)
                else:
                    rew = float(np.float32(rng.standard_normal()))
                row = {"reward": rew, "episode_done": i == L - 1, "episode_step": i,
                       "obs_1d": rng.standard_normal(2).astype(np.float32)}
                inputs.append(row)
                w.add(dict(row))
        out[name] = {
            "n_step": n_step, "gamma": gamma, "ep_lens": np.asarray(ep_lens),
            "in": {k: np.stack([np.asarray(r[k]).reshape(-1) for r in inputs]) for k in inputs[0]},
            "out": sink.stacked(),
        }
    save("nstep", out)


def l2_sparse_reward(ag, dg, thr=0.25):
    """Sparse goal reward of the shape used by the reference envs
    (franQ/Env/bitflip.py:143-152, classic_goal.py:88-93): -1 until within thr, done when 0."""
    d = np.linalg.norm(np.asarray(ag, np.float32) - np.asarray(dg, np.float32))
    reward = np.float32(-1.0) if d > thr else np.float32(0.0)
    return reward, bool(reward == 0)


def golden_her():
    out = {}
    rng = np.random.RandomState(5)
    for name, (mode, ep_lens, stack_nstep) in {
        "final": ("final", [6, 4], False),
        "random": ("random", [7], False),
        "final_nstep": ("final", [6, 5], True),
    }.items():
        sink = Sink()
        inner = NStepReturn(sink, 1000, 0.98) if stack_nstep else sink
        random.seed(3)
        w = HindsightNStepReplay(inner, l2_sparse_reward, mode=mode)
        inputs = []
        for L in ep_lens:
            dg = rng.uniform(-1, 1, 2).astype(np.float32)
            pos = rng.uniform(-1, 1, 2).astype(np.float32)
            for i in range(L):
                pos = (pos + rng.uniform(-0.3, 0.3, 2)).astype(np.float32)
                if i in (2, 3):  # revisit the same place so hindsight sub-episodes appear
                    pos = np.asarray([0.5, 0.5], np.float32) + np.float32(0.01 * i)
                rew, td = l2_sparse_reward(pos, dg)
                row = {"obs_1d": rng.standard_normal(3).astype(np.float32),
                       "achieved_goal": pos.copy(), "desired_goal": dg.copy(),
                       "action": rng.uniform(-1, 1, 2).astype(np.float32),
                       "reward": float(rew) + 0.125 * i,  # goal-agnostic component
                       "task_done": bool(td), "episode_done": i == L - 1, "episode_step": i,
                       "info": {}}
                inputs.append(row)
                w.add(dict(row))
        keys = [k for k in inputs[0] if k != "info"]
        out[name] = {
            "mode": mode, "ep_lens": np.asarray(ep_lens), "thr": 0.25,
            "nstep": int(stack_nstep), "gamma": 0.98,
            "in": {k: np.stack([np.asarray(r[k]).reshape(-1) for r in inputs]) for k in keys},
            "out": sink.stacked(),
        }
    save("her", out)


class TeeMemory(ReplayMemory):
    """The reference ring that also records every record it is handed (what the write wrappers emitted)."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.rows = []

    def add(self, d):
        self.rows.append({k: np.array(v) for k, v in d.items()})
        super().add(d)


def golden_her_vmap():
    """her_vmap.py:10-123 + nstep_return_vmap.py:8-74 as the reference stacks them
    (franQ/Replay/__init__.py:20-36): HindsightVmapWrite -> NStepReturnVmap -> ReplayMemory on the write side,
    HindsightVmapRead -> ReplayMemory on the read side."""
    out = {}
    for name, (K, ep_lens, n_step, gamma, T, B) in {
        "k4": (4, [6, 5, 3], 1000, 0.97, 3, 4),
        "k32_pop": (32, [9, 4], 4, 0.9, 2, 6),      # n_step 4: NStepReturnVmap._pop fires once per episode (q3)
    }.items():
        rng = np.random.RandomState(21 + K)
        ring = TeeMemory(64, B, T)
        inner = NStepReturnVmap(ring, n_step, gamma)
        w = HindsightVmapWrite(inner, l2_sparse_reward, num_virtual_goals=K)
        inputs, goal_idx = [], []
        np.random.seed(100 + K)
        for L in ep_lens:
            dg = rng.uniform(-1, 1, 2).astype(np.float32)
            pos = rng.uniform(-1, 1, 2).astype(np.float32)
            for i in range(L):
                pos = (pos + rng.uniform(-0.3, 0.3, 2)).astype(np.float32)
                if i in (1, 3):     # the same place twice: a virtual goal drawn from it is reached mid-episode
                    pos = np.asarray([0.4, -0.2], np.float32) + np.float32(0.01 * i)
                rew, td = l2_sparse_reward(pos, dg)
                row = {"obs_1d": rng.standard_normal(3).astype(np.float32),
                       "achieved_goal": pos.copy(), "desired_goal": dg.copy(),
                       "action": rng.uniform(-1, 1, 2).astype(np.float32),
                       "reward": float(np.float32(float(rew) + 0.125 * i)),
                       "task_done": bool(td) or (i == L - 2 and L > 3),   # a real done besides the hindsight ones
                       "episode_done": i == L - 1, "episode_step": i, "info": {}}
                inputs.append(row)
                if row["episode_done"]:
                    # the draw her_vmap.py:75 is about to make (indices into the NEWEST-first episode buffer)
                    st = np.random.get_state()
                    goal_idx.append(np.random.randint(0, L, size=K))
                    np.random.set_state(st)
                w.add(dict(row))
        keys = [k for k in inputs[0] if k != "info"]
        emitted = {k: np.stack([np.asarray(r[k], np.float32).reshape(-1) for r in ring.rows]) for k in sorted(ring.rows[0])}
        # read side: one virtual column for the whole batch (q11); starts and column re-derived from the seeds
        r = HindsightVmapRead(ring)
        np.random.seed(7)
        random.seed(7)
        sample = r.temporal_sample()
        np.random.seed(7)
        random.seed(7)
        starts = np.random.randint(0, len(ring) - T, B)
        col = random.randint(0, K)
        out[name] = {
            "K": K, "ep_lens": np.asarray(ep_lens), "n_step": n_step, "gamma": gamma, "thr": 0.25, "T": T, "B": B,
            "in": {k: np.stack([np.asarray(x[k], np.float32).reshape(-1) for x in inputs]) for k in keys},
            "goal_idx_newest_first": np.stack(goal_idx),
            "out": emitted, "ring_len": len(ring),
            "read": {"starts": starts, "column": col,
                     "sample": {k: np.asarray(v, np.float32) for k, v in sample.items()}},
        }
    save("her_vmap", out)


# --------------------------------------------------------------------------------------
# Agent update (loss -> backward -> Adam -> polyak)
# --------------------------------------------------------------------------------------
class NoiseTap:
    """Records every draw the reference's distributions make, in call order."""

    def __init__(self):
        self.normal, self.uniform = [], []

    def __enter__(self):
        import torch.distributions.normal as tdn
        import torch.distributions.relaxed_categorical as trc
        self._tdn, self._trc = tdn, trc
        self._orig_sn = tdn._standard_normal
        tap = self

        def sn(shape, dtype, device):
            x = tap._orig_sn(shape, dtype, device)
            tap.normal.append(x.detach().clone())
            return x

        tdn._standard_normal = sn

        class _TorchProxy:
            def __getattr__(s, item):
                return getattr(torch, item)

            def rand(s, *a, **k):
                x = torch.rand(*a, **k)
                tap.uniform.append(x.detach().clone())
                return x

        self._orig_torch = trc.torch
        trc.torch = _TorchProxy()
        return self

    def __exit__(self, *a):
        self._tdn._standard_normal = self._orig_sn
        self._trc.torch = self._orig_torch


def make_conf(case):
    conf = AgentConf()
    spaces = {"obs_1d": Space(shape=(case["obs"],))}
    if case.get("goal", 0):
        spaces["achieved_goal"] = Space(shape=(case["goal"],))
        spaces["desired_goal"] = Space(shape=(case["goal"],))
    conf.obs_space = Space(spaces=spaces)
    conf.discrete = bool(case.get("discrete", False))
    conf.action_space = Space(n=case["act"]) if conf.discrete else Space(shape=(case["act"],))
    conf.training_device = torch.device("cpu")
    conf.inference_device = torch.device("cpu")
    conf.batch_size = case["B"]
    conf.temporal_len = case["T"]
    conf.num_critics = case["C"]
    conf.num_q_predictions = case["Q"]
    conf.latent_state_dim = case["latent"]
    conf.pi_hidden_dims = list(case["pi_hidden"])
    conf.critic_hidden_dims = list(case["critic_hidden"])
    conf.use_distributional_sac = bool(case.get("distributional", True))
    conf.use_nStep_lowerbounds = bool(case.get("lowerbound", True))
    conf.use_max_entropy_q = bool(case.get("max_entropy", True))
    conf.use_hard_updates = bool(case.get("hard_updates", False))
    conf.use_bootstrap_minibatch_nstep = bool(case.get("bootstrap", False))
    if case.get("gru"):      # encoder.py:40-42: nn.GRU joiner; `gru` names the latent-state training mode
        ec = conf.encoder_conf
        ec.joiner_mode = type(ec).JoinerModeEnum.gru
        ec.rnn_latent_state_training_mode = type(ec).RnnLatentStateTrainMode[case["gru"]]
    if case.get("burn_in_portion", 0):
        conf.encoder_conf.use_burn_in = True
        conf.encoder_conf.burn_in_portion = float(case["burn_in_portion"])
    conf.encoder_conf.hidden_features = case["enc_features"]
    conf.encoder_conf.obs_1d_hidden_dims = tuple(case["enc_hidden"])
    conf.encoder_conf.joint_hidden_dims = tuple(case["joint_hidden"])
    conf.log_dir = "/tmp/fdql_golden_logs"
    return conf


def make_batch(case, seed):
    g = np.random.RandomState(seed)
    T, B = case["T"], case["B"]
    xp = {"obs_1d": g.standard_normal((T, B, case["obs"]))}
    if case.get("goal", 0):
        xp["achieved_goal"] = g.standard_normal((T, B, case["goal"]))
        xp["desired_goal"] = g.standard_normal((T, B, case["goal"]))
    if case.get("discrete", False):
        xp["action"] = g.randint(0, case["act"], (T, B, 1)).astype(np.float64)
    else:
        xp["action"] = g.uniform(-1, 1, (T, B, case["act"]))
    xp["reward"] = g.standard_normal((T, B, 1))
    xp["mc_return"] = 2.0 * g.standard_normal((T, B, 1))
    xp["task_done"] = (g.rand(T, B, 1) < case.get("p_done", 0.12)).astype(np.float64)
    step = np.zeros((T, B, 1))
    edone = np.zeros((T, B, 1))
    for b in range(B):
        s = g.randint(0, 40)
        brk = g.randint(1, T) if g.rand() < case.get("p_break", 0.6) else -1
        for t in range(T):
            if t == brk:
                s = 0
                edone[t - 1, b, 0] = 1
            step[t, b, 0] = s
            s += 1
    if B > 2:  # one window with nothing contiguous -> exercises the +1e-4 normaliser
        step[:, 1, 0] = 5
    xp["episode_step"] = step
    xp["episode_done"] = edone
    xp["idx"] = np.zeros((T, B, 1))
    if case.get("gru") == "store":   # hidden state the actor had when it took the step (runner.py:157)
        xp["agent_state"] = g.uniform(0, 1, (T, B, case["latent"]))
    return {k: torch.tensor(v, dtype=torch.float32) for k, v in xp.items()}


class FakeLoader:
    def __init__(self):
        self.batch = None

    def temporal_sample(self):
        return {k: v.clone() for k, v in self.batch.items()}

    def ready(self):
        return True


def tap_module(mod, store, name):
    orig = mod.forward

    def fwd(*a, **k):
        y = orig(*a, **k)
        ys = y if isinstance(y, tuple) else (y,)
        store.setdefault(name, []).append([t.detach().clone() if isinstance(t, torch.Tensor) else t for t in ys])
        return y

    mod.forward = fwd


def golden_update(name, case, n_steps=3):
    torch.manual_seed(case.get("seed", 0))
    conf = make_conf(case)
    agent = DeepQLearning(conf)
    agent.optimizers = [torch.optim.Adam(agent.parameters(), lr=conf.learning_rate)]
    loader = FakeLoader()
    agent.replays = [loader]

    out = {"case": {k: (np.asarray(v) if not isinstance(v, (list, tuple)) else np.asarray(v)) for k, v in case.items()},
           "init": {k: v.clone() for k, v in agent.state_dict().items()},
           "hyper": {"gamma": conf.gamma, "tau": conf.tau, "lr": conf.learning_rate,
                     "init_log_alpha": conf.init_log_alpha, "top_quantiles_to_drop": conf.top_quantiles_to_drop}}
    inter = {}
    ac = agent.actor_critic
    tap_module(agent.encoder, inter, "encoder")
    tap_module(ac.actor_target, inter, "actor_target")
    tap_module(ac.actor, inter, "actor")
    tap_module(ac.critic_target, inter, "critic_target")
    tap_module(ac.critic, inter, "critic")
    tap_module(ac.critic_frozen, inter, "critic_frozen")

    # also tap the loss pieces (q_loss / actor_loss outputs)
    orig_q, orig_a = ac.q_loss, ac.actor_loss
    pieces = {}

    def q_loss(c, n):
        r = orig_q(c, n)
        pieces.setdefault("q_loss", []).append(r[0].detach().clone())
        pieces.setdefault("q_summaries", []).append({k: torch.as_tensor(v).detach().clone() for k, v in r[2].items()})
        return r

    def actor_loss(c):
        r = orig_a(c)
        pieces.setdefault("pi_loss", []).append(r[0].detach().clone())
        pieces.setdefault("alpha_loss", []).append(r[1].detach().clone())
        return r

    ac.q_loss, ac.actor_loss = q_loss, actor_loss

    losses = []
    orig_get_losses = agent.get_losses

    def get_losses(xp):
        l = orig_get_losses(xp)
        losses.append(l.detach().clone())
        pieces.setdefault("is_contiguous", []).append(xp["is_contiguous"].detach().clone().float())
        return l

    agent.get_losses = get_losses

    for step in range(n_steps):
        loader.batch = make_batch(case, seed=100 * (1 + case.get("seed", 0)) + step)
        alpha_in = float(torch.as_tensor(ac.curr_alpha))
        with NoiseTap() as tap:
            agent.train_step()
        rec = {"batch": loader.batch, "alpha_in": np.float32(alpha_in), "loss": losses[-1]}
        if conf.discrete:
            assert len(tap.uniform) == 2 and len(tap.normal) == 0
            rec["noise_target"], rec["noise_actor"] = tap.uniform
        else:
            assert len(tap.normal) == 2 and len(tap.uniform) == 0
            rec["noise_target"], rec["noise_actor"] = tap.normal
        rec["state"] = inter["encoder"][-1][0]
        rec["next_action"], rec["next_log_pi"] = inter["actor_target"][-1][0], inter["actor_target"][-1][1]
        rec["next_z"] = inter["critic_target"][-1][0]
        rec["q_pred"] = inter["critic"][-1][0]
        rec["pi"], rec["log_pi"] = inter["actor"][-1][0], inter["actor"][-1][1]
        rec["q_frozen"] = inter["critic_frozen"][-1][0]
        rec["q_loss"] = pieces["q_loss"][-1]
        rec["pi_loss"] = pieces["pi_loss"][-1]
        rec["alpha_loss"] = pieces["alpha_loss"][-1]
        rec["is_contiguous"] = pieces["is_contiguous"][-1]
        rec["summaries"] = pieces["q_summaries"][-1]
        names = {id(p): n for n, p in agent.named_parameters()}
        # keep the fixtures small: full tensors for the first and last step only
        if step in (0, n_steps - 1):
            # a parameter the loss does not reach has no .grad and torch's Adam skips it (encoder.hidden_state
            # outside the `learned` mode): recorded as zeros, which updates nothing either
            rec["grad"] = {names[id(p)]: (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p))
                           for p in agent.parameters()}
            rec["after"] = {k: v.clone() for k, v in agent.state_dict().items()}
        if step == n_steps - 1:
            opt = agent.optimizers[0]
            rec["adam_m"] = {names[id(p)]: (opt.state[p]["exp_avg"].clone() if p in opt.state else torch.zeros_like(p))
                             for p in agent.parameters()}
            rec["adam_v"] = {names[id(p)]: (opt.state[p]["exp_avg_sq"].clone() if p in opt.state else torch.zeros_like(p))
                             for p in agent.parameters()}
        out[f"step{step}"] = rec
    out["param_order"] = np.asarray([n for n, _ in agent.named_parameters()
                                     if any(p is q for q in agent.parameters() for p in [_])])
    out["n_trainable"] = sum(int(p.numel()) for p in agent.parameters())
    save(name, out)


def golden_act(name, case, rows=7):
    """DeepQLearning.act (deepQlearning.py:155-187) on a small inference batch: weights after
    `torch.manual_seed`, observations, exploit_mask, the policy's own noise draw (tapped) and the
    returned action / info tensors."""
    torch.manual_seed(case.get("seed", 0) + 1000)
    conf = make_conf(case)
    conf.use_async_train = True          # act() must not take a train step (deepQlearning.py:157-160)
    conf.log_extra_debug_info = False
    agent = DeepQLearning(conf)
    # one "training-like" perturbation so biases are not all zero (xavier init leaves them 0)
    with torch.no_grad():
        for p in agent.parameters():
            p.add_(0.05 * torch.randn_like(p))
    g = np.random.RandomState(77 + case.get("seed", 0))
    xp = {"obs_1d": torch.tensor(g.standard_normal((rows, case["obs"])), dtype=torch.float32)}
    if case.get("goal", 0):
        xp["achieved_goal"] = torch.tensor(g.standard_normal((rows, case["goal"])), dtype=torch.float32)
        xp["desired_goal"] = torch.tensor(g.standard_normal((rows, case["goal"])), dtype=torch.float32)
    if case.get("gru"):      # the hidden state the runner carries between steps (runner.py:103-106, 157)
        xp["agent_state"] = torch.tensor(g.uniform(0, 1, (rows, case["latent"])), dtype=torch.float32)
    xp["exploit_mask"] = torch.tensor((np.arange(rows) % 3 == 1).reshape(-1, 1))
    out = {"case": {k: np.asarray(v) for k, v in case.items()},
           "init": {k: v.clone() for k, v in agent.state_dict().items()
                    if k.startswith("encoder.") or k.startswith("actor_critic.actor.")},
           "xp": {k: v.clone() for k, v in xp.items()}}
    assert conf.train_step.value % conf.log_interval == 0
    with NoiseTap() as tap:
        action, hidden, info = agent.act(xp)
    assert (hidden is None) == (not case.get("gru"))
    if hidden is not None:
        out["hidden_state"] = hidden
    draws = tap.uniform if conf.discrete else tap.normal
    assert len(draws) == 1 and len(tap.uniform) + len(tap.normal) == 1
    out["noise"] = draws[0]
    out["action"] = action
    out["log_prob"], out["explore_action"], out["exploit_action"] = (info["log_prob"], info["explore_action"],
                                                                    info["exploit_action"])
    save(name, out)


ACT_CASES = ("tqc_c5q2", "tqc_goal", "tqc_discrete", "gru_store")


UPDATE_CASES = OrderedDict(
    tqc_small=dict(obs=5, act=3, C=3, Q=4, latent=32, enc_features=32, enc_hidden=(32,), joint_hidden=(32,),
                   pi_hidden=(32,), critic_hidden=(32, 32), T=6, B=8, seed=0),
    tqc_c5q2=dict(obs=17, act=6, C=5, Q=2, latent=32, enc_features=32, enc_hidden=(32,), joint_hidden=(32,),
                  pi_hidden=(32,), critic_hidden=(32, 32), T=4, B=16, seed=1),
    tqc_goal=dict(obs=4, goal=2, act=2, C=2, Q=5, latent=32, enc_features=24, enc_hidden=(40,), joint_hidden=(24,),
                  pi_hidden=(16,), critic_hidden=(48, 40), T=5, B=6, seed=2),
    sac_min=dict(obs=3, act=1, C=2, Q=1, latent=32, enc_features=32, enc_hidden=(32,), joint_hidden=(32,),
                 pi_hidden=(32,), critic_hidden=(32, 32), T=6, B=8, seed=3, distributional=False),
    tqc_discrete=dict(obs=6, act=4, discrete=True, C=3, Q=4, latent=32, enc_features=32, enc_hidden=(32,),
                      joint_hidden=(32,), pi_hidden=(32,), critic_hidden=(32, 32), T=5, B=8, seed=4),
    sac_boot=dict(obs=3, act=2, C=2, Q=2, latent=32, enc_features=32, enc_hidden=(32,), joint_hidden=(32,),
                  pi_hidden=(32,), critic_hidden=(32, 32), T=5, B=24, seed=6, distributional=False, bootstrap=True,
                  p_done=0.02, p_break=0.1),
    tqc_burn=dict(obs=5, act=3, C=3, Q=4, latent=32, enc_features=32, enc_hidden=(32,), joint_hidden=(32,),
                  pi_hidden=(32,), critic_hidden=(32, 32), T=6, B=8, seed=7, burn_in_portion=0.4, p_break=0.3),
    gru_zero=dict(obs=5, act=3, C=3, Q=4, latent=32, enc_features=24, enc_hidden=(32,), joint_hidden=(32,),
                  pi_hidden=(32,), critic_hidden=(32, 32), T=6, B=8, seed=8, gru="zero", p_break=0.3, p_done=0.05),
    gru_learned=dict(obs=5, act=3, C=3, Q=4, latent=32, enc_features=24, enc_hidden=(32,), joint_hidden=(32,),
                     pi_hidden=(32,), critic_hidden=(32, 32), T=5, B=8, seed=9, gru="learned", p_break=0.3, p_done=0.05),
    gru_store=dict(obs=5, act=3, C=3, Q=4, latent=32, enc_features=24, enc_hidden=(32,), joint_hidden=(32,),
                   pi_hidden=(32,), critic_hidden=(32, 32), T=5, B=8, seed=10, gru="store", p_break=0.3, p_done=0.05),
    tqc_nolb=dict(obs=5, act=3, C=3, Q=4, latent=32, enc_features=32, enc_hidden=(32,), joint_hidden=(32,),
                  pi_hidden=(32,), critic_hidden=(32, 32), T=3, B=4, seed=5, lowerbound=False, max_entropy=False),
)


def main():
    only = set(sys.argv[1:])   # e.g. `make_golden.py act` regenerates the act fixtures alone
    if not only or "ring" in only:
        golden_ring()
    if not only or "nstep" in only:
        golden_nstep()
    if not only or "her" in only:
        golden_her()
    if not only or "her_vmap" in only:
        golden_her_vmap()
    picked = {a.split(":", 1)[1] for a in only if a.startswith("update:")}   # e.g. `update:sac_boot`
    if not only or "update" in only or picked:
        for name, case in UPDATE_CASES.items():
            if not picked or name in picked:
                golden_update("update_" + name, case)
    if not only or "act" in only:
        for name in ACT_CASES:
            golden_act("act_" + name, UPDATE_CASES[name])


if __name__ == "__main__":
    main()
