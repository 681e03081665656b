"""DeepQLearning: the franQ trainer/actor object over the native update.

Reference: franQ/Agent/deepQlearning.py:23-280.  Same public surface (``enable_training``,
``train_step``, ``get_losses``, ``act``, ``state_dict``/``load_state_dict``, ``save`` /
``load_from_file``, ``iteration``, ``param_queue``, ``get_random_hidden``, ``reset``,
``update_targets``) so franQ.Runner / Evaluator can drive it unchanged.  ``train_step`` is ONE
call into libfdql_hip.so per replay shard; weights are zero-copy views of the device arenas
under the reference's state_dict names.
"""
import copy
import itertools
import logging
import threading
from collections import OrderedDict
from pathlib import Path
from queue import Queue

import numpy as np
import torch

from .. import _native as N
from .._native import OversampleError
from ..common_utils import AttrDict
from ..core import NativeAgent, make_config
from ..Replay.wrappers import TorchDataLoader


def _space_dim(space):
    return int(np.prod(space.shape))


def native_config_from_conf(conf):
    """Translate a franQ conf (conf.py:8-98, encoder.py:16-33) into the C-ABI config."""
    spaces = conf.obs_space.spaces
    ec = conf.encoder_conf
    img, conv, u8_frames = (), (), False
    if "obs_2d" in spaces:
        # the reference's pixel encoder is dead code (encoder.py:16-23); this build's own conv stack takes its place:
        # EncoderConf.conv_layers = ((out_channels, kernel, stride), ...), default the 3-layer Atari stack
        img = tuple(int(v) for v in spaces["obs_2d"].shape)
        conv = tuple(getattr(ec, "conv_layers", ((32, 8, 4), (64, 4, 2), (64, 3, 1))))
        # uint8 frames (the usual pixel space) stay uint8 from the ring to the first conv layer when that layer has an
        # implicit-GEMM kernel (DeepQLearning.__init__ falls back to float32 frames when it has not)
        u8_frames = np.dtype(getattr(spaces["obs_2d"], "dtype", np.float32)) == np.uint8 and not getattr(conf, "pixel_frames_float32", False)
    gru = getattr(getattr(ec, "joiner_mode", None), "name", "feedforward") == "gru"
    gru_mode = getattr(getattr(ec, "rnn_latent_state_training_mode", None), "name", "zero")
    obs = _space_dim(spaces["obs_1d"]) if "obs_1d" in spaces else 0
    goal = _space_dim(spaces["desired_goal"]) if "desired_goal" in spaces else 0
    if conf.discrete:
        act = int(conf.action_space.n)
    else:
        act = int(conf.action_space.shape[-1])
    boot = bool(getattr(conf, "use_bootstrap_minibatch_nstep", False))
    if boot and (conf.use_distributional_sac or not conf.use_nStep_lowerbounds):
        # the reference multiplies `None` in these combinations (deepQlearning.py:227; the term only exists in
        # SoftActorCritic.q_loss under use_nStep_lowerbounds, soft_actor_critic.py:92-102)
        raise ValueError("use_bootstrap_minibatch_nstep needs use_distributional_sac=False and use_nStep_lowerbounds=True")
    return make_config(obs, act, int(conf.temporal_len), int(conf.batch_size), goal_dim=goal, discrete=bool(conf.discrete),
                       n_critics=int(conf.num_critics), n_quantiles=int(conf.num_q_predictions),
                       latent=int(conf.latent_state_dim), enc_features=int(ec.hidden_features),
                       enc_hidden=tuple(ec.obs_1d_hidden_dims), joint_hidden=tuple(ec.joint_hidden_dims),
                       pi_hidden=tuple(conf.pi_hidden_dims), critic_hidden=tuple(conf.critic_hidden_dims),
                       distributional=bool(conf.use_distributional_sac), use_lowerbound=bool(conf.use_nStep_lowerbounds),
                       use_max_entropy=bool(conf.use_max_entropy_q), bootstrap_nstep=boot, joiner_gru=gru, gru_state_mode=gru_mode, img=img, conv=conv,
                       obs_2d_u8=u8_frames,
                       burn_in_steps=int(conf.temporal_len * ec.burn_in_portion) if getattr(ec, "use_burn_in", False) else 0, hard_updates=bool(conf.use_hard_updates),
                       keep_frozen_copy=True, world_size=int(getattr(conf, "world_size", 1) or 1),
                       gamma=float(conf.gamma), tau=float(conf.tau), lr=float(conf.learning_rate),
                       init_log_alpha=float(conf.init_log_alpha), drop_frac=float(conf.top_quantiles_to_drop))


STATE_DICT_ORDER = ("encoder.", "actor_critic.log_alpha", "actor_critic.critic.", "actor_critic.critic_target.",
                    "actor_critic.critic_frozen.", "actor_critic.actor.", "actor_critic.actor_target.")


class DeepQLearning:
    def __init__(self, conf, **kwargs):
        conf = conf if isinstance(conf, AttrDict) else AttrDict(conf)
        self.conf = conf
        self.param_queue = kwargs.get("param_queue", Queue(maxsize=1))
        self.device = torch.device(conf.training_device)
        cfg = native_config_from_conf(conf)
        try:
            self.native = NativeAgent(cfg, self.device)
        except Exception:
            if not cfg.obs_2d_u8:
                raise
            cfg.obs_2d_u8 = 0   # this conv stack's first layer has no uint8 kernel: float32 frames through the gather, as before
            self.native = NativeAgent(cfg, self.device)
        self.native.init_weights(seed=int(kwargs.get("seed", 0)))
        self.replays = []
        self._seed = int(kwargs.get("seed", 0))
        self._last_xp = None
        self._side, self._ev = None, None   # data-parallel: side stream + events of the bucketed all-reduce
        self.last_summaries = None
        self._trainer = None
        self._stop_training = False
        self.trainer_error = None
        # How a step's launch list is issued (fdql_agent_set_launch_mode): conf.launch_mode = "eager" (default) | "graph" | "auto".
        # "auto": plans of at most 6 272 gradient rows (temporal_len 2, one rank's share of a data-parallel batch: 17-23 dependent
        # launches of 5-40 us) are timed both ways on this host once 20 steps have run - 2 x 2 x 39 ORDINARY training steps inside
        # that train_step() call, which is why it is opt-in - and the faster way is kept; larger plans are kernel-bound and stay
        # eager.  The bucketed data-parallel step is always eager.  bench.py calibrates every launch-bound workload the same way.
        self.launch_mode_choice = None
        mode = self.conf.get("launch_mode", "eager")
        rows = (int(conf.temporal_len) - 1) * int(conf.batch_size)
        self._calibrate_at = 20 if (mode == "auto" and rows <= 6272 and not self._distributed()) else None
        if mode == "graph":
            self.native.set_launch_mode(True)
        if kwargs.get("train_process", False):   # reference signature: run the trainer loop in place
            self._initialize_trainer_members(kwargs["replays"])
            self._infinite_loop_for_async_training_process()

    # ------------------------------------------------------------------ training
    def enable_training(self, replays):
        self._initialize_trainer_members(replays)
        self._stop_training = False
        if self.conf.use_async_train:
            # the reference forks a trainer process; the GPU is asynchronous already, so a host
            # thread that enqueues kernels is enough and the weights stay shared (no state_dict pickling)
            self._trainer = threading.Thread(target=self._infinite_loop_for_async_training_process, daemon=True)
            self._trainer.start()

    def _initialize_trainer_members(self, replays):
        self.replays = [TorchDataLoader(r, self.device, torch.float32) for r in replays]
        if self.native.cfg.obs_2d_u8:
            for loader in self.replays:
                loader.keep_uint8.add("obs_2d")
                if self.native.conv_reads_ring():   # ring shards hand out their uint8 block + row slots instead of a gathered batch
                    shard = loader
                    while hasattr(shard, "replay_buffer"):
                        shard = shard.replay_buffer
                    if hasattr(shard, "enable_in_place"):
                        shard.enable_in_place(("obs_2d",))

    def _infinite_loop_for_async_training_process(self):
        """deepQlearning.py:83-94 (+ the crash report of :37-43: traceback, "[Trainer Crashed]" warning).  The failure
        is also kept in ``trainer_error`` and re-raised by the next ``act()``: acting on frozen weights for the
        rest of a run is the silent failure mode the reference's warning is there to prevent."""
        import time
        import traceback
        import warnings
        try:
            for step_train in itertools.count():
                if self._stop_training:
                    return
                if not all(r.ready() for r in self.replays):
                    time.sleep(0.05)
                    continue
                try:
                    self.train_step()
                except OversampleError:      # a shard shrank below a window between ready() and the sample
                    time.sleep(0.05)         # (async_replay_memory.py:57-61 sleeps and retries)
                    continue
                if (step_train % self.conf.param_update_interval) == 0 and not self.param_queue.full():
                    try:
                        self.param_queue.put_nowait(OrderedDict((k, v.to("cpu")) for k, v in self.state_dict().items()))
                    except Exception:
                        pass
        except Exception as e:   # noqa: BLE001 - report like the reference, then stop training
            traceback.print_exc()
            warnings.warn("[Trainer Crashed]")
            self.trainer_error = e

    def disable_training(self, timeout=10.0):
        """Stop the trainer thread started by enable_training (not in the reference, whose trainer is a daemon
        process killed with its parent)."""
        self._stop_training = True
        t, self._trainer = self._trainer, None
        if t is not None:
            t.join(timeout)

    def _distributed(self):
        """world_size > 1, or conf.force_distributed_step (tests: the bucketed step over a process group of ONE rank, so that the
        RCCL path runs on a one-GPU box; conf.world_size keeps its meaning - the loss is scaled by 1 / (B * world_size))."""
        return int(getattr(self.conf, "world_size", 1) or 1) > 1 or bool(getattr(self.conf, "force_distributed_step", False))

    def _all_reduce(self, g):
        """all-reduce(sum) of a contiguous slice of the gradient arena: RCCL over xGMI when the process group
        is nccl; with gloo (CPU rehearsals, tests) the slice is staged through the host.  The row weights of every rank's
        loss already carry 1 / (B_local * world_size), so the sum IS the global-batch gradient."""
        import torch.distributed as dist
        if g.numel() == 0:
            return
        if dist.get_backend() == "nccl":
            dist.all_reduce(g)
        else:
            h = g.cpu()
            dist.all_reduce(h)
            g.copy_(h)

    def _distributed_step(self, xp, nt, na):
        """Two buckets: the critics' gradients (+ log_alpha: arena[b:], 2/3 of it at config 2) are final after
        PHASE_GRAD_CRITICS and their all-reduce runs on a side stream beside the actor / encoder backward
        (PHASE_GRAD_REST); the rest follows; Adam waits for both."""
        nat = self.native
        dev = nat.device
        if self._side is None:
            self._side = torch.cuda.Stream(dev)
            self._ev = [torch.cuda.Event() for _ in range(3)]
        main, side, (e_a, e_b, e_red) = torch.cuda.current_stream(dev), self._side, self._ev
        b, g = nat.grad_bucket(), nat.grads
        nat.update(xp, nt, na, seed=self._seed, phase=N.PHASE_GRAD_CRITICS)
        e_a.record(main)
        with torch.cuda.stream(side):
            side.wait_event(e_a)
            self._all_reduce(g[b:])
        nat.update(None, phase=N.PHASE_GRAD_REST)
        e_b.record(main)
        with torch.cuda.stream(side):
            side.wait_event(e_b)
            self._all_reduce(g[:b])
            e_red.record(side)
        main.wait_event(e_red)
        nat.update(None, phase=N.PHASE_APPLY)

    def train_step(self, noise=None):
        """deepQlearning.py:105-127: for every shard: sample -> loss -> backward -> Adam -> polyak.
        noise: optional (noise_target, noise_actor) device tensors replayed by the policy kernels instead of the
        device's Philox draws (parity runs; the reference draws from torch's global generator)."""
        nt, na = noise if noise is not None else (None, None)
        if self._calibrate_at is not None and noise is None and self.iteration >= self._calibrate_at:
            self._calibrate_at = None       # (the calibration's own steps come back through here)
            self.launch_mode_choice = self.native.calibrate_launch_mode(self.train_step)
        for replay in self.replays:
            xp = replay.temporal_sample()
            self._last_xp = xp
            in_place = "obs_2d_slots" in xp      # the update reads the ring's frames in place: ordered against writers on other streams
            if in_place:
                replay.external_read(True)
            if self._distributed():
                self._distributed_step(xp, nt, na)
            else:
                self.native.update(xp, nt, na, seed=self._seed, phase=N.PHASE_ALL)
            if in_place:
                replay.external_read(False)
            self._log_summaries()
            self.conf.train_step.value += 1

    def _log_summaries(self):
        """The reference writes its trainer scalars to tensorboard every log_interval steps and the gradient norms every
        4 x log_interval (deepQlearning.py:114-122, 231-247).  Here: when the conf carries a callable ``summary_hook(step,
        scalars: dict)`` the same quantities are read back on those steps (one small kernel + a device -> host copy; the
        step itself never synchronises) and handed to it; ``last_summaries`` keeps the latest."""
        hook = self.conf.get("summary_hook") if hasattr(self.conf, "get") else getattr(self.conf, "summary_hook", None)
        step = int(self.conf.train_step.value)
        every = int(getattr(self.conf, "log_interval", 50) or 50)
        if not callable(hook) or step % every:
            return
        sc = self.native.scalars()
        sm = self.native.summaries(grad_norms=(step % (4 * every) == 0))
        out = {"Trainer/RL_Loss/Critic": sc["q_loss"], "Trainer/RL_Loss/Actor": sc["pi_loss"], "Trainer/RL_Loss/Alpha": sc["alpha_loss"],
               "Trainer/Critic_q_pred_mu": sc["q_pred_mu"], "Trainer/Critic_q_pred_var": sm["q_pred_var"],
               "Trainer/Critic_mc_constraint_violations": sc["mc_constraint_violations"], "Trainer/alpha": sc["alpha"],
               "Trainer/Valid_Portion/mean": sm["valid_portion_mean"], "Trainer/Valid_Portion/max": sm["valid_portion_max"],
               "Trainer/Valid_Portion/min": sm["valid_portion_min"]}
        for k, v in sm.get("grad_norms", {}).items():
            out[f"GradNorms/{k}"] = v
        self.last_summaries = out
        hook(step, out)

    def get_losses(self, xp, noise_target=None, noise_actor=None):
        """deepQlearning.py:198-249: loss of one [T,B,*] batch (also leaves d loss/d theta in the
        gradient arena and, like the reference's actor_loss, advances the lagged alpha)."""
        self.native.update(xp, noise_target, noise_actor, seed=self._seed, phase=N.PHASE_GRAD)
        ws = self.native.debug("scalars")
        return ws[0]

    def update_targets(self):
        """Polyak/hard target update is fused into the optimiser kernel (FDQL_PHASE_APPLY)."""

    def reset(self):
        pass

    def get_random_hidden(self):
        """encoder.py:113-117: None for the feed-forward joiner, torch.rand(latent) for the GRU."""
        if not self.native.cfg.joiner_gru:
            return None
        return torch.rand(int(self.native.cfg.latent))

    @property
    def iteration(self):
        return int(self.conf.train_step.value)

    def parameters(self, *args, **kwargs):
        return [self.native.tensors[k] for k in self.native.trainable]

    def to(self, device):
        assert torch.device(device) == self.device, "the native agent lives on conf.training_device"
        return self

    # ------------------------------------------------------------------ weights
    def state_dict(self):
        t = self.native.tensors
        out = OrderedDict()
        for prefix in STATE_DICT_ORDER:
            for k, v in t.items():
                if k.startswith(prefix):
                    out[k] = v
        return out

    def load_state_dict(self, sd):
        missing = [k for k in self.native.tensors if k not in sd]
        if missing:
            raise KeyError(f"missing keys in state_dict: {missing[:4]}...")
        self.native.load_tensors(sd)

    def save(self, logdir):
        """deepQlearning.py:260-267: conf.tch + state_dict.tch (plus optimiser state, which the
        reference does not save, in opt_state.tch)."""
        logdir = Path(logdir)
        logdir.mkdir(parents=True, exist_ok=True)
        conf = copy.copy(self.conf)
        conf.train_step = conf.train_step.value
        torch.save(conf, logdir / "conf.tch")
        torch.save(OrderedDict((k, v.detach().cpu().clone()) for k, v in self.state_dict().items()), logdir / "state_dict.tch")
        sc = self.native.scalars()
        torch.save({"adam_m": {k: v.cpu().clone() for k, v in self.native.m_views.items()},
                    "adam_v": {k: v.cpu().clone() for k, v in self.native.v_views.items()},
                    "step": int(sc["step"]),
                    # DevState.alpha_next: the lagged alpha the NEXT update will use (update_kernels.h)
                    "alpha": float(self.native.debug("dev_state")[2])}, logdir / "opt_state.tch")

    @staticmethod
    def load_from_file(logdir):
        from torch import multiprocessing as mp
        logdir = Path(logdir)
        conf = torch.load(logdir / "conf.tch", weights_only=False)
        conf.train_step = mp.Value("i", conf.train_step)
        agent = DeepQLearning(conf)
        agent.load_state_dict(torch.load(logdir / "state_dict.tch", weights_only=False))
        opt = logdir / "opt_state.tch"
        if opt.exists():
            o = torch.load(opt, weights_only=False)
            # alpha: the one-step-lagged exp(log_alpha) (soft_actor_critic.py:41,152), which the reference
            # would restart from exp(init_log_alpha)
            agent.native.load_opt_state(o["adam_m"], o["adam_v"], o["step"], o.get("alpha"))
        return agent

    # ------------------------------------------------------------------ inference
    def act(self, experiences, noise=None):
        """deepQlearning.py:155-187: small-batch inference for the env actors, one
        `fdql_agent_act` call (encoder -> joiner -> actor -> sample/select kernels) on the weight
        arena the trainer updates in place, so the actors always see the current weights without
        the reference's state_dict hop (deepQlearning.py:136-148).  Returns the reference's
        (action, hidden_state, info); hidden_state is None for the feed-forward joiner
        (encoder.py:63-65).  `noise` ([rows, A] N(0,1) / U(0,1) draws) replays a fixed sample
        (parity tests); by default the device draws Philox noise keyed by (seed, call count)."""
        conf = self.conf
        if self.trainer_error is not None:
            err, self.trainer_error = self.trainer_error, None
            raise RuntimeError("the trainer thread crashed; act() would run on frozen weights") from err
        if not conf.use_async_train and self.replays and all(r.ready() for r in self.replays):
            self.train_step()
        log_now = (conf.train_step.value % conf.log_interval) == 0
        self._act_calls = getattr(self, "_act_calls", 0) + 1
        res = self.native.act(
            experiences.get("obs_1d"), experiences.get("achieved_goal"), experiences.get("desired_goal"),
            experiences.get("exploit_mask"), noise=noise, seed=self._seed ^ 0xAC7, counter=self._act_calls,
            want_info=log_now, agent_state=experiences.get("agent_state"), obs_2d=experiences.get("obs_2d"))
        action, log_prob, explore, exploit = res[:4]
        hidden = res[4] if len(res) > 4 else None   # GRU joiner: the state the runner feeds back (runner.py:157)
        info = {}
        if conf.discrete:   # deepQlearning.py:175-178: argmax(-1, True) -> int64 indices
            action = action.long()
        if log_now:
            if conf.discrete:
                explore, exploit = explore.long(), exploit.long()
            info["log_prob"], info["explore_action"], info["exploit_action"] = log_prob, explore, exploit
        return action, hidden, info
