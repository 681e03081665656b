"""HER with K virtual goals stored per step and ONE picked at read time ("vmap" mode).

Reference: franQ/Replay/wrappers/her_vmap.py:10-123.  SHIM-PINNED: the reference file needs jax, which the build
image lacks; it was run on a numpy-backed jax stand-in (tests/golden/_refimport.py) and this path is checked against the
vectors it produced (tests/golden/her_vmap.npz):
write: K goals ``achieved_goal[randint(0, n, K)]`` per finished episode; per step the K relabelled rewards /
dones plus the real ones as column K; read: one column for the WHOLE batch (quirk q11) replaces
desired_goal / reward / task_done / mc_return.  The relabel runs on the device (fdql_episode_her_vmap); the
column select is done inside the gather kernel (fdql_ring_sample_windows_sel)."""
import ctypes as C
import random

import numpy as np
import torch

from ... import _native as N
from .wrapper_base_class import ReplayMemoryWrapper
from .episode_ops import SparseL2Reward, _dev


class HindsightVmapWrite(ReplayMemoryWrapper):
    def __init__(self, replay_buffer, compute_reward, ignore_keys=("info",), num_virtual_goals=32, device=None, fused=True):
        super().__init__(replay_buffer)
        self.fused = fused    # False: always the per-record path (tests compare the two)
        # a SparseL2Reward is evaluated K x n times on the device; any other callable (the reference takes any
        # ``compute_reward(achieved_goal, goal) -> (reward, done)``, her_vmap.py:18-28) on the host like 'final'/'random' do
        self.compute_reward = compute_reward
        self._ignored_keys = ignore_keys
        self.num_virtual_goals = num_virtual_goals
        self._device = torch.device(device) if device is not None else getattr(replay_buffer, "device", torch.device("cuda:0"))
        self._reset()

    def _reset(self):
        self.buffer = []  # oldest first

    def add(self, experience):
        self.buffer.append(experience)
        if experience["episode_done"]:
            self._hindsight_flush()
            self._reset()

    def add_episode(self, episode):
        """A whole finished episode in ONE call (the reference only has the per-record ``add`` of
        her_vmap.py:66-90, which this is equivalent to when the records end with ``episode_done``).  ``episode`` is either the
        list of record dicts (oldest first) or ONE dict of stacked columns ``{key: array[n, ...]}`` - what a vectorised actor
        holds anyway - in which case no per-record Python runs at all: the columns are packed with one numpy slice assignment
        per key, copied to the device once, relabelled / scanned / appended there.  Returns the number of ring rows written
        (n, or n + 1 with the one-shot ``_pop`` record of the n-step wrapper)."""
        if self.buffer:
            raise RuntimeError("add_episode() with records of an unfinished episode pending from add()")
        if isinstance(episode, dict):
            n = int(np.asarray(episode["episode_done"]).reshape(-1).shape[0])
            done = np.asarray(episode["episode_done"]).reshape(-1)
            if n == 0 or not bool(done[-1]) or bool(done[:-1].any()):
                raise ValueError("add_episode(): exactly the last record of the episode must carry episode_done")
            target = self._fused_target()
            if target is not None:
                cols = {k: self._as_col(v, n) for k, v in episode.items() if k not in self._ignored_keys}
                first = {k: (episode[k][0] if np.ndim(episode[k]) > 0 else episode[k]) for k in cols}
                return self._flush_columns(target[0], target[1], cols, first, n)
            episode = [{k: (v[i] if np.ndim(v) > 0 else v) for k, v in episode.items()} for i in range(n)]   # host reward function
        else:
            episode = list(episode)
            if not episode or not episode[-1]["episode_done"] or any(r["episode_done"] for r in episode[:-1]):
                raise ValueError("add_episode(): exactly the last record of the episode must carry episode_done")
        self.buffer = episode
        try:
            return self._hindsight_flush()
        finally:
            self._reset()

    @staticmethod
    def _as_col(v, n):
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        return np.asarray(v, np.float32).reshape(n, -1)

    def _draw_goal_indices(self, n):
        """her_vmap.py:75 draws indices into the NEWEST-first buffer; converted to oldest-first."""
        newest_first = np.random.randint(0, n, size=self.num_virtual_goals)
        return (n - 1 - newest_first).astype(np.int32)

    def _fused_target(self):
        """(ring shard, NStepReturnVmap or None) when a finished episode can go to the ring as DEVICE rows in one call: a
        device reward function, and underneath nothing but (optionally) the vmap n-step wrapper with an empty buffer."""
        from ..replay_memory import ReplayMemory
        from .nstep_return_vmap import NStepReturnVmap
        if not self.fused or not isinstance(self.compute_reward, SparseL2Reward):
            return None
        child, nstep = self.replay_buffer, None
        if type(child) is NStepReturnVmap:
            nstep, child = child, child.replay_buffer
            if nstep.buffer:
                return None
        return (child, nstep) if isinstance(child, ReplayMemory) else None

    def _flush_fused(self, mem, nstep):
        """The relabel (fdql_episode_her_vmap), the per-column returns (fdql_episode_mc_return_vmap, incl. the one-shot _pop
        record of nstep_return_vmap.py:33-34, 50-57) and the packed rows stay on the device: one H2D copy of the episode's
        own columns, one append - no device -> host -> device round trip, no per-record add."""
        n = len(self.buffer)
        col = lambda k: np.asarray([np.asarray(x[k].detach().cpu().numpy() if isinstance(x[k], torch.Tensor) else x[k],
                                               np.float32).reshape(-1) for x in self.buffer], np.float32)
        # the record the per-record path would hand down (key order = insertion order), virtual columns last
        rec0 = {k: v for k, v in self.buffer[0].items() if k not in self._ignored_keys}
        return self._flush_columns(mem, nstep, {k: col(k) for k in rec0}, rec0, n)

    def _flush_columns(self, mem, nstep, cols, rec0, n):
        """``cols``: {key: float32 [n, dim]} of the episode's own keys; ``rec0``: its first record (the ring's template).
        ONE native call per episode (fdql_ring_append_episode_vmap): the packed rows and the K goal indices go to the device in
        one copy; relabel, per-column returns, the _pop record and the scatter into the ring run there."""
        K = self.num_virtual_goals
        g = cols["achieved_goal"].shape[1]
        if mem._ring is None:
            template = dict(rec0)
            template["virtual_goals"] = np.zeros((K + 1, g), np.float32)
            template["virtual_rewards"] = np.zeros(K + 1, np.float32)
            template["virtual_dones"] = np.zeros(K + 1, np.float32)
            if nstep is not None:
                template[nstep.return_name] = np.zeros(K + 1, np.float32)
            mem._ensure_ring(template)
        spec = getattr(self, "_vmap_spec", None)
        if spec is None or spec[0] is not mem or spec[1] != K:
            key = lambda name: mem._keys.index(name)
            sp = N.EpisodeVmapSpec()
            sp.reward_key, sp.task_done_key = key("reward"), key("task_done")
            sp.achieved_key, sp.desired_key = key("achieved_goal"), key("desired_goal")
            sp.vgoals_key, sp.vrewards_key, sp.vdones_key = key("virtual_goals"), key("virtual_rewards"), key("virtual_dones")
            sp.vreturn_key = key(nstep.return_name) if nstep is not None else -1
            sp.K, sp.n_step, sp.gamma = K, int(nstep.n_step) if nstep is not None else 0, float(nstep.discount) if nstep is not None else 0.0
            sp.reward_fn = self.compute_reward.native()
            offs = [(int(mem._offsets[j]), int(mem._offsets[j + 1])) for j in range(len(mem._keys))]
            spec = self._vmap_spec = (mem, K, sp, {k: offs[j] for j, k in enumerate(mem._keys)}, int(mem._offsets[-1]))
        _, _, sp, off, width = spec
        host = np.zeros((n, width), np.float32)
        for k, v in cols.items():
            if k in off:
                host[:, off[k][0]:off[k][1]] = v
        written = mem._ring.append_episode_vmap(host, self._draw_goal_indices(n), sp)
        if hasattr(mem, "_len"):          # AsyncReplayMemory's own saturating counter (async_replay_memory.py:27-29)
            mem._len = min(mem._len + written, mem._maxlen)
        return written

    def _hindsight_flush(self):
        target = self._fused_target()
        if target is not None:
            return self._flush_fused(*target)
        lib = N.load()
        n, K = len(self.buffer), self.num_virtual_goals
        dev = self._device
        col = lambda k: np.asarray([np.asarray(x[k], np.float32).reshape(-1) for x in self.buffer], np.float32)
        ag, dg = col("achieved_goal"), col("desired_goal")
        g = ag.shape[1]
        goal_idx = self._draw_goal_indices(n)
        if isinstance(self.compute_reward, SparseL2Reward):
            r, td = _dev(col("reward")[:, 0], dev), _dev(col("task_done")[:, 0], dev)
            idx = torch.as_tensor(goal_idx, dtype=torch.int32, device=dev)
            agd, dgd = _dev(ag, dev), _dev(dg, dev)
            vg = torch.empty(n, (K + 1) * g, device=dev)
            vr, vd = torch.empty(n, K + 1, device=dev), torch.empty(n, K + 1, device=dev)
            fn = self.compute_reward.native()
            with torch.cuda.device(dev):
                N.check(lib.fdql_episode_her_vmap(N.ptr(r), N.ptr(td), N.ptr(agd), N.ptr(dgd), C.c_void_p(idx.data_ptr()), n,
                                                  g, K, C.byref(fn), N.ptr(vg), N.ptr(vr), N.ptr(vd), N.current_stream(dev)))
            vg, vr, vd = vg.cpu().numpy().reshape(n, K + 1, g), vr.cpu().numpy(), vd.cpu().numpy()
        else:
            vg, vr, vd = self._host_relabel(ag, dg, col("reward")[:, 0], col("task_done")[:, 0] != 0, goal_idx)
        for i, row in enumerate(self.buffer):
            out = {k: v for k, v in row.items() if k not in self._ignored_keys}
            out["virtual_goals"] = vg[i]
            out["virtual_rewards"] = vr[i]
            out["virtual_dones"] = vd[i]
            self.replay_buffer.add(out)
        return n


    def _host_relabel(self, ag, dg, reward, done, goal_idx):
        """her_vmap.py:30-45 with a Python reward callable: float32 like jax (x64 off), (reward - R(ag, dg)) + R(ag, g_k),
        done_k = (done and not D(ag, dg)) or D(ag, g_k); column K keeps the real goal / reward / done."""
        n, K, g = len(reward), self.num_virtual_goals, ag.shape[1]
        vg = np.empty((n, K + 1, g), np.float32)
        vr, vd = np.empty((n, K + 1), np.float32), np.empty((n, K + 1), np.float32)
        goals = ag[goal_idx]
        for i in range(n):
            dr, dd = self.compute_reward(ag[i], dg[i])
            agnostic_r = np.float32(reward[i]) - np.float32(dr)
            agnostic_d = bool(done[i]) and not bool(dd)
            for k in range(K):
                r, d = self.compute_reward(ag[i], goals[k])
                vr[i, k] = agnostic_r + np.float32(r)
                vd[i, k] = float(agnostic_d or bool(d))
            vg[i, :K], vg[i, K] = goals, dg[i]
            vr[i, K], vd[i, K] = reward[i], float(done[i])
        return vg, vr, vd


class HindsightVmapRead(ReplayMemoryWrapper):
    """her_vmap.py:93-123: replaces goal / reward / done (/ mc_return) by ONE virtual column per batch."""

    VIRTUAL = {"virtual_goals": "desired_goal", "virtual_rewards": "reward", "virtual_dones": "task_done",
               "virtual_mc_return": "mc_return"}

    def temporal_sample(self, starts=None):
        """``starts``: optional window starts (parity runs); the virtual column comes from Python's ``random`` like
        the reference."""
        rb = self.replay_buffer
        keys, shapes = rb._keys, dict(zip(rb._keys, rb._shapes))
        k1 = shapes["virtual_rewards"][0]                      # K + 1 stored columns
        idx = random.randint(0, k1 - 1)                        # her_vmap.py:107-108 (inclusive bounds)
        g = int(np.prod(shapes["virtual_goals"])) // k1
        select = {"virtual_goals": (idx * g, g), "virtual_rewards": (idx, 1), "virtual_dones": (idx, 1),
                  "desired_goal": None, "reward": None, "task_done": None}
        if "virtual_mc_return" in keys:
            select["virtual_mc_return"] = (idx, 1)
            select["mc_return"] = None
        xp = rb.temporal_sample_select(select, starts=starts)
        for src, dst in self.VIRTUAL.items():
            if src in xp:
                xp[dst] = xp.pop(src)
        return xp

    def ready(self):
        r = getattr(self.replay_buffer, "ready", None)
        return bool(r()) if r is not None else True
