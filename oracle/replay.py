"""CPU oracle (numpy) for the replay side of the hot path.  TEST INFRASTRUCTURE ONLY.

Restates, from the reference's text (citations are path:line under /root/reference):

  RingOracle            franQ/Replay/replay_memory.py:18-73
  cast_like_loader      franQ/Replay/wrappers/torch_dataloader.py:36
  nstep_episode / NStepOracle   franQ/Replay/wrappers/nstep_return.py:8-72
  her_relabel / HerOracle       franQ/Replay/wrappers/her.py:7-95
  vmap_* (shim-pinned)          franQ/Replay/wrappers/her_vmap.py:26-123,
                                franQ/Replay/wrappers/nstep_return_vmap.py:61-74
"""
import numpy as np


class OversampleError(Exception):
    """replay_memory.py:6"""


class RingOracle:
    """Dict-of-arrays ring.  replay_memory.py:18-73.

    Quirk q1 (kept): ``len`` is ``max(top, len)`` evaluated AFTER the modular increment, so
    after the first wrap it stays at ``maxlen - 1`` and slot ``maxlen-1`` is written but
    never sampled (replay_memory.py:45-46).
    """

    def __init__(self, maxlen, batch_size, temporal_len):
        self.maxlen, self.B, self.T = int(maxlen), int(batch_size), int(temporal_len)
        self.top, self.len = 0, 0
        self.memory = {}

    def add(self, row):
        if not self.memory:  # replay_memory.py:23-35
            for k, v in row.items():
                if isinstance(v, np.ndarray):
                    self.memory[k] = np.zeros((self.maxlen,) + v.shape, v.dtype)
                else:
                    self.memory[k] = np.zeros((self.maxlen, 1), np.float32)
        for k, v in row.items():  # replay_memory.py:42-43
            self.memory[k][self.top] = v
        self.top = (self.top + 1) % self.maxlen
        self.len = max(self.top, self.len)

    def __len__(self):
        return self.len

    def window_indices(self, starts):
        """idx[t, b] = (t + start[b]) % len.  replay_memory.py:62-65."""
        starts = np.asarray(starts)
        return (np.arange(self.T)[:, None] + starts[None, :]) % self.len

    def check_temporal(self):
        """replay_memory.py:57-58"""
        if self.len < 2 * self.T or self.len < self.B:
            raise OversampleError("Trying to sample more memories than available!")

    def draw_starts(self, rng=np.random):
        """replay_memory.py:59: randint(0, len - T, B)."""
        self.check_temporal()
        return rng.randint(0, self.len - self.T, self.B)

    def gather(self, idx):
        idx = np.asarray(idx)
        return {k: v[idx] for k, v in self.memory.items()}

    def temporal_sample(self, starts=None, rng=np.random):
        if starts is None:
            starts = self.draw_starts(rng)
        else:
            self.check_temporal()
        return self.gather(self.window_indices(starts))

    def sample(self, idx=None, rng=np.random):
        """replay_memory.py:48-52"""
        if self.len < self.B:
            raise OversampleError("Trying to sample more memories than available!")
        if idx is None:
            idx = rng.randint(0, self.len, self.B)
        return self.gather(idx)


def cast_like_loader(sample):
    """torch_dataloader.py:36 — every key becomes float32 (bools -> 0/1, ints -> float)."""
    return {k: np.asarray(v).astype(np.float32) for k, v in sample.items()}


# ---------------------------------------------------------------------------------------
# n-step / Monte-Carlo return at write time
# ---------------------------------------------------------------------------------------
def discounted_return_newest_first(rewards_newest_first, gamma):
    """nstep_return.py:60-72.  ``rewards[0]`` is the NEWEST record (deque.appendleft);
    in float32: ret[i] = r[i] + gamma * ret[i-1]  (reward-to-go including own reward).
    The multiply uses the Python double ``gamma`` against a float32 element and the sum is
    stored back to float32, exactly as the reference's in-place loop does."""
    r = np.asarray(rewards_newest_first, dtype=np.float32).reshape(-1).copy()
    for i in range(1, r.shape[0]):
        r[i] = np.float32(r[i] + r[i - 1] * gamma)
    return r


def discounted_return_numba_arithmetic(rewards_newest_first, gamma):
    """The same recurrence in the arithmetic REAL numba gives the reference's ``@numba.njit _inner`` (nstep_return.py:69-72):
    ``rewards`` is a float32 array and ``gamma`` a Python float, which numba types float64, so ``rewards[i - 1] * gamma`` and
    the sum are formed in double and rounded ONCE when stored back into the float32 array.  (The goldens were produced with
    njit = identity under numpy 2, where the Python float is weak: float32 product, float32 sum, float32(gamma) - the
    function above.  This one is the known-answer yardstick the n-step tests report their distance to.)"""
    r = np.asarray(rewards_newest_first, dtype=np.float32).reshape(-1).copy()
    g = float(gamma)
    for i in range(1, r.shape[0]):
        r[i] = np.float32(float(r[i]) + float(r[i - 1]) * g)
    return r


class NStepOracle:
    """NStepReturn wrapper, nstep_return.py:8-57, writing into ``sink.add(dict)``.

    Quirk q3 (kept): ``_pop`` never removes what it emits, so it fires exactly once per
    episode (when the buffer first reaches ``n_step``) and that record is emitted again by
    the end-of-episode flush."""

    def __init__(self, sink, n_step, gamma, reward_name="reward", return_name="mc_return",
                 done_name="episode_done"):
        self.sink, self.n_step, self.gamma = sink, n_step, gamma
        self.reward_name, self.return_name, self.done_name = reward_name, return_name, done_name
        self.buf = []  # oldest first (the reference keeps newest first; same content)

    def add(self, row):
        self.buf.append(row)
        if row[self.done_name]:
            self._flush()
        elif len(self.buf) == self.n_step:
            self._pop()

    def _returns_oldest_first(self):
        newest_first = [r[self.reward_name] for r in reversed(self.buf)]
        return discounted_return_newest_first(newest_first, self.gamma)[::-1]

    def _flush(self):  # nstep_return.py:36-48 — emits oldest first
        ret = self._returns_oldest_first()
        for row, g in zip(self.buf, ret):
            out = dict(row)
            out[self.return_name] = g
            self.sink.add(out)
        self.buf = []

    def _pop(self):  # nstep_return.py:50-57
        ret = self._returns_oldest_first()
        out = dict(self.buf[0])
        out[self.return_name] = ret[0]
        self.sink.add(out)


# ---------------------------------------------------------------------------------------
# Hindsight relabel at write time (modes "final" / "random")
# ---------------------------------------------------------------------------------------
def her_relabel(reward, episode_step, achieved_goal, desired_goal, goal, compute_reward):
    """her.py:55-95 for one finished episode given OLDEST-FIRST arrays.

    Returns (reward', task_done', episode_step') oldest first.  The reference walks the
    episode newest -> oldest (its deques are newest-first), opens a new synthetic
    sub-episode whenever the relabelled step is ``done`` (or at the newest step), and
    rebases ``episode_step`` by the step of the LAST element appended to each sub-episode,
    i.e. the OLDEST step of that sub-episode (her.py:72-83)."""
    n = len(reward)
    r_new = [None] * n
    d_new = [None] * n
    step_new = [None] * n
    groups = []  # lists of indices (oldest-first indexing), built newest -> oldest
    for pos, i in enumerate(range(n - 1, -1, -1)):  # newest first
        goal_reward, d = compute_reward(achieved_goal[i], goal)
        agnostic = reward[i] - compute_reward(achieved_goal[i], desired_goal[i])[0]
        r_new[i] = agnostic + goal_reward
        d_new[i] = d
        if d or pos == 0:
            groups.append([])
        groups[-1].append(i)
    for g in groups:
        base = episode_step[g[-1]]
        for i in g:
            step_new[i] = episode_step[i] - base
    return r_new, d_new, step_new


class HerOracle:
    """HindsightNStepReplay, her.py:7-95.  ``mode`` is "final" or "random"; for "random"
    the chosen index into the NEWEST-FIRST achieved_goal buffer comes from ``choose``
    (default: Python's ``random.choice`` like her.py:51-53)."""

    def __init__(self, sink, compute_reward, mode="random", ignore_keys=("info",), choose=None):
        self.sink, self.compute_reward, self.mode = sink, compute_reward, mode
        self.ignore, self.choose = ignore_keys, choose
        self.buf = []

    def add(self, row):
        self.buf.append(row)
        if row["episode_done"]:
            self._flush()
            self._hindsight()
            self.buf = []

    def _flush(self):  # her.py:36-46
        for row in self.buf:
            self.sink.add({k: v for k, v in row.items() if k not in self.ignore})

    def _goal(self):
        ags = [r["achieved_goal"] for r in reversed(self.buf)]  # newest first
        if self.mode == "final":
            return ags[0]
        if self.choose is not None:
            return ags[self.choose(len(ags))]
        import random
        return random.choice(ags)

    def _hindsight(self):
        goal = self._goal()
        r, d, s = her_relabel([x["reward"] for x in self.buf], [x["episode_step"] for x in self.buf],
                              [x["achieved_goal"] for x in self.buf], [x["desired_goal"] for x in self.buf],
                              goal, self.compute_reward)
        for i, row in enumerate(self.buf):
            out = {k: v for k, v in row.items() if k not in self.ignore}
            out["desired_goal"] = goal
            out["task_done"] = d[i]
            out["episode_step"] = s[i]
            out["reward"] = r[i]
            self.sink.add(out)


# ---------------------------------------------------------------------------------------
# HER-vmap ("sample-time" relabel).  SHIM-PINNED: the reference files need jax, which the image lacks; they were run
# on the numpy-backed jax stand-in of tests/golden/_refimport.py and this restatement is checked against the vectors
# they produced (tests/golden/her_vmap.npz).
# ---------------------------------------------------------------------------------------
def vmap_virtual_episode(virtual_goals, achieved_goal, desired_goal, reward, task_done, compute_reward):
    """her_vmap.py:30-45.  Inputs are per-episode arrays [n, ...]; virtual_goals [K, goal].
    Returns virtual_rewards [K, n], virtual_dones [K, n]."""
    n, K = len(reward), len(virtual_goals)
    vr = np.zeros((K, n), np.float32)
    vd = np.zeros((K, n), bool)
    for i in range(n):
        dr, dd = compute_reward(achieved_goal[i], desired_goal[i])
        agnostic_r = reward[i] - dr
        agnostic_d = bool(task_done[i]) and not bool(dd)
        for k in range(K):
            r, d = compute_reward(achieved_goal[i], virtual_goals[k])
            vr[k, i] = agnostic_r + r
            vd[k, i] = agnostic_d or bool(d)
    return vr, vd


def vmap_return_newest_first(rewards, dones, gamma):
    """nstep_return_vmap.py:61-74 (quirk q10: multiplies by ``dones[i]``, not 1-dones)."""
    r = np.asarray(rewards, np.float32).copy()
    d = np.asarray(dones, bool)
    for i in range(1, r.shape[0]):
        r[i] = np.float32(r[i] + r[i - 1] * gamma * d[i])
    return r


def vmap_read_select(sample, idx):
    """her_vmap.py:104-123: ONE virtual-goal column for the whole batch (quirk q11)."""
    out = dict(sample)
    out["desired_goal"] = sample["virtual_goals"][:, :, idx]
    out["reward"] = sample["virtual_rewards"][:, :, idx, None]
    out["task_done"] = sample["virtual_dones"][:, :, idx, None]
    if "virtual_mc_return" in sample:
        out["mc_return"] = sample["virtual_mc_return"][:, :, idx, None]
    for k in ("virtual_goals", "virtual_rewards", "virtual_dones", "virtual_mc_return"):
        out.pop(k, None)
    return out


class VmapWriteOracle:
    """HindsightVmapWrite stacked on NStepReturnVmap as franQ/Replay/__init__.py:20-36 builds them:
    her_vmap.py:56-90 buffers an episode, draws K goal indices into the NEWEST-first buffer (``draw(n, K)``; the
    reference calls ``np.random.randint(0, n, size=K)``, her_vmap.py:75), relabels, and hands the records oldest first
    to nstep_return_vmap.py:26-59, which emits its one-shot ``_pop`` duplicate (quirk q3) and, at the episode end, every
    record with the per-column return (quirk q10).  ``sink.add(dict)`` receives what the ring would."""

    def __init__(self, sink, compute_reward, num_virtual_goals, n_step, gamma, draw=None, ignore_keys=("info",)):
        self.sink, self.compute_reward, self.K = sink, compute_reward, int(num_virtual_goals)
        self.n_step, self.gamma, self.ignore = int(n_step), gamma, ignore_keys
        self.draw = draw if draw is not None else (lambda n, K: np.random.randint(0, n, size=K))
        self.episode, self.pending = [], []

    def add(self, row):
        self.episode.append(row)
        if row["episode_done"]:
            self._hindsight_flush()
            self.episode = []

    def _hindsight_flush(self):
        ep, n = self.episode, len(self.episode)
        f32 = lambda key: np.asarray([np.asarray(x[key], np.float32).reshape(-1) for x in ep], np.float32)
        ag, dg = f32("achieved_goal"), f32("desired_goal")
        reward, done = f32("reward")[:, 0], np.asarray([bool(x["task_done"]) for x in ep])
        newest_first = np.asarray(self.draw(n, self.K))
        goals = ag[::-1][newest_first]                                  # her_vmap.py:75
        vr, vd = vmap_virtual_episode(goals, ag, dg, reward, done, self.compute_reward)
        for i, row in enumerate(ep):                                    # her_vmap.py:80-90, oldest first
            out = {k: v for k, v in row.items() if k not in self.ignore}
            out["virtual_goals"] = np.concatenate([goals, dg[i][None]])
            out["virtual_rewards"] = np.concatenate([vr[:, i], [np.float32(reward[i])]]).astype(np.float32)
            out["virtual_dones"] = np.concatenate([vd[:, i], [done[i]]])
            self._nstep_add(out)

    def _returns(self):   # nstep_return_vmap.py:61-74 on the newest-first buffers
        r = np.asarray([x["virtual_rewards"] for x in reversed(self.pending)], np.float32)
        d = np.asarray([x["virtual_dones"] for x in reversed(self.pending)], bool)
        return vmap_return_newest_first(r, d, self.gamma)

    def _nstep_add(self, row):   # nstep_return_vmap.py:26-59
        self.pending.append(row)
        if row["episode_done"]:
            ret = self._returns()[::-1]
            for x, g in zip(self.pending, ret):
                out = dict(x)
                out["virtual_mc_return"] = g
                self.sink.add(out)
            self.pending = []
        elif len(self.pending) == self.n_step:
            out = dict(self.pending[0])
            out["virtual_mc_return"] = self._returns()[-1]
            self.sink.add(out)
