"""franQ.Replay.ReplayMemory / AsyncReplayMemory shapes over the HBM ring.

Reference: franQ/Replay/replay_memory.py:9-73 (ring), async_replay_memory.py:9-70 (proxy),
wrappers/torch_dataloader.py:11-50 (host->device, f32 cast).  Here the ring lives in HBM, so
one object plays all three roles: ``add`` stages rows host-side (pinned) and the gather kernel
returns float32 device tensors directly — no child process, no queues, no per-key H2D copy.
"""
import threading

import numpy as np
import torch

from .. import _native as N
from ..core import NativeRing

OversampleError = N.OversampleError


def _is_numeric(v):
    return isinstance(v, (np.ndarray, np.generic, int, float, bool)) or (isinstance(v, torch.Tensor))


class ReplayMemory:
    """Same constructor and methods as the reference; returns torch device tensors (float32),
    i.e. what ``TorchDataLoader(ReplayMemory(...))`` returns in the reference."""

    def __init__(self, maxlen, batch_size, temporal_len, device=None, seed=0, sample_buffers=0, **kwargs):
        """sample_buffers: 0 (default, like the reference: fresh tensors on every call).  n > 0: temporal_sample() writes
        into a rotating pool of n persistent output sets - a returned batch is OVERWRITTEN once n further samples have
        been drawn (the reference's loader hands out one prefetched batch at a time, torch_dataloader.py:22,47-50) and
        the agent sees a small, recurring set of batch addresses: its launch plans are cached per address set.
        Replay.make() turns it on (3) for the shards it builds for the trainer."""
        self._batch_size, self._temporal_len, self._maxlen = int(batch_size), int(temporal_len), int(maxlen)
        self._pool, self._pool_i, self._pool_n = [], 0, max(0, int(sample_buffers))
        self._init_lock = threading.Lock()     # first add() from two writer threads: one of them lays the ring out
        self.batch_size, self.temporal_len = self._batch_size, self._temporal_len
        self.device = torch.device(device if device is not None else "cuda:0")
        self._seed, self._counter = int(seed), 0
        self._ring = None
        self._in_place, self._in_place_cache, self._blocks, self._slot_pool, self._start_pool = (), None, {}, {}, {}
        self._keys, self._shapes, self._dims, self._dtypes = [], [], [], []

    # ---------------------------------------------------------------- write
    def _jit_initialize(self, experience_dict):
        """replay_memory.py:18-35: arrays keep their shape, scalars/bools become [1] float32.
        Non-numeric values (the runner's ``info`` dict) are not stored."""
        for k, v in experience_dict.items():
            if not _is_numeric(v):
                continue
            a = np.asarray(v.cpu() if isinstance(v, torch.Tensor) else v)
            shape = tuple(a.shape) if a.ndim > 0 else (1,)
            self._keys.append(k)
            self._shapes.append(shape)
            self._dims.append(int(np.prod(shape)) if shape else 1)
            # replay_memory.py:26-35 keeps the source dtype: uint8 frames stay one byte per element in HBM
            self._dtypes.append("u8" if a.dtype == np.uint8 and a.ndim > 0 else "f32")
        self._offsets = np.cumsum([0] + self._dims)
        self._new_ring()

    def _new_ring(self):
        """(Re)creates the HBM ring for the current key layout.  Everything that points INTO a ring - the uint8 blocks handed
        out for in-place reads, their slot / start buffers, the persistent sample buffers sized for the old layout - is dropped
        with the old one: a batch sampled after a restore must never carry views of the ring it replaced."""
        self._ring = NativeRing(self._maxlen, self._dims, self.device, dtypes=self._dtypes)
        self._in_place_cache, self._blocks, self._slot_pool, self._start_pool = None, {}, {}, {}
        self._pool, self._pool_i = [], 0

    def _pack(self, experience_dict, out):
        for j, k in enumerate(self._keys):
            v = experience_dict[k]
            if isinstance(v, torch.Tensor):
                v = v.detach().cpu().numpy()
            out[self._offsets[j]:self._offsets[j + 1]] = np.asarray(v, dtype=np.float32).reshape(-1)

    def _ensure_ring(self, template):
        if self._ring is None:
            with self._init_lock:
                if self._ring is None:
                    self._jit_initialize(template)

    def add(self, experience_dict):
        self._ensure_ring(experience_dict)
        row = np.empty(self._offsets[-1], np.float32)
        self._pack(experience_dict, row)
        self._ring.add_rows(row[None])

    def add_rows(self, rows):
        """Bulk append of packed float32 rows [n, sum(dims)] (host numpy or device tensor)."""
        self._ring.add_rows(rows)

    def append_episode(self, records, return_name=None, n_step=0, discount=0.0, emit_pop=False, her=None):
        """Write-path ingestion: a finished episode's records (list of dicts, oldest
        first) go to the ring in one call; the Monte-Carlo return (``return_name``; nstep_return.py:36-72)
        and the hindsight copy (``her = (goal_row, SparseL2Reward)``; her.py:55-95) are computed on the
        device, in the order the reference's wrapper stack would have added them.  Returns rows appended."""
        if not records:
            return 0
        if self._ring is None:
            template = dict(records[0])
            if return_name is not None:
                template[return_name] = 0.0     # NStepReturn adds the key last (nstep_return.py:44-46)
            self._ensure_ring(template)
        rows = np.zeros((len(records), int(self._offsets[-1])), np.float32)
        for j, k in enumerate(self._keys):
            if k == return_name:
                continue
            col = [r[k].detach().cpu().numpy() if isinstance(r[k], torch.Tensor) else r[k] for r in records]
            rows[:, self._offsets[j]:self._offsets[j + 1]] = np.asarray(col, dtype=np.float32).reshape(len(records), -1)
        key = lambda name: self._keys.index(name) if name in self._keys else -1
        sp = N.EpisodeSpec()
        sp.reward_key, sp.return_key = key("reward"), key(return_name) if return_name is not None else -1
        sp.emit_pop, sp.n_step, sp.gamma = int(bool(emit_pop)), int(n_step), float(discount)
        if her is not None:
            goal_row, fn = her
            sp.her = 1
            sp.achieved_key, sp.desired_key = key("achieved_goal"), key("desired_goal")
            sp.task_done_key, sp.step_key = key("task_done"), key("episode_step")
            sp.goal_row = int(goal_row)
            sp.reward_fn = fn.native()
        return self._ring.append_episode(rows, sp)

    # ---------------------------------------------------------------- read
    def _named(self, outs, lead):
        return {k: o.view(tuple(lead) + s) for k, s, o in zip(self._keys, self._shapes, outs)}

    def _check_init(self):
        if self._ring is None:
            raise OversampleError("Trying to sample more memories than available!")

    def sample(self, idxes=None):
        """replay_memory.py:48-52: [B, *shape] per key."""
        self._check_init()
        self._counter += 1
        outs = self._ring.sample_rows(self._batch_size, idx=idxes, seed=self._seed, counter=self._counter)
        return self._named(outs, (self._batch_size,))

    # ---------------------------------------------------------------- keys read in place (uint8 pixel frames)
    def enable_in_place(self, keys=("obs_2d",)):
        """Asks temporal_sample() NOT to gather these keys: for a uint8 key ``k`` the batch then carries the ring's own block
        (``xp[k]``: uint8 [maxlen, *shape], a zero-copy view) and ``xp[k + "_slots"]`` (int32 [T, B]: the slot of every row of
        the sampled windows, (start[b] + t) % len as in replay_memory.py:63-65).  The native agent's first conv layer reads the
        frames straight from there (fdql_batch_t.obs_2d_u8 / obs_2d_slots): no float32 frame batch is ever materialised.
        Keys that turn out not to be stored as uint8 keep being gathered.  Not part of the reference's interface: switched on
        by DeepQLearning.enable_training() for its own shards when the agent can use it."""
        self._in_place = tuple(keys)
        self._in_place_cache = None

    def _in_place_active(self):
        want = getattr(self, "_in_place", ())
        if not want or self._ring is None:
            return {}
        if getattr(self, "_in_place_cache", None) is None:
            self._in_place_cache = {k: self._keys.index(k) for k in want if k in self._keys and self._dtypes[self._keys.index(k)] == "u8"}
            self._blocks = {k: self._ring.key_block_u8(j).view((self._maxlen,) + tuple(self._shapes[j])) for k, j in self._in_place_cache.items()}
            self._slot_pool, self._start_pool = {}, {}
        return self._in_place_cache

    def external_read(self, begin):
        """Bracket around a consumer that reads the ring's blocks in place on the current stream (see enable_in_place)."""
        if self._ring is not None:
            self._ring.external_read(begin)

    def _temporal_sample_in_place(self, starts, active):
        T, B = self._temporal_len, self._batch_size
        slot = self._pool_i % self._pool_n if self._pool_n else None
        outs = None
        if self._pool_n:
            if len(self._pool) < self._pool_n:
                # (no float32 buffer for a key read in place: 2.9 GB per pool entry at config 5)
                self._pool.append([None if j in active.values() else torch.empty((T, B, d), dtype=torch.float32, device=self.device)
                                   for j, d in enumerate(self._dims)])
            outs = [None if j in active.values() else o for j, o in enumerate(self._pool[self._pool_i % len(self._pool)])]
        if slot is None or slot not in self._slot_pool:
            so = torch.empty(B, dtype=torch.int64, device=self.device)
            sl = torch.empty((T, B), dtype=torch.int32, device=self.device)
            if slot is not None:
                self._start_pool[slot], self._slot_pool[slot] = so, sl
        else:
            so, sl = self._start_pool[slot], self._slot_pool[slot]
        outs, so = self._ring.sample_windows(T, B, starts=starts, seed=self._seed, counter=self._counter, outs=outs,
                                             select={j: None for j in active.values()}, starts_out=so)
        self._ring.window_slots(T, B, so, out=sl)
        if self._pool_n:
            self._pool_i = (self._pool_i + 1) % self._pool_n
        res = {k: o.view((T, B) + s) for k, s, o in zip(self._keys, self._shapes, outs) if o is not None}
        for k in active:
            res[k], res[k + "_slots"] = self._blocks[k], sl
        return res

    def temporal_sample(self, starts=None):
        """replay_memory.py:54-65: [T, B, *shape] per key; raises OversampleError when
        len < 2T or len < B."""
        self._check_init()
        self._counter += 1
        active = self._in_place_active()
        if active:
            return self._temporal_sample_in_place(starts, active)
        outs = None
        if self._pool_n:
            if len(self._pool) < self._pool_n:
                self._pool.append([torch.empty((self._temporal_len, self._batch_size, d), dtype=torch.float32,
                                               device=self.device) for d in self._dims])
            outs = self._pool[self._pool_i % len(self._pool)]
        outs = self._ring.sample_windows(self._temporal_len, self._batch_size, starts=starts, seed=self._seed,
                                         counter=self._counter, outs=outs)
        if self._pool_n:
            self._pool_i = (self._pool_i + 1) % self._pool_n   # advanced only after a successful sample
        return self._named(outs, (self._temporal_len, self._batch_size))

    def temporal_sample_select(self, select, starts=None):
        """Windowed sample with per-key column selection (read side of HER-vmap).
        select: {key: None (drop) | (offset, dim)}; other keys are gathered whole."""
        self._check_init()
        self._counter += 1
        sel = {self._keys.index(k): v for k, v in select.items() if k in self._keys}
        outs = self._ring.sample_windows(self._temporal_len, self._batch_size, starts=starts, seed=self._seed,
                                         counter=self._counter, select=sel)
        res = {}
        for k, shp, o in zip(self._keys, self._shapes, outs):
            if o is None:
                continue
            res[k] = o if self._keys.index(k) in sel else o.view((self._temporal_len, self._batch_size) + shp)
        return res

    def __getitem__(self, idxes):
        """replay_memory.py:67-70: gather arbitrary index arrays (any shape)."""
        self._check_init()
        if isinstance(idxes, torch.Tensor):
            idxes = idxes.detach().cpu().numpy()
        idx = np.asarray(idxes)
        if idx.dtype == np.bool_ or not np.issubdtype(idx.dtype, np.integer):
            raise IndexError("ReplayMemory.__getitem__ takes integer slot indices")
        idx = idx.astype(np.int64)
        # numpy semantics on the reference's [maxlen, ...] arrays: negatives count from maxlen, anything else
        # outside [0, maxlen) is an IndexError; slots in [len, maxlen) hold their (possibly zero) contents
        bad = (idx < -self._maxlen) | (idx >= self._maxlen)
        if bad.any():
            raise IndexError(f"index {int(idx[bad].flat[0])} is out of bounds for axis 0 with size {self._maxlen}")
        idx = np.where(idx < 0, idx + self._maxlen, idx)
        outs = self._ring.gather_rows(torch.from_numpy(idx.reshape(-1)))
        return {k: o.view(tuple(idx.shape) + s) for k, s, o in zip(self._keys, self._shapes, outs)}

    def __len__(self):
        return len(self._ring) if self._ring is not None else 0

    def ready(self):
        n = len(self)
        return n >= 2 * self._temporal_len and n >= self._batch_size

    # ---------------------------------------------------------------- checkpoint (not in the reference: SURVEY 8f rank 3)
    def state_dict(self):
        """Everything a restart needs to continue sampling the same stream: key layout, ring contents
        (slot order), write position, length and the sample counter."""
        if self._ring is None:
            return {"keys": [], "shapes": [], "rows": np.zeros((0, 0), np.float32), "top": 0, "len": 0,
                    "counter": self._counter, "maxlen": self._maxlen}
        rows, top, n = self._ring.snapshot()
        return {"keys": list(self._keys), "shapes": [tuple(s) for s in self._shapes], "rows": rows, "top": top, "len": n,
                "counter": self._counter, "maxlen": self._maxlen, "dtypes": list(self._dtypes)}

    def load_state_dict(self, sd):
        if int(sd["maxlen"]) != self._maxlen:
            raise ValueError(f"ring checkpoint holds maxlen {sd['maxlen']}, this ring {self._maxlen}")
        self._counter = int(sd["counter"])
        if not sd["keys"]:
            return
        self._keys, self._shapes = list(sd["keys"]), [tuple(s) for s in sd["shapes"]]
        self._dims = [int(np.prod(s)) if s else 1 for s in self._shapes]
        self._offsets = np.cumsum([0] + self._dims)
        self._dtypes = list(sd.get("dtypes") or ["f32"] * len(self._keys))
        self._new_ring()
        self._ring.restore(sd["rows"], sd["top"], sd["len"])

    def save(self, path):
        sd = self.state_dict()
        np.savez(path, rows=sd["rows"], keys=np.asarray(sd["keys"]), top=sd["top"], len=sd["len"], counter=sd["counter"],
                 maxlen=sd["maxlen"], shapes=np.asarray([",".join(str(int(x)) for x in s) for s in sd["shapes"]]),
                 dtypes=np.asarray(sd.get("dtypes", [])), proxy_len=int(sd.get("proxy_len", -1)))

    def load(self, path):
        z = np.load(path if str(path).endswith(".npz") else str(path) + ".npz", allow_pickle=False)
        sd = {"rows": z["rows"], "keys": [str(k) for k in z["keys"]], "top": int(z["top"]),
              "len": int(z["len"]), "counter": int(z["counter"]), "maxlen": int(z["maxlen"]),
              "shapes": [tuple(int(x) for x in s.split(",") if x) for s in z["shapes"]],
              "dtypes": [str(t) for t in z["dtypes"]] if "dtypes" in z.files else None}
        if "proxy_len" in z.files and int(z["proxy_len"]) >= 0:
            sd["proxy_len"] = int(z["proxy_len"])
        self.load_state_dict(sd)


class AsyncReplayMemory(ReplayMemory):
    """Name kept for drop-in use.  The reference's proxy keeps its own saturating counter
    (async_replay_memory.py:27-29) which, unlike the ring's, reaches ``maxlen``; the HBM ring is
    "asynchronous" by construction (adds and samples are ordered on one HIP stream)."""

    def __init__(self, maxlen, batch_size, temporal_len, **kwargs):
        kwargs.pop("log_dir", None)
        super().__init__(maxlen, batch_size, temporal_len, **kwargs)
        self._len = 0

    def add(self, experience_dict):
        self._len = min(self._len + 1, self._maxlen)
        super().add(experience_dict)

    def append_episode(self, records, **kw):
        n = super().append_episode(records, **kw)
        self._len = min(self._len + n, self._maxlen)
        return n

    def state_dict(self):
        sd = super().state_dict()
        sd["proxy_len"] = self._len
        return sd

    def load_state_dict(self, sd):
        super().load_state_dict(sd)
        self._len = int(sd.get("proxy_len", min(int(sd["len"]) + 1, self._maxlen) if int(sd["len"]) == self._maxlen - 1
                               else int(sd["len"])))

    def __len__(self):
        return self._len

    def ready(self):
        n = ReplayMemory.__len__(self)
        return n >= 2 * self._temporal_len and n >= self._batch_size
