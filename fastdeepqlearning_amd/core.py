"""Thin object wrappers over the C ABI: device ring and device agent.

torch supplies device memory, streams and (for N > 1) torch.distributed; every byte of
compute on the hot path runs in libfdql_hip.so.  The franQ-shaped classes in
fastdeepqlearning_amd.Replay / .Agent are built on these two.
"""
import ctypes as C
from collections import OrderedDict

import numpy as np
import torch

from . import _native as N


def _device_view_u8(ptr, nbytes, device, owner):
    """A torch uint8 tensor over device memory owned by `owner` (kept alive through the tensor's attribute)."""
    class _Arr:      # __cuda_array_interface__: torch builds a non-owning view
        pass
    a = _Arr()
    a.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 3, "strides": None}
    with torch.cuda.device(device):
        t = torch.as_tensor(a, device=device)
    t._fdql_owner = owner
    return t


class _NoGuard:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def _on_device(device):
    """`with torch.cuda.device(device)` only when another device is current: the guard costs the per-step host path ~10 us,
    a launch-bound step (17-23 kernels of 5-40 us) has 100-120 us of host time per step to spend."""
    idx = device.index if device.index is not None else 0
    return _NO_GUARD if torch.cuda.current_device() == idx else torch.cuda.device(device)


class NativeRing:
    """SoA replay ring in HBM (include/fdql.h `fdql_ring_*`;
    reference franQ/Replay/replay_memory.py:18-73)."""

    def __init__(self, maxlen, dims, device="cuda:0", dtypes=None):
        """dtypes: optional per-key storage type, "f32" (default) or "u8" (pixel keys: one byte per element in
        HBM, widened to float32 by the gather, like the reference's uint8 ring + TorchDataLoader cast)."""
        self.lib = N.load()
        self.device = torch.device(device)
        self.dims = [int(d) for d in dims]
        self.dtypes = ["f32"] * len(self.dims) if dtypes is None else [str(t) for t in dtypes]
        self.maxlen = int(maxlen)
        self.handle = C.c_void_p()
        arr = (C.c_int32 * len(self.dims))(*self.dims)
        types = (C.c_int32 * len(self.dims))(*[{"f32": 0, "u8": 1}[t] for t in self.dtypes])
        with torch.cuda.device(self.device):
            N.check(self.lib.fdql_ring_create_typed(C.byref(self.handle), self.maxlen, len(self.dims), arr, types))
        self.row_floats = sum(self.dims)
        self._arr_cache = {}

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            try:
                self.lib.fdql_ring_destroy(h)
            except Exception:
                pass

    def __len__(self):
        return int(self.lib.fdql_ring_len(self.handle))

    @property
    def top(self):
        return int(self.lib.fdql_ring_top(self.handle))

    def add_rows(self, rows):
        """rows: float32 numpy [n, row_floats] (host) or torch device tensor of that shape."""
        if isinstance(rows, torch.Tensor) and rows.is_cuda:
            rows = rows.contiguous()
            assert rows.dtype == torch.float32 and rows.shape[-1] == self.row_floats
            N.check(self.lib.fdql_ring_add_device(self.handle, C.c_void_p(rows.data_ptr()), rows.shape[0],
                                                  N.current_stream(self.device)))
            return
        rows = np.ascontiguousarray(rows, dtype=np.float32).reshape(-1, self.row_floats)
        N.check(self.lib.fdql_ring_add(self.handle, rows.ctypes.data_as(C.c_void_p), rows.shape[0],
                                       N.current_stream(self.device)))

    def append_episode(self, rows, spec):
        """One finished episode (packed float32 rows [n, row_floats], oldest first, host) -> ring with the
        n-step scan / hindsight relabel on the device (fdql_ring_append_episode); returns rows appended."""
        rows = np.ascontiguousarray(rows, dtype=np.float32).reshape(-1, self.row_floats)
        out = C.c_int64(0)
        with torch.cuda.device(self.device):
            N.check(self.lib.fdql_ring_append_episode(self.handle, rows.ctypes.data_as(C.c_void_p), rows.shape[0],
                                                      C.byref(spec), C.byref(out), N.current_stream(self.device)))
        return int(out.value)

    def append_episode_vmap(self, rows, goal_idx, spec):
        """One finished episode of the "vmap" hindsight stack (packed float32 host rows [n, row_floats], K goal indices) ->
        ring: relabel, per-column returns, _pop record and scatter on the device (fdql_ring_append_episode_vmap)."""
        rows = np.ascontiguousarray(rows, dtype=np.float32).reshape(-1, self.row_floats)
        goal_idx = np.ascontiguousarray(goal_idx, dtype=np.int32)
        out = C.c_int64(0)
        N.check(self.lib.fdql_ring_append_episode_vmap(self.handle, rows.ctypes.data_as(C.c_void_p), rows.shape[0],
                                                       goal_idx.ctypes.data_as(C.c_void_p), C.byref(spec), C.byref(out),
                                                       N.current_stream(self.device)))
        return int(out.value)

    def flush(self):
        N.check(self.lib.fdql_ring_flush(self.handle, N.current_stream(self.device)))

    def snapshot(self):
        """(rows [n_slots, row_floats] float32 in slot order, top, len): everything a restart needs."""
        n_len, top = len(self), self.top
        n_slots = self.maxlen if n_len == self.maxlen - 1 else top   # every slot is written once the ring wrapped
        rows = np.empty((n_slots, self.row_floats), np.float32)
        with torch.cuda.device(self.device):
            N.check(self.lib.fdql_ring_snapshot(self.handle, rows.ctypes.data_as(C.c_void_p), n_slots,
                                                N.current_stream(self.device)))
        return rows, top, n_len

    def restore(self, rows, top, length):
        rows = np.ascontiguousarray(rows, dtype=np.float32).reshape(-1, self.row_floats)
        with torch.cuda.device(self.device):
            N.check(self.lib.fdql_ring_restore(self.handle, rows.ctypes.data_as(C.c_void_p), rows.shape[0], int(top),
                                               int(length), N.current_stream(self.device)))

    def _outs(self, lead):
        outs = [torch.empty(tuple(lead) + (d,), dtype=torch.float32, device=self.device) for d in self.dims]
        arr = (C.c_void_p * len(outs))(*[o.data_ptr() for o in outs])
        return outs, arr

    def sample_windows(self, T, B, starts=None, seed=0, counter=0, outs=None, return_starts=False, select=None, starts_out=None):
        """[T, B, dim_k] per key.  starts: optional int64 device tensor [B].
        select: optional {key_index: None (skip the key) | (offset, dim) (gather a sub-row)} — the read side
        of HER-vmap, and how a key that is read in place (window_slots) is left out; returns a list with None for skipped keys.
        outs: optional persistent output tensors (with select: one per key, None for skipped keys); starts_out: optional
        persistent int64 [B] tensor that receives the window starts (implies return_starts)."""
        if select is not None:
            return self._sample_windows_sel(T, B, starts, seed, counter, select, return_starts, outs, starts_out)
        if outs is None:
            outs, arr = self._outs((T, B))
        else:   # persistent output sets (a trainer's rotating pool): the pointer array of a set is built once
            key = tuple(o.data_ptr() for o in outs)
            arr = self._arr_cache.get(key)
            if arr is None:
                if len(self._arr_cache) >= 16:
                    self._arr_cache.clear()
                arr = self._arr_cache[key] = (C.c_void_p * len(outs))(*key)
        sp = None
        if starts is not None:
            starts = torch.as_tensor(starts, dtype=torch.int64, device=self.device).contiguous()
            sp = C.c_void_p(starts.data_ptr())
        so = starts_out if starts_out is not None else (torch.empty(B, dtype=torch.int64, device=self.device) if return_starts else None)
        with _on_device(self.device):
            N.check(self.lib.fdql_ring_sample_windows(self.handle, T, B, sp, seed, counter, arr,
                                                      C.c_void_p(so.data_ptr()) if so is not None else None,
                                                      N.current_stream(self.device)))
        return (outs, so) if (return_starts or starts_out is not None) else outs

    def _sample_windows_sel(self, T, B, starts, seed, counter, select, return_starts, given=None, starts_out=None):
        nk = len(self.dims)
        outs, ptrs = [], (C.c_void_p * nk)()
        off, dim = (C.c_int32 * nk)(), (C.c_int32 * nk)()
        for k in range(nk):
            sel = select.get(k, "all")
            if sel is None:
                outs.append(None)
                ptrs[k] = None
                continue
            if sel == "all":
                off[k], dim[k] = 0, 0
                d = self.dims[k]
            else:
                off[k], dim[k] = int(sel[0]), int(sel[1])
                d = int(sel[1])
            t = given[k] if given is not None and given[k] is not None else torch.empty((T, B, d), dtype=torch.float32, device=self.device)
            assert t.numel() == T * B * d and t.dtype == torch.float32 and t.is_contiguous()
            outs.append(t)
            ptrs[k] = t.data_ptr()
        sp = None
        if starts is not None:
            starts = torch.as_tensor(starts, dtype=torch.int64, device=self.device).contiguous()
            sp = C.c_void_p(starts.data_ptr())
        so = starts_out if starts_out is not None else (torch.empty(B, dtype=torch.int64, device=self.device) if return_starts else None)
        with torch.cuda.device(self.device):
            N.check(self.lib.fdql_ring_sample_windows_sel(self.handle, T, B, sp, seed, counter, ptrs, off, dim,
                                                          C.c_void_p(so.data_ptr()) if so is not None else None,
                                                          N.current_stream(self.device)))
        return (outs, so) if (return_starts or starts_out is not None) else outs

    def key_block_u8(self, key):
        """Zero-copy uint8 view [maxlen, dims[key]] of a uint8 key's block in the ring (fdql_ring_key_ptr_u8): what a consumer that
        reads the ring in place is handed (the pixel encoder's first layer, fdql_batch_t.obs_2d_u8 + obs_2d_slots)."""
        p = C.c_void_p()
        N.check(self.lib.fdql_ring_key_ptr_u8(self.handle, int(key), C.byref(p)))
        return _device_view_u8(p.value, self.maxlen * self.dims[key], self.device, self).view(self.maxlen, self.dims[key])

    def window_slots(self, T, B, starts, out=None):
        """int32 [T, B] slot of every row of the windows that begin at `starts` (int64 device tensor [B]): (start[b] + t) % len."""
        starts = torch.as_tensor(starts, dtype=torch.int64, device=self.device).contiguous()
        if out is None:
            out = torch.empty((T, B), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            N.check(self.lib.fdql_ring_window_slots(self.handle, T, B, C.c_void_p(starts.data_ptr()), C.c_void_p(out.data_ptr()),
                                                    N.current_stream(self.device)))
        return out

    def external_read(self, begin):
        """Bracket around a kernel that reads the ring's blocks in place on the current stream (fdql_ring_external_read)."""
        with torch.cuda.device(self.device):
            N.check(self.lib.fdql_ring_external_read(self.handle, 1 if begin else 0, N.current_stream(self.device)))

    def gather_rows(self, idx):
        """ring[k][idx] for explicit slot indices in [0, maxlen) (reference __getitem__, replay_memory.py:67-70)."""
        idx = torch.as_tensor(idx, dtype=torch.int64, device=self.device).contiguous().reshape(-1)
        outs, arr = self._outs((int(idx.numel()),))
        if idx.numel():
            with torch.cuda.device(self.device):
                N.check(self.lib.fdql_ring_gather_rows(self.handle, int(idx.numel()), C.c_void_p(idx.data_ptr()), arr,
                                                       N.current_stream(self.device)))
        return outs

    def sample_rows(self, B, idx=None, seed=0, counter=0):
        outs, arr = self._outs((B,))
        ip = None
        if idx is not None:
            idx = torch.as_tensor(idx, dtype=torch.int64, device=self.device).contiguous()
            ip = C.c_void_p(idx.data_ptr())
        with torch.cuda.device(self.device):
            N.check(self.lib.fdql_ring_sample_rows(self.handle, B, ip, seed, counter, arr, None,
                                                   N.current_stream(self.device)))
        return outs


def make_config(obs_dim, act_dim, T, B, goal_dim=0, discrete=False, n_critics=2, n_quantiles=10, latent=256,
                enc_features=256, enc_hidden=(256,), joint_hidden=(256,), pi_hidden=(256,), critic_hidden=(256, 256),
                distributional=True, use_lowerbound=True, use_max_entropy=True, hard_updates=False,
                keep_frozen_copy=True, world_size=1, gamma=0.99, tau=5e-2, lr=3e-4, beta1=0.9, beta2=0.999,
                adam_eps=1e-8, init_log_alpha=-2.0, drop_frac=0.2, bootstrap_nstep=False, burn_in_steps=0, joiner_gru=False,
                gru_state_mode=0, img=(), conv=(), obs_2d_u8=False):
    c = N.AgentConfig()
    c.obs_dim, c.goal_dim, c.act_dim, c.discrete = obs_dim, goal_dim, act_dim, int(discrete)
    c.n_critics, c.n_quantiles, c.latent, c.enc_features = n_critics, n_quantiles, latent, enc_features
    for name, val in (("enc_hidden", enc_hidden), ("joint_hidden", joint_hidden), ("pi_hidden", pi_hidden),
                      ("critic_hidden", critic_hidden)):
        val = tuple(int(v) for v in val)
        if len(val) > N.MAX_HIDDEN:
            raise ValueError(f"{name}: at most {N.MAX_HIDDEN} hidden layers")
        setattr(c, "n_" + name, len(val))
        arr = getattr(c, name)
        for i, v in enumerate(val):
            arr[i] = v
    c.distributional, c.use_lowerbound, c.use_max_entropy = int(distributional), int(use_lowerbound), int(use_max_entropy)
    c.hard_updates, c.keep_frozen_copy = int(hard_updates), int(keep_frozen_copy)
    c.bootstrap_nstep = int(bootstrap_nstep)
    c.burn_in_steps = int(burn_in_steps)
    c.obs_2d_u8 = int(bool(obs_2d_u8) and bool(img))
    if img:      # pixel encoder (a design of this build: the reference has none)
        c.img_c, c.img_h, c.img_w = (int(v) for v in img)
        if not 1 <= len(conv) <= N.MAX_CONV:
            raise ValueError(f"pixel input needs 1..{N.MAX_CONV} conv layers (out_channels, kernel, stride)")
        c.n_conv = len(conv)
        for i, (co, k, st) in enumerate(conv):
            c.conv_out[i], c.conv_k[i], c.conv_s[i] = int(co), int(k), int(st)
    c.joiner_gru = int(bool(joiner_gru))
    c.gru_state_mode = {"zero": 0, "store": 1, "learned": 2}.get(gru_state_mode, gru_state_mode) if joiner_gru else 0
    c.T, c.B, c.world_size = T, B, world_size
    c.gamma, c.tau, c.lr, c.beta1, c.beta2, c.adam_eps = gamma, tau, lr, beta1, beta2, adam_eps
    c.init_log_alpha, c.drop_frac = init_log_alpha, drop_frac
    return c


BATCH_KEYS = ("obs_1d", "achieved_goal", "desired_goal", "action", "reward", "mc_return", "task_done", "episode_step",
              "obs_2d", "agent_state")
_BATCH_CACHE_KEYS = BATCH_KEYS + ("obs_2d_slots",)


class NativeAgent:
    """One SAC/TQC learner on one GPU (include/fdql.h `fdql_agent_*`;
    reference franQ/Agent/deepQlearning.py:105-127,198-258).

    Parameters, gradients, Adam moments and targets live in flat fp32 torch tensors
    ("arenas"); ``tensors`` maps the reference's state_dict names to zero-copy views."""

    def __init__(self, cfg, device="cuda:0"):
        self.lib = N.load()
        self.cfg = cfg
        self.device = torch.device(device)
        self.handle = C.c_void_p()
        N.check(self.lib.fdql_agent_create(C.byref(self.handle), C.byref(cfg)))
        n0, n1, n2 = (int(self.lib.fdql_agent_arena_floats(self.handle, i)) for i in range(3))
        f32 = dict(dtype=torch.float32, device=self.device)
        self.params, self.grads = torch.zeros(n0, **f32), torch.zeros(n0, **f32)
        self.adam_m, self.adam_v = torch.zeros(n0, **f32), torch.zeros(n0, **f32)
        self.targets = torch.zeros(n1, **f32)
        self.frozen = torch.zeros(n2, **f32)
        self.workspace = torch.zeros(int(self.lib.fdql_agent_workspace_bytes(self.handle)) + 256, dtype=torch.uint8,
                                     device=self.device)
        self._ws_off = (-self.workspace.data_ptr()) % 256
        with torch.cuda.device(self.device):
            N.check(self.lib.fdql_agent_bind(self.handle, *[C.c_void_p(t.data_ptr()) for t in (
                self.params, self.grads, self.adam_m, self.adam_v, self.targets, self.frozen)],
                C.c_void_p(self.workspace.data_ptr() + self._ws_off), self.workspace.numel() - 256))
        self.tensors, self.grad_views, self.m_views, self.v_views = OrderedDict(), OrderedDict(), OrderedDict(), OrderedDict()
        arenas = {0: self.params, 1: self.targets, 2: self.frozen}
        n = self.lib.fdql_agent_tensor_info(self.handle, -1, None, 0, None, None, None)
        name = C.create_string_buffer(256)
        arena, off, shape = C.c_int32(), C.c_int64(), (C.c_int32 * 2)()
        for i in range(n):
            N.check(self.lib.fdql_agent_tensor_info(self.handle, i, name, 256, C.byref(arena), C.byref(off), shape))
            shp = (shape[0], shape[1]) if shape[1] else ((shape[0],) if shape[0] else ())
            cnt = int(np.prod(shp)) if shp else 1
            key = name.value.decode()
            self.tensors[key] = arenas[arena.value][off.value: off.value + cnt].view(shp)
            if arena.value == 0:
                self.grad_views[key] = self.grads[off.value: off.value + cnt].view(shp)
                self.m_views[key] = self.adam_m[off.value: off.value + cnt].view(shp)
                self.v_views[key] = self.adam_v[off.value: off.value + cnt].view(shp)
        self.trainable = list(self.grad_views.keys())
        self._keep = None
        self._batch_cache = {}

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            try:
                self.lib.fdql_agent_destroy(h)
            except Exception:
                pass

    # ------------------------------------------------------------------ update
    def _batch(self, xp):
        """fdql_batch_t from a dict of device tensors.  Pixel agents created with ``obs_2d_u8``: ``xp["obs_2d"]`` is a uint8
        tensor - the [T, B, c, h, w] frames, or (with ``xp["obs_2d_slots"]``, int32 [T, B]) the ring's own block of the key."""
        b = N.Batch()
        keep = []
        for k in BATCH_KEYS:
            t = xp.get(k)
            if t is None:
                continue
            if k == "obs_2d" and t.dtype == torch.uint8:
                assert t.is_cuda and t.is_contiguous(), "batch[obs_2d] (uint8) must be a contiguous device tensor"
                keep.append(t)
                b.obs_2d_u8 = t.data_ptr()
                sl = xp.get("obs_2d_slots")
                if sl is not None:
                    assert sl.is_cuda and sl.dtype == torch.int32 and sl.is_contiguous(), "batch[obs_2d_slots] must be an int32 device tensor"
                    keep.append(sl)
                    b.obs_2d_slots = sl.data_ptr()
                continue
            assert t.is_cuda and t.dtype == torch.float32, f"batch[{k}] must be a float32 device tensor"
            t = t.contiguous()
            keep.append(t)
            setattr(b, k, t.data_ptr())
        return b, keep

    def conv_reads_ring(self):
        """True when the first conv layer may read the ring's uint8 block in place (batch key ``obs_2d_slots``)."""
        return bool(self.lib.fdql_agent_conv_reads_ring(self.handle))

    def update(self, xp, noise_target=None, noise_actor=None, seed=0, phase=N.PHASE_ALL):
        """One train_step (deepQlearning.py:105-127) on the current torch stream; asynchronous."""
        if xp is None:
            b, keep = N.Batch(), []
        else:
            # the fdql_batch_t of a recurring batch (a rotating pool of sample buffers) is built once: keyed by the device
            # pointers themselves, so a dead tensor's recycled id can never alias a cached entry
            vals = [xp.get(k) for k in _BATCH_CACHE_KEYS]
            key = tuple(0 if t is None else (t.data_ptr(), t.dtype) for t in vals)
            b, keep = self._batch_cache.get(key), vals
            if b is None:
                b, keep = self._batch(xp)      # (checks device / dtype; copies a non-contiguous tensor - such a batch is not cached)
                if all(t is None or t.is_contiguous() for t in vals):
                    if len(self._batch_cache) >= 16:
                        self._batch_cache.clear()
                    self._batch_cache[key], keep = b, vals
        for t in (noise_target, noise_actor):
            if t is not None:
                keep.append(t)
        self._keep = keep  # keep the tensors alive until the next call
        with _on_device(self.device):
            N.check(self.lib.fdql_agent_update(self.handle, C.byref(b), N.ptr(noise_target), N.ptr(noise_actor), seed,
                                               phase, N.current_stream(self.device)))

    def set_launch_mode(self, graph):
        """PHASE_ALL as one hipGraphLaunch per step (True) or one launch per stage (False); fdql_agent_set_launch_mode."""
        N.check(self.lib.fdql_agent_set_launch_mode(self.handle, 1 if graph else 0))
        self.launch_mode = "graph" if graph else "eager"

    def calibrate_launch_mode(self, step, steps=30, warmup=9, rounds=2):
        """Times `step()` (a callable that issues one whole train step, sampler included, on the current stream) with the
        launch list issued eagerly and replayed as a hipGraph, `rounds` x `steps` steps each in alternation, and keeps the
        faster.  The answer depends on the host as much as on the plan (a launch-bound step of 17-23 dependent kernels is
        host-issue-bound on a slow or busy host and device-latency-bound on a fast one), so it is measured where the job runs,
        like the data-parallel launch list (bench.py::dp_run_best).  Returns {"chosen", "ms_per_step": {"eager", "graph"}}."""
        import time
        ms = {"eager": [], "graph": []}
        for _ in range(rounds):
            for mode in ("eager", "graph"):
                self.set_launch_mode(mode == "graph")
                for _ in range(warmup):          # (a plan's graph is captured on its second run in graph mode)
                    step()
                torch.cuda.synchronize(self.device)
                t0 = time.perf_counter()
                for _ in range(steps):
                    step()
                torch.cuda.synchronize(self.device)
                ms[mode].append(1e3 * (time.perf_counter() - t0) / steps)
        best = {m: min(v) for m, v in ms.items()}
        chosen = "graph" if best["graph"] < 0.98 * best["eager"] else "eager"      # a tie stays eager
        self.set_launch_mode(chosen == "graph")
        return {"chosen": chosen, "steps": steps, "rounds": rounds, "ms_per_step": {m: round(v, 4) for m, v in best.items()}}

    def grad_bucket(self):
        """First float of the gradient arena's EARLY bucket: grads[b:] (critics + log_alpha) is final after
        PHASE_GRAD_CRITICS, grads[:b] after PHASE_GRAD_REST; b == grads.numel() when the agent is not data-parallel."""
        out = C.c_int64(0)
        N.check(self.lib.fdql_agent_grad_bucket(self.handle, C.byref(out)))
        return int(out.value)

    def profile_update(self, xp, noise_target=None, noise_actor=None, seed=0):
        b, keep = self._batch(xp)
        arr = (N.KernelTime * 256)()
        with torch.cuda.device(self.device):
            n = self.lib.fdql_agent_profile_update(self.handle, C.byref(b), N.ptr(noise_target), N.ptr(noise_actor), seed,
                                                   arr, 256, N.current_stream(self.device))
        if n < 0:
            N.check(n)
        return [(arr[i].name.decode(), float(arr[i].ms), float(arr[i].flops), float(arr[i].bytes)) for i in range(n)]

    def act(self, obs_1d, achieved_goal=None, desired_goal=None, exploit_mask=None, noise=None, seed=0, counter=0,
            want_info=True, agent_state=None, obs_2d=None):
        """deepQlearning.py:155-187 on `rows` observations: encoder -> actor on the online weights in the
        shared arena.  Returns (action, log_prob, explore_action, exploit_action); the last three are None
        unless want_info.  Asynchronous on the current torch stream."""
        dev, cfg = self.device, self.cfg
        f32 = lambda x: torch.as_tensor(x, dtype=torch.float32, device=dev).contiguous()
        obs = img = None
        if cfg.obs_dim:
            obs = f32(obs_1d)
            if obs.dim() != 2 or obs.shape[1] != cfg.obs_dim:
                raise ValueError(f"act(): obs_1d must be [rows, {cfg.obs_dim}], got {tuple(obs.shape)}")
        if cfg.img_c:
            img = f32(obs_2d)
            if img.dim() != 4 or tuple(img.shape[1:]) != (cfg.img_c, cfg.img_h, cfg.img_w):
                raise ValueError(f"act(): obs_2d must be [rows, {cfg.img_c}, {cfg.img_h}, {cfg.img_w}]")
        rows = (obs if obs is not None else img).shape[0]
        ag = dg = None
        if cfg.goal_dim:
            ag, dg = f32(achieved_goal), f32(desired_goal)
            if tuple(ag.shape) != (rows, cfg.goal_dim) or tuple(dg.shape) != (rows, cfg.goal_dim):
                raise ValueError("act(): goal tensors must be [rows, goal_dim]")
        mask = None
        if exploit_mask is not None:
            mask = torch.as_tensor(exploit_mask, device=dev).reshape(-1).to(torch.uint8).contiguous()
            if mask.numel() != rows:
                raise ValueError("act(): exploit_mask must have one entry per row")
        if noise is not None:
            noise = f32(noise)
            if tuple(noise.shape) != (rows, cfg.act_dim):
                raise ValueError(f"act(): noise must be [rows, {cfg.act_dim}]")
        need = self.lib.fdql_agent_act_workspace_bytes(self.handle, rows)
        if getattr(self, "_act_ws", None) is None or self._act_ws.numel() < need:
            self._act_ws = torch.empty(max(need, 256), dtype=torch.uint8, device=dev)
        adim = 1 if cfg.discrete else cfg.act_dim
        action = torch.empty(rows, adim, device=dev)
        logp = torch.empty(rows, 1, device=dev) if want_info else None
        explore = torch.empty(rows, adim, device=dev) if want_info else None
        exploit = torch.empty(rows, adim, device=dev) if want_info else None
        hs_in = hs_out = None
        if cfg.joiner_gru:
            hs_out = torch.empty(rows, cfg.latent, device=dev)
            if agent_state is not None:
                hs_in = f32(agent_state)
                if tuple(hs_in.shape) != (rows, cfg.latent):
                    raise ValueError(f"act(): agent_state must be [rows, {cfg.latent}]")
        self._act_keep = (obs, ag, dg, mask, noise, hs_in, img)
        with torch.cuda.device(dev):
            N.check(self.lib.fdql_agent_act(self.handle, N.ptr(obs), N.ptr(ag), N.ptr(dg), N.ptr(img), N.ptr(hs_in),
                                            C.c_void_p(mask.data_ptr()) if mask is not None else None, N.ptr(noise),
                                            int(seed), int(counter), rows, N.ptr(action), N.ptr(logp), N.ptr(explore),
                                            N.ptr(exploit), N.ptr(hs_out), C.c_void_p(self._act_ws.data_ptr()),
                                            self._act_ws.numel(),
                                            N.current_stream(dev)))
        if cfg.joiner_gru:
            return action, logp, explore, exploit, hs_out
        return action, logp, explore, exploit

    def scalars(self):
        out = (C.c_float * 8)()
        with torch.cuda.device(self.device):
            N.check(self.lib.fdql_agent_scalars(self.handle, out, N.current_stream(self.device)))
        keys = ("loss", "q_loss", "pi_loss", "alpha_loss", "q_pred_mu", "mc_constraint_violations", "alpha", "step")
        return dict(zip(keys, [float(x) for x in out]))

    def summaries(self, grad_norms=False):
        """What franQ's trainer logs besides the losses (deepQlearning.py:231-247): q_pred_var, Valid_Portion mean / max / min
        and - grad_norms - the L2 norm of every trainable tensor's gradient; computed on demand, outside the step."""
        cap = 4 + len(self.trainable)
        out = (C.c_float * cap)()
        with torch.cuda.device(self.device):
            n = self.lib.fdql_agent_summaries(self.handle, out, cap, int(bool(grad_norms)), N.current_stream(self.device))
        if n < 0:
            N.check(n)
        d = dict(zip(("q_pred_var", "valid_portion_mean", "valid_portion_max", "valid_portion_min"), [float(x) for x in out[:4]]))
        if grad_norms:
            d["grad_norms"] = dict(zip(self.trainable, [float(x) for x in out[4:n]]))
        return d

    def stats(self):
        s = N.AgentStats()
        N.check(self.lib.fdql_agent_stats(self.handle, C.byref(s)))
        return {"gemm_flops": s.gemm_flops, "skinny_flops": s.skinny_flops, "n_launches": s.n_launches,
                "n_gemm_launches": s.n_gemm_launches, "params": s.params, "plans_built": s.plans_built,
                "graph_launches": s.graph_launches}

    def set_alpha(self, alpha):
        N.check(self.lib.fdql_agent_set_alpha(self.handle, float(alpha), N.current_stream(self.device)))

    def set_step(self, step):
        N.check(self.lib.fdql_agent_set_step(self.handle, int(step), N.current_stream(self.device)))

    def debug(self, name, shape=None):
        """Copy of a named intermediate of the last update (parity tests)."""
        p, cnt = C.c_void_p(), C.c_int64()
        N.check(self.lib.fdql_agent_debug_ptr(self.handle, name.encode(), C.byref(p), C.byref(cnt)))
        off = p.value - self.workspace.data_ptr()
        t = self.workspace[off: off + 4 * cnt.value].view(torch.float32).clone()
        return t.view(shape) if shape is not None else t

    # ------------------------------------------------------------------ weights
    def load_tensors(self, sd):
        """Copy tensors (reference state_dict names) into the arenas."""
        for k, v in sd.items():
            if k in self.tensors:
                self.tensors[k].copy_(torch.as_tensor(v, dtype=torch.float32).to(self.device))

    def load_opt_state(self, adam_m=None, adam_v=None, step=None, alpha=None):
        """Load Adam moments / step / lagged alpha (resume or test synchronisation)."""
        for views, src in ((self.m_views, adam_m), (self.v_views, adam_v)):
            if src is not None:
                for k, v in src.items():
                    views[k].copy_(torch.as_tensor(v, dtype=torch.float32).to(self.device))
        if step is not None:
            self.set_step(step)
        if alpha is not None:
            self.set_alpha(alpha)

    def init_weights(self, seed=0):
        """xavier_uniform(gain 1) weights, zero biases, targets = copies
        (franQ/Agent/models/mlp.py:5-8,86; soft_actor_critic.py:34,38)."""
        g = torch.Generator(device="cpu").manual_seed(seed)
        for k, v in self.tensors.items():
            if k.startswith("encoder.joiner.") and self.cfg.joiner_gru:   # nn.GRU.reset_parameters
                a = 1.0 / float(self.cfg.latent) ** 0.5
                v.copy_(((torch.rand(v.shape, generator=g) * 2 - 1) * a).to(self.device))
            elif k == "encoder.hidden_state":                              # encoder.py:42,117
                v.copy_(torch.rand(v.shape, generator=g).to(self.device))
            elif k.endswith("weight"):
                fan_out, fan_in = v.shape
                a = (6.0 / (fan_in + fan_out)) ** 0.5
                v.copy_(((torch.rand(v.shape, generator=g) * 2 - 1) * a).to(self.device))
            elif k.endswith("bias"):
                v.zero_()
            elif k.endswith("log_alpha"):
                v.fill_(float(self.cfg.init_log_alpha))
        for k, v in self.tensors.items():
            for src, dst in ((".actor.", ".actor_target."), (".critic.", ".critic_target.")):
                if dst in k:
                    v.copy_(self.tensors[k.replace(dst, src)])
