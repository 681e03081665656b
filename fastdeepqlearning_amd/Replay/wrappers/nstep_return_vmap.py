"""NStepReturnVmap: per-virtual-goal Monte-Carlo return attached at write time.

Reference: franQ/Replay/wrappers/nstep_return_vmap.py:8-74 (control flow as NStepReturn, incl. the one-shot
``_pop``); the recurrence multiplies by ``dones[i]`` (quirk q10).  SHIM-PINNED together with her_vmap.py (it only ever
runs behind the jax-based HindsightVmapWrite; vectors in tests/golden/her_vmap.npz).  The K+1 column scans run on the
device (fdql_episode_mc_return_vmap)."""
import numpy as np
import torch

from ... import _native as N
from .wrapper_base_class import ReplayMemoryWrapper
from .episode_ops import _dev


class NStepReturnVmap(ReplayMemoryWrapper):
    def __init__(self, replay_buffer, n_step, discount, reward_name="virtual_rewards", task_done_name="virtual_dones",
                 return_name="virtual_mc_return", done_name="episode_done", device=None):
        ReplayMemoryWrapper.__init__(self, replay_buffer)
        self.n_step, self.discount = n_step, discount
        self.reward_name, self.task_done_name = reward_name, task_done_name
        self.return_name, self.done_name = return_name, done_name
        self._device = torch.device(device) if device is not None else getattr(replay_buffer, "device", torch.device("cuda:0"))
        self.buffer = []

    def add(self, experience):
        self.buffer.append(experience)
        if experience[self.done_name]:
            self._flush()
        elif len(self.buffer) == self.n_step:
            self._pop()

    def _returns(self):
        lib = N.load()
        dev = self._device
        r = _dev(np.asarray([np.asarray(x[self.reward_name], np.float32).reshape(-1) for x in self.buffer]), dev)
        d = _dev(np.asarray([np.asarray(x[self.task_done_name], np.float32).reshape(-1) for x in self.buffer]), dev)
        out = torch.empty_like(r)
        with torch.cuda.device(dev):
            N.check(lib.fdql_episode_mc_return_vmap(N.ptr(r), N.ptr(d), N.ptr(out), r.shape[0], r.shape[1],
                                                    float(self.discount), N.current_stream(dev)))
        return out.cpu().numpy()

    def _flush(self):
        ret = self._returns()
        for row, g in zip(self.buffer, ret):
            out = dict(row)
            out[self.return_name] = g
            self.replay_buffer.add(out)
        self.buffer = []

    def _pop(self):
        ret = self._returns()
        out = dict(self.buffer[0])
        out[self.return_name] = ret[0]
        self.replay_buffer.add(out)
