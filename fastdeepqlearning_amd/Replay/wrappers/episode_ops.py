"""Write-time episode transforms on the device (C ABI: fdql_episode_mc_return /
fdql_episode_her_relabel).  One finished episode at a time: upload the episode's columns,
run the kernel, read the few result columns back for the per-record emit the wrapper API
requires (franQ wrappers pass one dict per transition to the next wrapper)."""
import ctypes as C

import numpy as np
import torch

from ... import _native as N


class SparseL2Reward:
    """R(ag, g) = miss_reward if ||ag - g||_2 > threshold else 0; done = (R == 0).
    Callable like an env's ``compute_reward`` (franQ/Env/bitflip.py:143-152,
    classic_control_goal/classic_goal.py:88-93) AND describable to the device relabel kernel."""

    def __init__(self, threshold, miss_reward=-1.0):
        self.threshold, self.miss_reward = float(threshold), float(miss_reward)

    def __call__(self, achieved_goal, desired_goal):
        d = np.linalg.norm(np.asarray(achieved_goal, np.float32) - np.asarray(desired_goal, np.float32))
        r = np.float32(self.miss_reward) if d > self.threshold else np.float32(0.0)
        return r, bool(r == 0)

    def native(self):
        return N.RewardFn(0, self.threshold, self.miss_reward)


def _dev(a, device):
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).to(device)


def mc_return_device(rewards_oldest_first, gamma, device):
    """nstep_return.py:60-72 on the device; returns float32 numpy [n] (oldest first)."""
    lib = N.load()
    r = _dev(np.asarray(rewards_oldest_first, np.float32).reshape(-1), device)
    out = torch.empty_like(r)
    with torch.cuda.device(device):
        N.check(lib.fdql_episode_mc_return(N.ptr(r), N.ptr(out), r.numel(), float(gamma), N.current_stream(device)))
    return out.cpu().numpy()


def her_relabel_device(reward, episode_step, achieved_goal, desired_goal, goal, fn: SparseL2Reward, device):
    """her.py:55-95 on the device for one episode (oldest-first); returns (reward', done', step')."""
    lib = N.load()
    n = len(reward)
    ag = _dev(np.asarray(achieved_goal, np.float32).reshape(n, -1), device)
    dg = _dev(np.asarray(desired_goal, np.float32).reshape(n, -1), device)
    gl = _dev(np.asarray(goal, np.float32).reshape(-1), device)
    r, st = _dev(np.asarray(reward, np.float32).reshape(-1), device), _dev(np.asarray(episode_step, np.float32).reshape(-1), device)
    ro, do, so = (torch.empty(n, device=device) for _ in range(3))
    nf = fn.native()
    with torch.cuda.device(device):
        N.check(lib.fdql_episode_her_relabel(N.ptr(r), N.ptr(st), N.ptr(ag), N.ptr(dg), N.ptr(gl), n, ag.shape[1],
                                             C.byref(nf), N.ptr(ro), N.ptr(do), N.ptr(so), N.current_stream(device)))
    return ro.cpu().numpy(), do.cpu().numpy() != 0, so.cpu().numpy()
