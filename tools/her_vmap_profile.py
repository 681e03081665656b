#!/usr/bin/env python3
"""Where HindsightVmapWrite.add_episode(stacked columns) spends its time (cProfile, config 3 with K = 32 virtual goals)."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from fastdeepqlearning_amd.Replay import ReplayMemory
from fastdeepqlearning_amd.Replay.wrappers import HindsightVmapWrite, NStepReturnVmap, SparseL2Reward

dev = torch.device("cuda:0")
ring = ReplayMemory(1_000_000, 256, 50, device=dev)
w = HindsightVmapWrite(NStepReturnVmap(ring, 1000, 0.99), SparseL2Reward(0.05, -1.0), num_virtual_goals=32)
rng = np.random.RandomState(0)
n = 1000


def episode():
    return {"obs_1d": rng.standard_normal((n, 28)).astype(np.float32), "achieved_goal": rng.uniform(-1, 1, (n, 10)).astype(np.float32),
            "desired_goal": np.tile(rng.uniform(-1, 1, 10).astype(np.float32), (n, 1)), "action": rng.uniform(-1, 1, (n, 6)).astype(np.float32),
            "reward": np.full(n, -1.0), "task_done": np.zeros(n, bool), "episode_done": np.arange(n) == n - 1, "episode_step": np.arange(n)}


eps = [episode() for _ in range(20)]
for e in eps[:3]:
    w.add_episode(e)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(200):
    w.add_episode(eps[i % 20])
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{200 * n / dt:.0f} records/s ({dt / 200 * 1e3:.3f} ms per episode of {n})")
pr = cProfile.Profile()
pr.enable()
for i in range(100):
    w.add_episode(eps[i % 20])
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
