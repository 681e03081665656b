"""Write-path ingestion rate (records/s through the wrapper stack into the HBM ring):
fused fdql_ring_append_episode vs the per-record wrapper path.  Config 2 (NStepReturn, episodes of
1000) and config 3 (HindsightNStepReplay over NStepReturn, goal rows, episodes of 50)."""
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from fastdeepqlearning_amd.Replay import ReplayMemory
from fastdeepqlearning_amd.Replay.wrappers import HindsightNStepReplay, NStepReturn, SparseL2Reward
from fastdeepqlearning_amd.Replay.wrappers.wrapper_base_class import ReplayMemoryWrapper


def episodes(cfg, n_eps, L, rng):
    fn = SparseL2Reward(0.05, -1.0)
    eps = []
    for _ in range(n_eps):
        ep = []
        for i in range(L):
            r = {"obs_1d": rng.standard_normal(17 if cfg == 2 else 28).astype(np.float32),
                 "action": rng.uniform(-1, 1, 6).astype(np.float32), "reward": float(rng.standard_normal()),
                 "task_done": False, "episode_done": i == L - 1, "episode_step": i, "idx": 0}
            if cfg == 3:
                r["achieved_goal"] = rng.standard_normal(10).astype(np.float32)
                r["desired_goal"] = rng.standard_normal(10).astype(np.float32)
                r["info"] = {}
            ep.append(r)
        eps.append(ep)
    return eps, fn


def main():
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(0)
    for cfg, L, n_eps in ((2, 1000, 6), (3, 50, 60)):
        eps, fn = episodes(cfg, n_eps, L, rng)
        for fused in (True, False):
            mem = ReplayMemory(1_000_000, 256, 50, device=dev)
            base = mem if fused else ReplayMemoryWrapper(mem)
            stack = NStepReturn(base, 5000, 0.99)
            if cfg == 3:
                stack = HindsightNStepReplay(stack, fn, mode="final", device=dev)
            random.seed(0)
            for rec in eps[0]:           # warm up (ring allocation, staging buffers)
                stack.add(dict(rec))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for ep in eps[1:]:
                for rec in ep:
                    stack.add(dict(rec))
            mem._ring.flush()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            n = (n_eps - 1) * L
            print(f"config {cfg} episodes of {L}: {'fused append_episode' if fused else 'per-record wrappers '} "
                  f"{n / dt:10.0f} env records/s  ({dt / (n_eps - 1) * 1e3:.2f} ms per episode, ring len {len(mem)})", flush=True)


if __name__ == "__main__":
    main()
