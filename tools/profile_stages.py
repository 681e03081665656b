#!/usr/bin/env python3
"""Per-stage timing of one update (HIP events inside the C library), config 2 by default."""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastdeepqlearning_amd.core import NativeAgent, make_config

ap = argparse.ArgumentParser()
ap.add_argument("--T", type=int, default=50); ap.add_argument("--B", type=int, default=256)
ap.add_argument("--obs", type=int, default=17); ap.add_argument("--act", type=int, default=6)
ap.add_argument("--C", type=int, default=5); ap.add_argument("--Q", type=int, default=2)
ap.add_argument("--hid", type=int, default=256); ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--gru", default="", help="GRU joiner with this latent-state mode: zero | learned")
ap.add_argument("--world", type=int, default=1, help="> 1: the data-parallel (two-bucket) plan of one rank")
a = ap.parse_args()
dev = torch.device("cuda:0")
cfg = make_config(a.obs, a.act, a.T, a.B, n_critics=a.C, n_quantiles=a.Q, latent=a.hid, enc_features=a.hid,
                  enc_hidden=(a.hid,), joint_hidden=(a.hid,), pi_hidden=(a.hid,), critic_hidden=(a.hid, a.hid),
                  joiner_gru=bool(a.gru), gru_state_mode=a.gru or 0, world_size=a.world)
ag = NativeAgent(cfg, dev); ag.init_weights(0)
T, B = a.T, a.B
xp = {"obs_1d": torch.randn(T, B, a.obs, device=dev), "action": torch.rand(T, B, a.act, device=dev) * 2 - 1,
      "reward": torch.randn(T, B, 1, device=dev), "mc_return": torch.randn(T, B, 1, device=dev),
      "task_done": (torch.rand(T, B, 1, device=dev) < 0.001).float(),
      "episode_step": (torch.arange(T, device=dev).view(T, 1, 1) + torch.randint(0, 900, (1, B, 1), device=dev)).float()}
for _ in range(3): ag.update(xp, seed=1)
acc = {}
order = []
for r in range(a.reps):
    for i, (name, ms, fl, by) in enumerate(ag.profile_update(xp, seed=1)):
        key = (i, name)
        if key not in acc: acc[key] = [0.0, fl, by]; order.append(key)
        acc[key][0] += ms
tot = 0
print(f"{'stage':34s} {'ms':>8s} {'GFLOP':>9s} {'TFLOP/s':>8s} {'MB':>8s} {'GB/s':>8s}")
for key in order:
    ms = acc[key][0] / a.reps; fl, by = acc[key][1], acc[key][2]; tot += ms
    print(f"{key[1]:34s} {ms:8.4f} {fl/1e9:9.3f} {fl/ms/1e9 if ms>0 else 0:8.2f} {by/1e6:8.2f} {by/ms/1e6 if ms>0 else 0:8.1f}")
print(f"total {tot:.4f} ms; stats {ag.stats()}")
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(50): ag.update(xp, seed=1)
torch.cuda.synchronize()
print(f"update-only wall: {(time.perf_counter()-t0)/50*1e3:.4f} ms/step")
