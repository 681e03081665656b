#!/usr/bin/env python3
"""Which workspace buffers / gradients hold non-finite values after one update on the inputs of
tests/test_gpu_parity.py::test_update_matches_oracle_config2[50-256-256] (debugging aid)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import update as oup
import test_gpu_parity as tg
dev = torch.device("cuda:0")
T, B, hid = 50, 256, 256
spec = oup.Spec(obs=17, act=6, C=5, Q=2, latent=hid, enc_features=hid, enc_hidden=(hid,), joint_hidden=(hid,),
                pi_hidden=(hid,), critic_hidden=(hid, hid), T=T, B=B)
params = oup.init_params(spec, seed=3)
ag = tg._agent_for(spec, dev)
ag.load_tensors(params)
g = torch.Generator().manual_seed(1)
xp = {"obs_1d": torch.randn(T, B, 17, generator=g), "action": torch.rand(T, B, 6, generator=g) * 2 - 1,
      "reward": torch.randn(T, B, 1, generator=g), "mc_return": torch.randn(T, B, 1, generator=g) * 2,
      "task_done": (torch.rand(T, B, 1, generator=g) < 0.05).float(),
      "episode_step": (torch.arange(T).view(T, 1, 1) + torch.randint(0, 900, (1, B, 1), generator=g)).float()}
xp["episode_step"][T // 2:, ::7] = torch.arange(T - T // 2).view(-1, 1, 1).float()
nt, na = torch.randn(T - 1, B, 6, generator=g), torch.randn(T - 1, B, 6, generator=g)
ag.update({k: v.to(dev) for k, v in xp.items()}, nt.to(dev), na.to(dev))
torch.cuda.synchronize()
names = ["state", "crit0.h0", "crit0.h1", "crit_f0.h0", "crit_f0.h1", "crit_t0.h0", "crit_t0.h1", "hf.parts", "q_pred", "q_frozen", "next_z",
         "td_target", "w", "dz", "dzf", "crit0.dpre1", "crit0.dpre0", "crit_f0.dpre0", "crit4.dpre1", "crit4.dpre0", "dpi_part", "dlogits",
         "actor.dpre0", "dstate.parts", "dstate", "joiner.dpre0", "enc_obs.dpre0", "slabs"]
for n in names:
    try:
        t = ag.debug(n).float().flatten()
    except Exception as e:
        print(n, "unavailable", str(e)[:60]); continue
    fin = torch.isfinite(t)
    bad = (~fin).sum().item()
    print(f"{n:16s} n={t.numel():9d} nonfinite={bad:8d} absmax={t[fin].abs().max().item() if bad < t.numel() else float('nan'):.4g}")
bad = {k: int((~torch.isfinite(v)).sum()) for k, v in ag.grad_views.items()}
print("grads with non-finite entries:", {k: v for k, v in bad.items() if v})
print("scalars", ag.scalars())
