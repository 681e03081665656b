"""Diagnostic: per-operation shader-clock stamps of the middle workgroup of the critic chain launch (config 2)."""
import ctypes as C, os, sys
os.environ.setdefault("FDQL_CHAIN", "all")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastdeepqlearning_amd.core import NativeAgent, make_config
from fastdeepqlearning_amd import _native as nat
dev = torch.device("cuda:0"); T, B, hid = 50, 256, 256
cfg = make_config(17, 6, T, B, n_critics=5, n_quantiles=2, latent=hid, enc_features=hid, enc_hidden=(hid,), joint_hidden=(hid,),
                  pi_hidden=(hid,), critic_hidden=(hid, hid))
ag = NativeAgent(cfg, dev); ag.init_weights(0)
nat.load().fdql_debug_chain_stamps(None, 1)   # recording on
xp = {"obs_1d": torch.randn(T, B, 17, device=dev), "action": torch.rand(T, B, 6, device=dev) * 2 - 1,
      "reward": torch.randn(T, B, 1, device=dev), "mc_return": torch.randn(T, B, 1, device=dev),
      "task_done": (torch.rand(T, B, 1, device=dev) < 0.001).float(),
      "episode_step": (torch.arange(T, device=dev).view(T, 1, 1) + torch.randint(0, 900, (1, B, 1), device=dev)).float()}
for _ in range(3): ag.update(xp, seed=1)
torch.cuda.synchronize()
names = [n for n, *_ in ag.profile_update(xp, seed=1)]
lib = nat.load()
buf = (C.c_uint64 * 256)()
# stamps of the LAST chain launch of the update = critics.fwd
ag.update(xp, seed=1); torch.cuda.synchronize()
n = lib.fdql_debug_chain_stamps(buf, 256)
st = [buf[i] for i in range(n)]
print("stamps:", n, "deltas (cycles):", [st[i + 1] - st[i] for i in range(n - 1)], "total", st[-1] - st[0])
