"""BASELINE config 5 end to end on one GPU: uint8 frame ring (4x84x84) -> windowed gather -> conv encoder ->
discrete SAC/TQC update.  No reference exists for the conv encoder, so this is a throughput figure only
(SURVEY 8d); parity of the kernels is against a torch conv2d oracle in the tests.
Usage: python tools/config5_bench.py [--T 50] [--B 512] [--ring 200000] [--steps 10]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from fastdeepqlearning_amd.core import NativeAgent, NativeRing, make_config

ap = argparse.ArgumentParser()
ap.add_argument("--T", type=int, default=50)
ap.add_argument("--B", type=int, default=512)
ap.add_argument("--ring", type=int, default=200_000)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--dense-shape", type=int, default=-1, help="tile shape of the dense GEMM launches (test hook fdql_debug_set_gemm_dense_shape: 0 = 128x128, 3 = 64x128, 5 = 64x64, the default)")
ap.add_argument("--f32", action="store_true", help="float32 frame batch (the im2col first layer) instead of reading the uint8 ring in place")
a = ap.parse_args()
dev = torch.device("cuda:0")
IMG, ACT = (4, 84, 84), 6
dims = [IMG[0] * IMG[1] * IMG[2], 1, 1, 1, 1, 1]
keys = ["obs_2d", "action", "reward", "mc_return", "task_done", "episode_step"]
ring = NativeRing(a.ring, dims, dev, dtypes=["u8", "f32", "f32", "f32", "f32", "f32"])
g = torch.Generator(device=dev).manual_seed(0)
done = 0
while done < a.ring:                      # synthetic fill, on the device
    n = min(4096, a.ring - done)
    rows = torch.empty(n, sum(dims), device=dev)
    rows[:, :dims[0]] = torch.randint(0, 256, (n, dims[0]), device=dev, generator=g).float()
    rows[:, dims[0]] = torch.randint(0, ACT, (n,), device=dev, generator=g).float()
    rows[:, dims[0] + 1] = torch.randn(n, device=dev, generator=g)
    rows[:, dims[0] + 2] = torch.randn(n, device=dev, generator=g)
    rows[:, dims[0] + 3] = (torch.rand(n, device=dev, generator=g) < 0.001).float()
    rows[:, dims[0] + 4] = ((torch.arange(n, device=dev) + done) % 1000).float()
    ring.add_rows(rows)
    done += n
cfg = make_config(0, ACT, a.T, a.B, discrete=True, n_critics=5, n_quantiles=2, img=IMG, conv=((32, 8, 4), (64, 4, 2), (64, 3, 1)),
                  obs_2d_u8=not a.f32)
if a.dense_shape >= 0:
    from fastdeepqlearning_amd import _native as nat
    nat.check(nat.load().fdql_debug_set_gemm_dense_shape(a.dense_shape))
agent = NativeAgent(cfg, dev)
agent.init_weights(0)
print(f"ring {a.ring} frames of {dims[0]} B as uint8; workspace {agent.workspace.numel() / 2**30:.1f} GiB; frames "
      f"{'gathered and widened to a float32 batch (im2col path for layer 0)' if a.f32 else 'read from the ring in place'}", flush=True)
if a.f32:
    outs = [torch.empty((a.T, a.B) + (IMG if k == "obs_2d" else (1,)), device=dev) for k in keys]
    xp = dict(zip(keys, outs))
    flat = [o.view(a.T, a.B, -1) for o in outs]

    def step(i):
        ring.sample_windows(a.T, a.B, seed=7, counter=i, outs=flat)
        agent.update(xp, seed=7)
else:
    outs = [None if k == "obs_2d" else torch.empty((a.T, a.B, d), device=dev) for k, d in zip(keys, dims)]
    starts = torch.empty(a.B, dtype=torch.int64, device=dev)
    slots = torch.empty((a.T, a.B), dtype=torch.int32, device=dev)
    xp = {k: o for k, o in zip(keys, outs) if o is not None}
    xp["obs_2d"], xp["obs_2d_slots"] = ring.key_block_u8(0), slots

    def step(i):
        ring.sample_windows(a.T, a.B, seed=7, counter=i, outs=outs, select={0: None}, starts_out=starts)
        ring.window_slots(a.T, a.B, starts, out=slots)
        agent.update(xp, seed=7)


for i in range(3):
    step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(a.steps):
    step(10 + i)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
st = agent.stats()
print(f"config 5 (T={a.T}, B={a.B}): {1 / dt:.2f} gradient-steps/s, {dt * 1e3:.2f} ms/step, {a.T * a.B / dt / 1e3:.0f} k frames/s, "
      f"{st['gemm_flops'] / 1e9:.0f} GFLOP/step through the GEMM kernel = {st['gemm_flops'] / dt / 1e12:.1f} TFLOP/s", flush=True)
top = agent.profile_update(xp, seed=7)   # every launch of the step, in launch order
for name, ms, fl, by in top:
    print(f"  {name:32s} {ms:8.3f} ms  {fl / ms / 1e9 if ms else 0:7.1f} TF  {by / ms / 1e6 if ms else 0:8.0f} GB/s", flush=True)
