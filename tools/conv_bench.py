#!/usr/bin/env python3
"""Event-timed launches of the implicit-GEMM convolution kernels (csrc/conv.hip) at BASELINE config 5's sizes
(T=50, B=512: 25 600 frame stacks forward, 25 088 backward): ms, TFLOP/s of useful work and the share of the fp32 MFMA peak."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from fastdeepqlearning_amd import _native as nat

dev = torch.device("cuda:0")
lib = nat.load()
LAYERS = [("conv0 32x8/4 on 4x84x84 u8", dict(C=4, H=84, W=84, k=8, s=4, co=32, u8=1)),
          ("conv1 64x4/2 on 20x20x32", dict(C=32, H=20, W=20, k=4, s=2, co=64, u8=0)),
          ("conv2 64x3/1 on 9x9x64", dict(C=64, H=9, W=9, k=3, s=1, co=64, u8=0))]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 25600
REPS = 5
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None


def timed(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REPS


for name, L in LAYERS:
    K, OH = L["C"] * L["k"] ** 2, (L["H"] - L["k"]) // L["s"] + 1
    pos = OH * OH
    x = (torch.randint(0, 256, (N, L["C"] * L["H"] * L["W"]), device=dev, dtype=torch.uint8) if L["u8"]
         else torch.randn(N, L["H"] * L["W"] * L["C"], device=dev))
    W = torch.randn(L["co"], K, device=dev) * 0.05
    bias = torch.zeros(L["co"], device=dev)
    out = torch.empty(N, pos, L["co"], device=dev)
    dpre = torch.randn(N, pos, L["co"], device=dev)
    dprev = torch.empty(N, L["H"] * L["W"] * L["C"], device=dev) if not L["u8"] else None
    dw = torch.empty(L["co"] * K + L["co"], device=dev)
    scratch = torch.empty(4096 * (L["co"] * K + L["co"]), device=dev)
    st = nat.current_stream(dev)
    flops = 2.0 * N * pos * K * L["co"]

    def call(mode):
        nat.check(lib.fdql_test_conv(mode, p(x), L["u8"], None, p(W), p(bias), p(dpre), p(x) if not L["u8"] else None,
                                     p(out if mode == 0 else (dprev if mode == 1 else dw)), p(scratch), scratch.numel(), N, L["C"], L["H"],
                                     L["W"], L["k"], L["s"], L["co"], st))

    for mode, what in ((0, "forward"), (1, "data gradient"), (2, "weight gradient (+ slab reduction)")):
        if mode == 1 and L["u8"]:
            continue
        ms = timed(lambda: call(mode))
        print(f"{name:30s} {what:36s} {ms:8.3f} ms  {flops / ms / 1e9:7.1f} TF  {flops / ms / 1e9 / 157.3:5.2f} of peak", flush=True)
    del x, out, dpre, dprev, scratch
    torch.cuda.empty_cache()
