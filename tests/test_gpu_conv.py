"""Implicit-GEMM convolutions of the pixel encoder (csrc/conv.hip, BASELINE config 5) against torch conv2d in float64.

The reference has no conv encoder to compare with (franQ/Agent/components/encoder.py:16-23 is dead code: SURVEY 8d), so the
parity target is torch's conv2d / its autograd in double precision, one layer at a time through the C ABI test hook
(fdql_test_conv): forward from uint8 frames (a packed batch and the ring block read through window starts, wrap included) and
from NHWC maps, the gather-form data gradient, the output-stationary weight gradient with its bias sums - at image counts
that leave a partial last group, that give a workgroup several groups, and at the full config-5 batch's per-launch strides."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

L0 = dict(C=4, H=84, W=84, k=8, s=4, co=32, u8=True)
L1 = dict(C=32, H=20, W=20, k=4, s=2, co=64, u8=False)
L2 = dict(C=64, H=9, W=9, k=3, s=1, co=64, u8=False)
TOL = 1e-5


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _w_nchw(W, L):
    """[cout, K] in the kernel's K order -> conv2d's [cout, C, k, k]."""
    co, Cc, k = L["co"], L["C"], L["k"]
    return W.view(co, Cc, k, k) if L["u8"] else W.view(co, k, k, Cc).permute(0, 3, 1, 2)


def _w_from_nchw(g, L):
    co = L["co"]
    return g.reshape(co, -1) if L["u8"] else g.permute(0, 2, 3, 1).reshape(co, -1)


def _call(mode, L, nimg, dev, inp=None, slots=None, W=None, bias=None, dpre=None, act_prev=None, out=None):
    from fastdeepqlearning_amd import _native as nat
    lib = nat.load()
    K = L["C"] * L["k"] ** 2
    scratch = torch.empty(4096 * (L["co"] * K + L["co"]), device=dev) if mode == 2 else None
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    with torch.cuda.device(dev):
        nat.check(lib.fdql_test_conv(mode, p(inp), int(L["u8"]), p(slots), p(W), p(bias), p(dpre), p(act_prev),
                                     p(out), p(scratch), scratch.numel() if scratch is not None else 0, int(nimg), L["C"], L["H"], L["W"],
                                     L["k"], L["s"], L["co"], nat.current_stream(dev)))
    torch.cuda.synchronize(dev)
    return out


def _inputs(L, nimg, seed, dev):
    g = torch.Generator().manual_seed(seed)
    K = L["C"] * L["k"] ** 2
    W = (torch.rand(L["co"], K, generator=g) * 2 - 1) * (3.0 / K) ** 0.5
    bias = torch.randn(L["co"], generator=g) * 0.1
    if L["u8"]:
        x = torch.randint(0, 256, (nimg, L["C"], L["H"], L["W"]), generator=g, dtype=torch.uint8)
    else:
        x = torch.randn(nimg, L["H"], L["W"], L["C"], generator=g)
        x = torch.where(x > 0, x, 0.01 * x)          # what a LeakyReLU layer leaves
    return x, W, bias


def _x_nchw64(x, L):
    return x.double() / 255.0 if L["u8"] else x.double().permute(0, 3, 1, 2)


@pytest.mark.parametrize("name,L", [("layer0_u8", L0), ("layer1", L1), ("layer2", L2)])
@pytest.mark.parametrize("nimg", [1, 7, 600])
def test_conv_forward(dev, name, L, nimg):
    x, W, bias = _inputs(L, nimg, 3, dev)
    OH = (L["H"] - L["k"]) // L["s"] + 1
    out = torch.full((nimg, OH * OH, L["co"]), float("nan"), device=dev)
    _call(0, L, nimg, dev, inp=x.to(dev), W=W.to(dev), bias=bias.to(dev), out=out)
    y = torch.nn.functional.conv2d(_x_nchw64(x, L).to(dev), _w_nchw(W, L).double().to(dev), bias.double().to(dev), stride=L["s"])
    ref = torch.nn.functional.leaky_relu(y, 0.01).permute(0, 2, 3, 1).reshape(nimg, OH * OH, L["co"])
    err = float((out.double() - ref).abs().max() / ref.abs().max())
    assert err < TOL, (name, nimg, err)


def test_conv_forward_reads_the_ring_through_window_starts(dev):
    """Layer 0 on the ring's own uint8 block through one slot index per image: image (t, b) is slot (starts[b] + t) % ring_len
    (replay_memory.py:63-65; fdql_ring_window_slots computes them on the device), the window of b = 1 wraps past the ring's end."""
    L, T, B, slots = L0, 5, 3, 40
    g = torch.Generator().manual_seed(11)
    ring = torch.randint(0, 256, (slots, L["C"], L["H"], L["W"]), generator=g, dtype=torch.uint8)
    ring_len = slots - 1                                       # quirk q1: the last slot is never sampled
    starts = torch.tensor([4, ring_len - 2, 17], dtype=torch.int64)
    _, W, bias = _inputs(L, 1, 5, dev)
    idx = (starts[None, :] + torch.arange(T)[:, None]) % ring_len          # [T, B]
    x = ring[idx.reshape(-1)]
    out = torch.full((T * B, 400, L["co"]), float("nan"), device=dev)
    _call(0, L, T * B, dev, inp=ring.to(dev), slots=idx.reshape(-1).to(torch.int32).to(dev), W=W.to(dev), bias=bias.to(dev), out=out)
    y = torch.nn.functional.conv2d(_x_nchw64(x, L), _w_nchw(W, L).double(), bias.double(), stride=L["s"])
    ref = torch.nn.functional.leaky_relu(y, 0.01).permute(0, 2, 3, 1).reshape(T * B, 400, L["co"])
    assert float((out.cpu().double() - ref).abs().max() / ref.abs().max()) < TOL


@pytest.mark.parametrize("name,L", [("layer1", L1), ("layer2", L2)])
@pytest.mark.parametrize("nimg", [1, 5, 600])
def test_conv_data_gradient(dev, name, L, nimg):
    x, W, _ = _inputs(L, nimg, 7, dev)
    OH = (L["H"] - L["k"]) // L["s"] + 1
    g = torch.Generator().manual_seed(8)
    dpre = torch.randn(nimg, OH * OH, L["co"], generator=g)
    out = torch.full((nimg, L["H"] * L["W"], L["C"]), float("nan"), device=dev)
    _call(1, L, nimg, dev, W=W.to(dev), dpre=dpre.to(dev), act_prev=x.to(dev), out=out)
    d = torch.nn.functional.conv_transpose2d(dpre.double().to(dev).view(nimg, OH, OH, L["co"]).permute(0, 3, 1, 2),
                                             _w_nchw(W, L).double().to(dev), stride=L["s"])
    gate = torch.where(x.to(dev) > 0, 1.0, 0.01).double()
    ref = d.permute(0, 2, 3, 1) * gate
    err = float((out.double().view_as(ref) - ref).abs().max() / ref.abs().max())
    assert err < TOL, (name, nimg, err)


@pytest.mark.parametrize("name,L", [("layer0_u8", L0), ("layer1", L1), ("layer2", L2)])
@pytest.mark.parametrize("nimg", [1, 5, 777])
def test_conv_weight_gradient(dev, name, L, nimg):
    """dW and db over all images and positions; 777 images = 3+ groups per workgroup, an odd count (a partial last group)."""
    x, W, _ = _inputs(L, nimg, 9, dev)
    OH = (L["H"] - L["k"]) // L["s"] + 1
    K = L["C"] * L["k"] ** 2
    g = torch.Generator().manual_seed(10)
    dpre = torch.randn(nimg, OH * OH, L["co"], generator=g)
    out = torch.full((L["co"] * K + L["co"],), float("nan"), device=dev)
    _call(2, L, nimg, dev, inp=x.to(dev), dpre=dpre.to(dev), out=out)
    w64 = _w_nchw(W, L).double().to(dev).requires_grad_(True)
    y = torch.nn.functional.conv2d(_x_nchw64(x, L).to(dev), w64, stride=L["s"])
    (y * dpre.double().to(dev).view(nimg, OH, OH, L["co"]).permute(0, 3, 1, 2)).sum().backward()
    ref_w = _w_from_nchw(w64.grad, L)
    ref_b = dpre.double().sum((0, 1)).to(dev)
    got_w, got_b = out[:L["co"] * K].double().view(L["co"], K), out[L["co"] * K:].double()
    ew = float((got_w - ref_w).abs().max() / ref_w.abs().max())
    eb = float((got_b - ref_b).abs().max() / ref_b.abs().max())
    assert ew < 2 * TOL and eb < 2 * TOL, (name, nimg, ew, eb)


def test_conv_weight_gradient_through_window_starts(dev):
    L, T, B, slots = L0, 4, 3, 30
    g = torch.Generator().manual_seed(12)
    ring = torch.randint(0, 256, (slots, L["C"], L["H"], L["W"]), generator=g, dtype=torch.uint8)
    ring_len = slots - 1
    starts = torch.tensor([ring_len - 1, 3, 11], dtype=torch.int64)
    idx = (starts[None, :] + torch.arange(T)[:, None]) % ring_len
    x = ring[idx.reshape(-1)]
    K, nimg = L["C"] * 64, T * B
    dpre = torch.randn(nimg, 400, L["co"], generator=g)
    out = torch.full((L["co"] * K + L["co"],), float("nan"), device=dev)
    _call(2, L, nimg, dev, inp=ring.to(dev), slots=idx.reshape(-1).to(torch.int32).to(dev), dpre=dpre.to(dev), out=out)
    w64 = torch.zeros(L["co"], L["C"], 8, 8, dtype=torch.float64, requires_grad=True)
    y = torch.nn.functional.conv2d(_x_nchw64(x, L), w64, stride=L["s"])
    (y * dpre.double().view(nimg, 20, 20, L["co"]).permute(0, 3, 1, 2)).sum().backward()
    ref = w64.grad.reshape(L["co"], K)
    got = out[:L["co"] * K].cpu().double().view(L["co"], K)
    assert float((got - ref).abs().max() / ref.abs().max()) < 2 * TOL


def test_conv_rejects_unknown_geometry(dev):
    from fastdeepqlearning_amd import _native as nat
    lib = nat.load()
    x = torch.zeros(2, 12, 12, 8, device=dev)
    rc = lib.fdql_test_conv(0, C.c_void_p(x.data_ptr()), 0, None, None, None, None, None, None, None, 0, 2, 8, 12, 12, 3, 1, 16, None)
    assert rc == nat.FDQL_EINVAL if hasattr(nat, "FDQL_EINVAL") else rc != 0
