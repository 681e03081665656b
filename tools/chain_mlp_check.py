"""Debug / parity helper: one SkipHeadMLP forward through the chain kernel vs torch (fp64 reference)."""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fastdeepqlearning_amd import _native as nat


def run(rows, din, hid, dout, dev, seed=0):
    lib = nat.load()
    g = torch.Generator().manual_seed(seed)
    pad4 = lambda n: (n + 3) // 4 * 4
    x = torch.randn(rows, din, generator=g)
    parts, Ws, bs = [], [], []
    prev = din
    for h in hid:
        W = torch.randn(h, prev, generator=g) / prev ** 0.5; b = torch.randn(h, generator=g) * 0.1
        Ws.append(W); bs.append(b)
        for t in (W, b):
            f = torch.zeros(pad4(t.numel())); f[:t.numel()] = t.reshape(-1); parts.append(f)
        prev = h
    ld = din + sum(hid)
    Wh = torch.randn(dout, ld, generator=g) / ld ** 0.5; bh = torch.randn(dout, generator=g) * 0.1
    for t in (Wh, bh):
        f = torch.zeros(pad4(t.numel())); f[:t.numel()] = t.reshape(-1); parts.append(f)
    w = torch.cat(parts).to(dev)
    xd = x.to(dev)
    hs = [torch.full((rows, h), float("nan"), device=dev) for h in hid]
    out = torch.full((rows, dout), float("nan"), device=dev)
    hp = (C.c_void_p * max(len(hid), 1))(*[t.data_ptr() for t in hs])
    ha = (C.c_int32 * max(len(hid), 1))(*hid)
    nat.check(lib.fdql_test_chain_mlp(nat.ptr(xd), rows, din, ha, len(hid), dout, nat.ptr(w), hp, nat.ptr(out), nat.current_stream()))
    feats, h = [x.double()], x.double()
    errs = []
    for i, (W, b) in enumerate(zip(Ws, bs)):
        h = torch.nn.functional.leaky_relu(h @ W.double().t() + b.double(), 0.01)
        feats.append(h)
        errs.append(float((hs[i].cpu().double() - h).abs().max() / h.abs().max()))
    ref = torch.cat(feats, -1) @ Wh.double().t() + bh.double()
    errs.append(float((out.cpu().double() - ref).abs().max() / ref.abs().max()))
    return errs


if __name__ == "__main__":
    dev = torch.device("cuda:0")
    for cfg in [(64, 8, [32], 2), (64, 32, [64], 4), (100, 5, [32], 24), (200, 262, [256, 256], 2), (130, 17, [256], 256),
                (77, 33, [18, 21], 5), (64, 40, [], 3), (3000, 262, [256, 256], 25)]:
        print(cfg, ["%.2e" % e for e in run(*cfg, dev)], flush=True)
