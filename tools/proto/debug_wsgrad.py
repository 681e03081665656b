import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fastdeepqlearning_amd import _native as nat
lib = nat.load(); dev = torch.device("cuda:0"); st = nat.current_stream(dev)
M, ninst, Q = int(sys.argv[1]), int(sys.argv[2]), 2
g = torch.Generator().manual_seed(1)
rnd = lambda *s: torch.randn(*s, generator=g)
R = M * ninst
bmm = lambda x, w, k: torch.bmm(x.double().view(ninst, M, k), w.double())
W0 = rnd(ninst, 256, 256) / 16
A1, W1, ref = rnd(R, Q), rnd(ninst, Q, 256), rnd(R, 256)
fzh, fzw = rnd(R, 256), rnd(ninst, Q, 300)
pre1 = bmm(A1, fzw[:, :, :256], Q)
a0 = torch.where(fzh.double().view(ninst, M, 256) > 0, pre1, 0.01 * pre1)
pre = torch.bmm(a0, W0.double()) + bmm(A1, W1, Q)
want = torch.where(ref.double().view(ninst, M, 256) > 0, pre, 0.01 * pre)
d = lambda t: t.to(dev).contiguous()
A0_d = torch.full((R, 256), float("nan"), device=dev)
C = torch.full((R, 256), float("nan"), device=dev)
cs = torch.zeros(ninst, M // 64, 256, device=dev); fcs = torch.zeros(ninst, M // 64, 256, device=dev)
A1_d, W0_d, W1_d, ref_d, fzh_d, fzw_d = map(d, (A1, W0, W1, ref, fzh, fzw))
rc = lib.fdql_test_rowgemm(nat.ptr(A0_d), nat.ptr(A1_d), Q, None, 0, nat.ptr(W0_d), 256, nat.ptr(W1_d), None, None, nat.ptr(C), None, nat.ptr(ref_d),
                           nat.ptr(cs), None, 300, 0, None, None, M, ninst, 1, 1, 0, 8, nat.ptr(fzh_d), nat.ptr(fzw_d), 300, nat.ptr(fcs), st)
assert rc == 0, lib.fdql_last_error().decode()
torch.cuda.synchronize()
eA = (A0_d.double().cpu().view(ninst, M, 256) - a0).abs()
eC = (C.double().cpu().view(ninst, M, 256) - want).abs()
print("A0 err max", float(eA.max()), "C err max", float(eC.max()), "scale", float(want.abs().max()))
bad = (eC > 1e-3).nonzero()
print("bad C count", len(bad), "of", eC.numel())
if len(bad):
    print("bad rows", sorted(set(bad[:, 1].tolist()))[:40])
    print("bad cols", sorted(set(bad[:, 2].tolist()))[:40])
badA = (eA > 1e-4).nonzero() | torch.isnan(A0_d.cpu().view(ninst, M, 256)).nonzero() if False else (torch.isnan(eA) | (eA > 1e-4)).nonzero()
print("bad A count", len(badA))
if len(badA):
    print("bad A rows", sorted(set(badA[:, 1].tolist()))[:40]); print("bad A cols", sorted(set(badA[:, 2].tolist()))[:20])
print("colsum err", float((cs.double().cpu().sum(1) - want.sum(1)).abs().max()), "fcs err", float((fcs.double().cpu().sum(1) - a0.sum(1)).abs().max()))
# which dY did the in-loop staging use?  A0[r] / gate(h[r]) = d0 * fzw[0] + d1 * fzw[1]  -> least squares for (d0, d1)
A0c = A0_d.double().cpu().view(ninst, M, 256)
gate = torch.where(fzh.double().view(ninst, M, 256) > 0, 1.0, 0.01)
Wq = fzw[0, :, :256].double().t()                      # [256, 2]
for r in [32, 33, 36, 40, 63]:
    sol = torch.linalg.lstsq(Wq, (A0c[0, r] / gate[0, r]).unsqueeze(1)).solution.flatten()
    dist = ((A1.double()[:M] - sol) ** 2).sum(1)
    print("row", r, "used dY ~", sol.tolist(), "own", A1[r].tolist(), "closest row", int(dist.argmin()), float(dist.min()))
