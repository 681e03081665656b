"""Where does a row-block prototype differ from the fp64 reference?  usage: check_rb.py name [M]"""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fastdeepqlearning_amd import _native as nat
nat.load(); dev = torch.device("cuda:0"); st = nat.current_stream()
name = sys.argv[1]; M = int(sys.argv[2]) if len(sys.argv) > 2 else 192000; K = N = 256
torch.manual_seed(0)
A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
ref = torch.nn.functional.leaky_relu(A.double() @ W.double().t() + b.double(), 0.01)
pl = ctypes.CDLL(os.path.join(ROOT, "build_ab", f"libproto_{name}.so"))
pl.proto_rowblock.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 2 + [ctypes.c_void_p]
for trial in range(3):
    C = torch.full((M, N), float("nan"), device=dev)
    assert pl.proto_rowblock(A.data_ptr(), W.data_ptr(), b.data_ptr(), C.data_ptr(), M, K, st) == 0
    torch.cuda.synchronize()
    bad = ~((C.double() - ref).abs() < 1e-4)
    nb = int(bad.sum())
    print(f"trial {trial}: {nb} bad elements of {M*N}")
    if nb:
        rows = bad.any(1).nonzero().flatten()
        blks = torch.unique(rows // 64)
        print("  bad rows:", len(rows), "bad blocks:", len(blks), "first blocks:", blks[:16].tolist())
        print("  rows within block:", torch.unique(rows % 64).tolist()[:64])
        cols = bad.any(0).nonzero().flatten()
        print("  bad cols:", len(cols), cols[:16].tolist())
        r = int(rows[0]); c = int(bad[r].nonzero()[0])
        print("  e.g. row", r, "col", c, "got", float(C[r, c]), "want", float(ref[r, c]))
