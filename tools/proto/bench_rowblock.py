"""Prototype check: row-block stationary layer (tools/proto/rowblock.hip) vs the product grouped GEMM on
C = lrelu(A W^T + b), N = 256.  Interleaved rounds in one process (box clocks drift between invocations)."""
import ctypes, os, sys, torch
os.environ.setdefault("FDQL_ROWGEMM_FORMS", "7")   # the forward form is off by default in the update
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fastdeepqlearning_amd import _native as nat
lib = nat.load(); dev = torch.device("cuda:0"); st = nat.current_stream()
M, K = [int(x) for x in (sys.argv[1:3] if len(sys.argv) > 2 else (192000, 256))]
names = sys.argv[3:] or ["rowblock"]
N = 256
A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
ref = torch.nn.functional.leaky_relu(A.double() @ W.double().t() + b.double(), 0.01)
def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
C0 = torch.empty(M, N, device=dev)
runs = {}
def prod():
    nat.check(lib.fdql_test_gemm(nat.ptr(A), K, 1, nat.ptr(W), K, 1, nat.ptr(b), nat.ptr(C0), N, M, N, K, 1, None, 0, 1, st))
runs["product 64x64"] = (prod, C0)
C2 = torch.zeros(M, N, device=dev)
def rows():
    nat.check(lib.fdql_test_rowgemm(nat.ptr(A), None, 0, None, 0, nat.ptr(W), K, None, None, nat.ptr(b), nat.ptr(C2), None, None, None,
                                    None, 0, 0, None, None, M, 1, 0, 0, 0, 0, None, None, 0, None, st))
if K == 256: runs["product rows"] = (rows, C2)
for name in names:
    pl = ctypes.CDLL(os.path.join(ROOT, "build_ab", f"libproto_{name}.so"))
    pl.proto_rowblock.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 2 + [ctypes.c_void_p]
    C1 = torch.zeros(M, N, device=dev)
    Wk = W
    if "pk" in name:   # weights pre-packed in MFMA-fragment order [wave][tn][g][j][lh][li][c] (K = 256)
        Wk = W.view(4, 2, 32, K // 32, 2, 4, 4).permute(0, 1, 3, 5, 4, 2, 6).contiguous()
    def mk(pl, C1, Wk=Wk):
        def f():
            rc = pl.proto_rowblock(A.data_ptr(), Wk.data_ptr(), b.data_ptr(), C1.data_ptr(), M, K, st)
            assert rc == 0, rc
        return f
    runs[f"proto {name}"] = (mk(pl, C1), C1)
acc = {k: [] for k in runs}
for rnd in range(6):
    for k, (f, _) in runs.items():
        acc[k].append(timed(f))
for k, v in acc.items():
    err = float((runs[k][1].double() - ref).abs().max())
    v = sorted(v[1:]); med = v[len(v) // 2]
    print(f"M={M} K={K} {k:18s} median {med*1e3:7.1f} us {2*M*N*K/med/1e9:6.1f} TF  (min {v[0]*1e3:.1f} max {v[-1]*1e3:.1f}) max|err| {err:.2e}", flush=True)
