"""Prototype check: weight-stationary row-block layer (tools/proto/wstat.hip variants) vs rowblock3 and the product kernels on
C = lrelu(A W^T + b), N = K = 256.  Interleaved rounds in one process (box clocks drift between invocations)."""
import ctypes, glob, os, sys, torch
os.environ.setdefault("FDQL_ROWGEMM_FORMS", "7")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from fastdeepqlearning_amd import _native as nat
lib = nat.load(); dev = torch.device("cuda:0"); st = nat.current_stream()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 192000
names = sys.argv[2:] or sorted(os.path.basename(p)[:-3] for p in glob.glob(os.path.join(HERE, "build", "*.so")))
N = K = 256
A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
ref = torch.nn.functional.leaky_relu(A.double() @ W.double().t() + b.double(), 0.01)
def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
runs = {}
C0 = torch.empty(M, N, device=dev)
def prod():
    nat.check(lib.fdql_test_gemm(nat.ptr(A), K, 1, nat.ptr(W), K, 1, nat.ptr(b), nat.ptr(C0), N, M, N, K, 1, None, 0, 1, st))
runs["product 64x64 tiles"] = (prod, C0)
C2 = torch.zeros(M, N, device=dev)
def rows():
    nat.check(lib.fdql_test_rowgemm(nat.ptr(A), None, 0, None, 0, nat.ptr(W), K, None, None, nat.ptr(b), nat.ptr(C2), None, None, None,
                                    None, 0, 0, None, None, M, 1, 0, 0, 0, 0, None, None, 0, None, st))
if M % 64 == 0: runs["product k_rowgemm"] = (rows, C2)
for name in names:
    pl = ctypes.CDLL(os.path.join(HERE, "build", f"{name}.so"))
    pl.proto_rowblock.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 2 + [ctypes.c_void_p]
    C1 = torch.zeros(M, N, device=dev)
    def mk(pl, C1):
        def f():
            rc = pl.proto_rowblock(A.data_ptr(), W.data_ptr(), b.data_ptr(), C1.data_ptr(), M, K, st)
            assert rc == 0, rc
        return f
    runs[name] = (mk(pl, C1), C1)
acc = {k: [] for k in runs}
for rnd in range(6):
    for k, (f, _) in runs.items():
        acc[k].append(timed(f))
for k, v in acc.items():
    err = float((runs[k][1].double() - ref).abs().max())
    v = sorted(v[1:]); med = v[len(v) // 2]
    print(f"M={M} {k:22s} median {med*1e3:7.1f} us {2*M*N*K/med/1e9:6.1f} TF  (min {v[0]*1e3:.1f} max {v[-1]*1e3:.1f}) max|err| {err:.2e}", flush=True)
