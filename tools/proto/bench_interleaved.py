"""Same-process, interleaved timing of the product GEMM shapes and the prototype variants (box clocks drift by +-10 %
between invocations, so only interleaved rounds compare)."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fastdeepqlearning_amd import _native as nat
lib = nat.load(); dev = torch.device("cuda:0"); st = nat.current_stream()
M, N, K = [int(x) for x in sys.argv[1:4]]
protos = sys.argv[4:]
A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); C = torch.empty(M, N, device=dev)
def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
runs = {}
def prod(shape):
    def f():
        nat.check(lib.fdql_debug_set_gemm_dense_shape(shape))
        nat.check(lib.fdql_test_gemm(nat.ptr(A), K, 1, nat.ptr(B), K, 1, None, nat.ptr(C), N, M, N, K, 0, None, 0, 1, st))
    return f
for s in (5, 7, 8, 9): runs[f"product s{s}"] = prod(s)
import ctypes as C_
for alt in [a for a in os.environ.get("FDQL_ALT_LIBS", "").split(",") if a]:
    al = C_.CDLL(os.path.join(ROOT, "build_ab", f"libfdql_{alt}.so"))
    def mk(al, shape):
        def f():
            al.fdql_debug_set_gemm_dense_shape(shape)
            rc = al.fdql_test_gemm(C_.c_void_p(A.data_ptr()), K, 1, C_.c_void_p(B.data_ptr()), K, 1, None, C_.c_void_p(C.data_ptr()), N, M, N, K, 0, None, 0, 1, st)
            assert rc == 0
        return f
    for s_ in [int(x) for x in os.environ.get("FDQL_ALT_SHAPES", "7,9").split(",")]: runs[f"{alt} s{s_}"] = mk(al, s_)
for name in protos:
    pl = ctypes.CDLL(os.path.join(ROOT, "build_ab", f"libproto_{name}.so"))
    pl.proto_gemm_nt.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p]
    runs[f"proto {name}"] = (lambda pl: lambda: pl.proto_gemm_nt(A.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, K, st))(pl)
acc = {k: [] for k in runs}
for rnd in range(6):
    for k, f in runs.items():
        acc[k].append(timed(f))
for k, v in acc.items():
    v = sorted(v[1:])
    med = v[len(v) // 2]
    print(f"{k:16s} median {med*1e3:7.1f} us {2*M*N*K/med/1e9:6.1f} TF   (min {v[0]*1e3:.1f} max {v[-1]*1e3:.1f})")
nat.check(lib.fdql_debug_set_gemm_dense_shape(5))
