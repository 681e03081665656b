"""Prototype check: LDS-DMA staged GEMM (tools/proto/gemm_glds.hip) vs the product grouped GEMM on one NT problem."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fastdeepqlearning_amd import _native as nat
lib = nat.load()
dev = torch.device("cuda:0")
M, N, K = [int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (192000, 256, 256))]
A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev)
ref = A @ B.t()
st = nat.current_stream()
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
C0 = torch.empty(M, N, device=dev)
def prod():
    nat.check(lib.fdql_test_gemm(nat.ptr(A), K, 1, nat.ptr(B), K, 1, None, nat.ptr(C0), N, M, N, K, 0, None, 0, 1, st))
for shape in (5, 9, 8, 7):
    nat.check(lib.fdql_debug_set_gemm_dense_shape(shape))
    ms = bench(prod)
    print(f"product shape {shape}    : {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:7.1f} TF  err {float((C0-ref).abs().max()):.2e}")
for name in sys.argv[4:] or ["glds"]:
    pl = ctypes.CDLL(os.path.join(ROOT, "build_ab", f"libproto_{name}.so"))
    pl.proto_gemm_nt.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p]
    C1 = torch.zeros(M, N, device=dev)
    def proto():
        rc = pl.proto_gemm_nt(A.data_ptr(), B.data_ptr(), C1.data_ptr(), M, N, K, st)
        assert rc == 0, rc
    ms = bench(proto)
    print(f"proto {name:12s} : {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:7.1f} TF  err {float((C1-ref).abs().max()):.2e}")
