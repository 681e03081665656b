"""Row-block kernel (fdql_test_rowgemm) on `ninst` instances of one layer, against the 64x64-tile grouped kernel on the
same work: bench_rows.py [M_per_instance] [ninst] [hf]"""
import os, sys, torch
os.environ.setdefault("FDQL_ROWGEMM_FORMS", "7")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fastdeepqlearning_amd import _native as nat
lib = nat.load(); dev = torch.device("cuda:0"); st = nat.current_stream()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 12544
ninst = int(sys.argv[2]) if len(sys.argv) > 2 else 15
hf = int(sys.argv[3]) if len(sys.argv) > 3 else 0
K = N = 256
A = torch.randn(ninst * M, K, device=dev); W = torch.randn(ninst, N, K, device=dev) / 16; b = torch.randn(ninst, N, device=dev)
C = torch.zeros(ninst * M, N, device=dev); C0 = torch.zeros(ninst * M, N, device=dev)
hw = torch.randn(ninst, 2, 774, device=dev); hout = torch.zeros(ninst, 8, M, 2, device=dev)
def rows():
    nat.check(lib.fdql_test_rowgemm(nat.ptr(A), None, 0, None, 0, nat.ptr(W), K, None, None, nat.ptr(b), nat.ptr(C), None, None, None,
                                    nat.ptr(hw) if hf else None, 774, 2 if hf else 0, nat.ptr(hout) if hf else None, None, M, ninst, 0, 0, 0, 8, None, None, 0, None, st))
def tiles():
    for i in range(ninst):
        nat.check(lib.fdql_test_gemm(A[i * M:].data_ptr(), K, 1, W[i].data_ptr(), K, 1, b[i].data_ptr(), C0[i * M:].data_ptr(), N, M, N, K, 1, None, 0, 1, st))
def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
ref = torch.nn.functional.leaky_relu(torch.bmm(A.view(ninst, M, K).double(), W.double().transpose(1, 2)) + b.double()[:, None, :], 0.01).view(-1, N)
rows(); torch.cuda.synchronize()
print("rows max|err|", float((C.double() - ref).abs().max()))
if hf:
    x = ref.view(ninst, M, 8, 32); w = hw[:, :, :256].double().view(ninst, 2, 8, 32)
    want = torch.einsum("impc,iqpc->ipmq", x, w)
    print("hf max|err|", float((hout.double() - want).abs().max()))
for rnd in range(4):
    t = timed(rows)
    print(f"rows  M={M} x {ninst}: {t*1e3:7.1f} us  {2*ninst*M*N*K/t/1e9:6.1f} TF", flush=True)
if not hf:
    tiles(); torch.cuda.synchronize()
    print("tiles max|err|", float((C0.double() - ref).abs().max()))
