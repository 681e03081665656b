"""Product GEMM forms (nt / nn / tn) on the register-staged 64x64 shape vs the LDS-DMA shapes, one problem, random operands."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fastdeepqlearning_amd import _native as nat
lib = nat.load(); dev = torch.device("cuda:0"); st = nat.current_stream()
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
cases = [("nt", 192000, 256, 256, 1), ("nn", 192000, 256, 256, 1), ("nn+lrelu'", 192000, 256, 256, 1), ("tn ksplit", 256, 256, 192000, 128)]
for form, M, N, K, ks in cases:
    A = torch.randn(M, K, device=dev); Bm = torch.randn(K, N, device=dev); ref = A @ Bm
    gate = torch.randn(M, N, device=dev) if "lrelu" in form else None
    if form == "nt": a, lda, akc, b, ldb, bkc = A, K, 1, Bm.t().contiguous(), K, 1
    elif form.startswith("nn"): a, lda, akc, b, ldb, bkc = A, K, 1, Bm, N, 0
    else: a, lda, akc, b, ldb, bkc = A.t().contiguous(), M, 0, Bm, N, 0
    C = torch.empty(ks, M, N, device=dev)
    line = f"{form:10s} M={M} N={N} K={K}:"
    for shape in (5, 9, 8, 7):
        nat.check(lib.fdql_debug_set_gemm_dense_shape(shape))
        def run():
            nat.check(lib.fdql_test_gemm(nat.ptr(a), lda, akc, nat.ptr(b), ldb, bkc, None, nat.ptr(C), N, M, N, K, 2 if gate is not None else 0,
                                         nat.ptr(gate) if gate is not None else None, N if gate is not None else 0, ks, st))
        ms = bench(run)
        want = ref if gate is None else ref * torch.where(gate > 0, 1.0, 0.01)
        err = float((C.sum(0) - want).abs().max() / want.abs().max())
        line += f"  s{shape} {ms*1e3:7.1f}us {2*M*N*K/ms/1e9:6.1f}TF e{err:.0e}"
    print(line)
nat.check(lib.fdql_debug_set_gemm_dense_shape(5))
