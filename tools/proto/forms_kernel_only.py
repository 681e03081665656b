import os, sys, torch
sys.path.insert(0, "/root/repo")
from fastdeepqlearning_amd import _native as nat
lib = nat.load(); dev = torch.device("cuda:0"); st = nat.current_stream()
for form, M, N, K in [("nt", 192000, 256, 256), ("nn", 192000, 256, 256), ("nt", 192000, 256, 512), ("nn", 192000, 256, 512)]:
    A = torch.randn(M, K, device=dev); Bm = torch.randn(K, N, device=dev); C = torch.empty(M, N, device=dev)
    if form == "nt": a, lda, akc, b, ldb, bkc = A, K, 1, Bm.t().contiguous(), K, 1
    else: a, lda, akc, b, ldb, bkc = A, K, 1, Bm, N, 0
    for _ in range(8):
        nat.check(lib.fdql_test_gemm(nat.ptr(a), lda, akc, nat.ptr(b), ldb, bkc, None, nat.ptr(C), N, M, N, K, 0, None, 0, 1, st))
    torch.cuda.synchronize()
    print(form, K, flush=True)
