import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastdeepqlearning_amd import _native as nat
lib = nat.load(); dev = torch.device("cuda:0")
M, N, K = [int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (188160, 256, 256))]
variant = int(sys.argv[4]) if len(sys.argv) > 4 else 1
lib.fdql_debug_set_gemm_variant(variant)
A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); Cm = torch.empty(M, N, device=dev)
for _ in range(5):
    nat.check(lib.fdql_test_gemm(nat.ptr(A), K, 1, nat.ptr(B), K, 1, None, nat.ptr(Cm), N, M, N, K, 0, None, 0, 1, nat.current_stream()))
torch.cuda.synchronize()
