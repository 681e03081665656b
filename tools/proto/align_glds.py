"""Does global_load_lds_dwordx4 accept sources that are only 4-byte aligned?  (prototype kernel, shifted operands)"""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device("cuda:0")
pl = ctypes.CDLL(os.path.join(ROOT, "build_ab", "libproto_n4.so"))
pl.proto_gemm_nt.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p]
M, N, K = 1024, 256, 256
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
for sa, sb in [(0, 0), (1, 0), (0, 2), (3, 1)]:
    Af = torch.randn(M * K + 8, device=dev); Bf = torch.randn(N * K + 8, device=dev)
    A = Af[sa:sa + M * K].view(M, K); B = Bf[sb:sb + N * K].view(N, K)
    C = torch.zeros(M, N, device=dev)
    rc = pl.proto_gemm_nt(Af.data_ptr() + 4 * sa, Bf.data_ptr() + 4 * sb, C.data_ptr(), M, N, K, st)
    torch.cuda.synchronize()
    print(f"shift A {sa} B {sb} floats: rc {rc} max err {float((C - A @ B.t()).abs().max()):.3e}")
