"""Per-phase cycle split of the register-staged GEMM main loop (instrumented build from make_timing_build.py:
s_memtime at loop top / after the operand requests / after the MFMA k-steps / after the LDS stores / after the barrier)."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["FDQL_LIB_PATH"] = os.path.join(ROOT, "build_ab", "libfdql_timing.so")
sys.path.insert(0, ROOT)
from fastdeepqlearning_amd import _native as nat
lib = nat.load(); dev = torch.device("cuda:0"); st = nat.current_stream()
raw = ctypes.CDLL(os.environ["FDQL_LIB_PATH"])
raw.fdql_debug_phase_times.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
def phases(nslots):
    out = (ctypes.c_uint64 * 8)()
    torch.cuda.synchronize()
    assert raw.fdql_debug_phase_times(out, nslots) == 0
    return list(out)
for form, M, N, K in [("nt", 192000, 256, 256), ("nt", 192000, 256, 512), ("nn", 192000, 256, 256), ("nt", 12544, 256, 512)]:
    A = torch.randn(M, K, device=dev); Bm = torch.randn(K, N, device=dev); C = torch.empty(M, N, device=dev)
    if form == "nt": a, lda, akc, b, ldb, bkc = A, K, 1, Bm.t().contiguous(), K, 1
    else: a, lda, akc, b, ldb, bkc = A, K, 1, Bm, N, 0
    def run(): nat.check(lib.fdql_test_gemm(nat.ptr(a), lda, akc, nat.ptr(b), ldb, bkc, None, nat.ptr(C), N, M, N, K, 0, None, 0, 1, st))
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    p = phases(((M + 63) // 64) * ((N + 63) // 64) * 4)
    tot = sum(p[:4]); n = max(p[4], 1)
    print(f"{form} {M}x{N}x{K}: {e0.elapsed_time(e1)*1e3:7.1f} us; per wave-iteration cycles: request {p[0]/n:6.0f} (locate {p[5]/n:4.0f})  mfma {p[1]/n:6.0f}  "
          f"load-wait+store {p[2]/n:6.0f}  barrier {p[3]/n:6.0f}  (sum {tot/n:6.0f}; shares {p[0]/tot:.2f} {p[1]/tot:.2f} {p[2]/tot:.2f} {p[3]/tot:.2f})")
