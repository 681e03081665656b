#!/usr/bin/env python3
"""Time the grouped fp32-MFMA GEMM on its own (fdql_test_gemm) for a few shapes."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastdeepqlearning_amd import _native as nat
lib = nat.load()
dev = torch.device("cuda:0")
shapes = [(4096, 4096, 4096, "nt"), (8192, 8192, 1024, "nt"), (188160, 256, 256, "nt"), (188160, 256, 262, "nt"),
          (12544, 256, 256, "nt"), (12544, 256, 1558, "nn"), (256, 256, 12544, "tn"), (125440, 256, 256, "nn")]
import itertools
for (M, N, K, form), variant in itertools.product(shapes, [int(v) for v in os.environ.get("VARIANTS", "0,1,2,3").split(",")]):
    lib.fdql_debug_set_gemm_variant(variant)
    if form == "nt":
        A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); lda, akc, ldb, bkc = K, 1, K, 1
    elif form == "nn":
        A = torch.randn(M, K, device=dev); B = torch.randn(K, N, device=dev); lda, akc, ldb, bkc = K, 1, N, 0
    else:
        A = torch.randn(K, M, device=dev); B = torch.randn(K, N, device=dev); lda, akc, ldb, bkc = M, 0, N, 0
    Cm = torch.empty(M, N, device=dev)
    ks = 8 if form == "tn" else 1
    if ks > 1: Cm = torch.empty(ks, M, N, device=dev)
    def run():
        nat.check(lib.fdql_test_gemm(nat.ptr(A), lda, akc, nat.ptr(B), ldb, bkc, None, nat.ptr(Cm), N, M, N, K, 0, None, 0, ks, nat.current_stream()))
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # fdql_test_gemm synchronises internally; time with events around the launch only
    ts = []
    for _ in range(10):
        e0.record(); run(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ms = sorted(ts)[len(ts)//2]
    err = ""
    if variant and M * N <= 70_000_000 and form != "tn":
        lib.fdql_debug_set_gemm_variant(0); ref = torch.empty_like(Cm)
        nat.check(lib.fdql_test_gemm(nat.ptr(A), lda, akc, nat.ptr(B), ldb, bkc, None, nat.ptr(ref), N, M, N, K, 0, None, 0, ks, nat.current_stream()))
        err = f" maxdiff_vs_v0={float((ref - Cm).abs().max()):.2e}"
    print(f"v{variant} {form} M={M} N={N} K={K} ksplit={ks}: {ms:.4f} ms  {2.0*M*N*K/ms/1e9:.1f} TFLOP/s "+err)
