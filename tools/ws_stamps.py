#!/usr/bin/env python3
"""Where a workgroup of the weight-stationary forward kernel (csrc/wstat.hip) spends its life: shader-clock stamps at
the seams (entry / operands + first image in / first tile done / last tile done / flushed) of a diagnostic build
(-DWS_STAMPS, built by this script into tools/proto/build/libfdql_stamps.so when run without a GPU; on the GPU box it
loads that build and times one layer of M rows x ninst instances).

    python tools/ws_stamps.py build          # here (no GPU): compile the diagnostic library
    python tools/ws_stamps.py [M] [ninst]    # on the GPU box"""
import ctypes, os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tools", "proto", "build", "libfdql_stamps.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    td = tempfile.mkdtemp()
    src = os.path.join(ROOT, "fastdeepqlearning_amd", "csrc")
    dst = os.path.join(td, "fastdeepqlearning_amd", "csrc")
    shutil.copytree(src, dst, ignore=shutil.ignore_patterns("*.o"))
    shutil.copytree(os.path.join(ROOT, "include"), os.path.join(td, "include"))
    subprocess.run(["make", "-C", dst, "-j8", "EXTRA=-DWS_STAMPS", "OUT=" + OUT], check=True)
    print("built", OUT)
    sys.exit(0)
os.environ["FDQL_LIB_PATH"] = OUT
sys.path.insert(0, ROOT)
import numpy as np, torch
from fastdeepqlearning_amd import _native as nat
lib = nat.load(); dev = torch.device("cuda:0"); st = nat.current_stream(dev)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 12544
ninst = int(sys.argv[2]) if len(sys.argv) > 2 else 15
R = M * ninst
A0 = torch.randn(R, 256, device=dev); W0 = torch.randn(ninst, 256, 256, device=dev) / 16; bias = torch.randn(ninst, 256, device=dev)
hfw = torch.randn(ninst, 2, 300, device=dev); C = torch.empty(R, 256, device=dev); hfo = torch.empty(ninst, 8, M, 2, device=dev)
def run():
    rc = lib.fdql_test_rowgemm(nat.ptr(A0), None, 0, None, 0, nat.ptr(W0), 256, None, None, nat.ptr(bias), nat.ptr(C), None, None, None,
                               nat.ptr(hfw), 300, 2, nat.ptr(hfo), None, M, ninst, 0, 0, 0, 8, None, None, 0, None, st)
    assert rc == 0, lib.fdql_last_error().decode()
for _ in range(5): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"M={M} ninst={ninst}: {ms*1e3:.1f} us/launch, {2.0*R*256*256/ms/1e9:.1f} TF")
raw = ctypes.CDLL(OUT)
buf = (ctypes.c_uint64 * 8192)()
n = raw.wstat_debug_stamps(buf, 8192)
v = np.array(buf[:n], dtype=np.float64).reshape(-1, 8)
v = v[v[:, 0] > 0]
d = lambda a, b: (v[:, b] - v[:, a])
tiles = v[:, 6]
print(f"{len(v)} workgroups, tiles per workgroup {tiles.min():.0f}..{tiles.max():.0f}")
print(f"cycles: operands + first image {d(0,1).mean():9.0f} (max {d(0,1).max():.0f})")
print(f"        first tile             {d(1,2).mean():9.0f}")
print(f"        steady tiles, each     {(d(2,3) / np.maximum(tiles - 1, 1)).mean():9.0f}   (ideal 256 MFMA x 64 = 16384)")
print(f"        flush                  {d(3,4).mean():9.0f}")
print(f"        whole life             {d(0,4).mean():9.0f} (max {d(0,4).max():.0f}) = {d(0,4).max()/ms/1e6:.2f} GHz x launch time")
