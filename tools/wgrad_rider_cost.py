"""What a rider costs the output-stationary weight-gradient kernel: 15 blocks x 12 544 rows, no riders / narrow-output
rider on every block / both riders on every block (csrc/wgrad.h), time per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastdeepqlearning_amd import _native as nat

dev = torch.device("cuda:0")
lib = nat.load(); st = nat.current_stream(dev)
M, nprob, nslab, ldw, ldw3 = 12544, 15, 32, 262, 774
G = torch.randn(nprob * M, 256, device=dev); X = torch.randn(nprob * M, 256, device=dev)
X2 = torch.randn(nprob * M, 6, device=dev); G2 = torch.randn(nprob * M, 2, device=dev)
n_dense, n_head = nprob * 256 * ldw, nprob * 2 * ldw3
stride = n_dense + n_head
slabs = torch.zeros(nslab, stride, device=dev)
base = slabs.data_ptr()

def run(x2, g2, every=1):
    rc = lib.fdql_test_wgrad_stat_riders(nat.ptr(G), nat.ptr(X), base, M, nprob, ldw, nslab, stride,
                                         nat.ptr(X2) if x2 else None, 6, 6, base + 4 * 256, ldw, every,
                                         nat.ptr(G2) if g2 else None, 2, 2, base + 4 * n_dense, ldw3, st)
    assert rc == 0, lib.fdql_last_error().decode()

variants = (("plain", 0, 0, 1), ("G2 on all", 0, 1, 1), ("X2 on all", 1, 0, 1), ("X2 + G2 on all", 1, 1, 1),
            ("X2 + G2 on every 3rd, G2 on the rest", 1, 1, 3))
best = {v[0]: 1e9 for v in variants}
for rnd in range(6):   # interleaved rounds: the chip's clock moves with what ran before
    for name, x2, g2, every in variants:
        for _ in range(3): run(x2, g2, every)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): run(x2, g2, every)
        e1.record(); torch.cuda.synchronize()
        best[name] = min(best[name], e0.elapsed_time(e1) / 30)
for name, *_ in variants:
    print("%-45s %.4f ms / launch" % (name, best[name]), flush=True)
