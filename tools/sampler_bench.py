"""HBM throughput of the windowed gather (k_draw_starts + k_gather_windows) at the BASELINE row sizes.
Algorithmic bytes = 2 * T * B * rowbytes + 8 * B."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from fastdeepqlearning_amd.core import NativeRing

CASES = [("config 2 (obs 17, act 6)", [17, 6, 1, 1, 1, 1, 1, 1], 1_000_000, 50, 256),
         ("config 3 (obs 28 + 2 goals of 10)", [28, 10, 10, 6, 1, 1, 1, 1, 1, 1], 1_000_000, 50, 256),
         ("config 4 (obs 376, act 17), B=1024", [376, 17, 1, 1, 1, 1, 1, 1], 2_000_000, 50, 1024),
         ("config 4, B=128 (per GPU of 8)", [376, 17, 1, 1, 1, 1, 1, 1], 2_000_000, 50, 128),
         ("config 5 (4x84x84 frames as f32), T=8, B=512", [28224, 1, 1, 1, 1, 1], 200_000, 8, 512)]


def main():
    dev = torch.device("cuda:0")
    reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 50
    for name, dims, maxlen, T, B in CASES:
        ring = NativeRing(maxlen, dims, dev)
        rowf = sum(dims)
        chunk = max(1, min(maxlen, (256 << 20) // (rowf * 4)))
        done = 0
        while done < maxlen:                      # fill on the device: contents do not matter for bandwidth
            n = min(chunk, maxlen - done)
            ring.add_rows(torch.rand(n, rowf, device=dev))
            done += n
        outs = [torch.empty(T, B, d, device=dev) for d in dims]
        for i in range(5):
            ring.sample_windows(T, B, seed=1, counter=i, outs=outs)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(reps):
            ring.sample_windows(T, B, seed=1, counter=100 + i, outs=outs)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        nbytes = 2.0 * T * B * rowf * 4 + 8 * B
        # the same bytes as ONE plain device-to-device copy (hipMemcpyAsync DtoD: reads n, writes n), timed the same way:
        # what "the rate of a device copy" means on this box for this size
        src = torch.empty(int(nbytes // 8), dtype=torch.float32, device=dev).normal_()
        dst = torch.empty_like(src)
        for _ in range(3):
            dst.copy_(src)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            dst.copy_(src)
        e1.record()
        torch.cuda.synchronize()
        cms = e0.elapsed_time(e1) / reps
        print(f"{name:48s} {nbytes / 1e6:9.2f} MB/sample  {ms * 1e3:8.1f} us  {nbytes / ms / 1e6:8.1f} GB/s "
              f"({nbytes / ms / 1e6 / 8000 * 100:.1f} % of 8 TB/s)   device copy of the same bytes: {cms * 1e3:8.1f} us "
              f"{nbytes / cms / 1e6:8.1f} GB/s", flush=True)
        del src, dst
        del ring, outs
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
