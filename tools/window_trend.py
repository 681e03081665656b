#!/usr/bin/env python3
"""Steps/s of consecutive 20-step windows right after start-up (what the driver's `--steps 20 --warmup 5` sees)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
w = bench.WORKLOADS["config2"]
job = bench.Job(w, dev, w["B"], w["T"])
el = job.timed(20, 5)
out = [20 / el]
first = 25
for _ in range(12):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(20): job.step(first + i)
    torch.cuda.synchronize(); out.append(20 / (time.perf_counter() - t0)); first += 20
print("20-step windows, steps/s:", [round(x, 1) for x in out])
