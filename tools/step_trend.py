"""Per-step HIP-event times of the first 80 steps of bench.py's Job right after set-up (config 2): the start-up clock ramp."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
w = bench.WORKLOADS["config2"]
job = bench.Job(w, dev, 256, 50)
evs = [torch.cuda.Event(enable_timing=True) for _ in range(81)]
torch.cuda.synchronize()
evs[0].record()
for i in range(80):
    job.step(i)
    evs[i + 1].record()
torch.cuda.synchronize()
ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(80)]
print("per-step ms:", " ".join("%.3f" % m for m in ms))
