"""temporal_len 2 (the 1-step-minibatch reading of batch=256): step time eager vs. hipGraph replay, and the launch list."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda:0")
w = bench.WORKLOADS["config2"]
job = bench.Job(w, dev, 256, 2, ring_slots=200_000)
for n in (300, 300, 300):
    e = job.timed(n, 30)
    print("FDQL_GRAPH=%s  T=2 B=256: %.4f ms/step  %.1f steps/s" % (os.environ.get("FDQL_GRAPH", "0"), 1e3 * e / n, n / e), flush=True)
if len(sys.argv) > 1 and sys.argv[1] == "stages":
    rows = job.agent.profile_update(job.xp, seed=1)
    for r in rows:
        print("  %-40s %.4f ms" % (r[0], r[1]))
