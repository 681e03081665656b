"""Does the start-up ramp of the step time return after the GPU idled?  (It does: device power management.)"""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
dev = torch.device("cuda:0")
w = bench.WORKLOADS["config2"]
job = bench.Job(w, dev, 256, 50)
def run(n, first):
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    torch.cuda.synchronize(); evs[0].record()
    for i in range(n):
        job.step(first + i); evs[i + 1].record()
    torch.cuda.synchronize()
    return [evs[i].elapsed_time(evs[i + 1]) for i in range(n)]
a = run(40, 0)
print("start :", " ".join("%.3f" % m for m in a[:24]))
time.sleep(0.1)
b = run(30, 100)
print("after 100 ms idle:", " ".join("%.3f" % m for m in b[:24]))
time.sleep(1.0)
c = run(30, 200)
print("after 1 s idle   :", " ".join("%.3f" % m for m in c[:24]))
