#!/usr/bin/env python3
"""Does a collective's channel kernel get compute units beside the data-parallel step's persistent launches - and what does
it cost them?    One GPU, no node needed.

The data-parallel step (Agent/deepQlearning.py::_distributed_step, bench.py::DPRun.step) all-reduces the critics' share of the
gradient arena on a side stream while FDQL_PHASE_GRAD_REST runs.  The dense kernels of that phase are persistent launches of
one 256-thread workgroup per CU with ~158 KB of LDS each: nothing else fits on a CU beside one.  Here a STAND-IN for RCCL's
channel kernels (fdql_debug_side_copy: W workgroups of 256 threads copying the bucket `passes` times) is launched at the two
bucket points of the real step, on the side stream with the real event chain, and the step is timed

    none      no side kernel (compute only)
    side W    the stand-in on the side stream, W workgroups (RCCL on MI300-class parts runs 16-64 channels)
    serial    the same stand-in on the MAIN stream (no overlap by construction: the upper bound of its cost)

    python tools/dp_overlap.py                       # table
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/dp_overlap.py --trace-run
    python tools/dp_overlap.py --summarize DIR       # which launches each stand-in kernel overlapped, from the timestamps
"""
import argparse
import ctypes as C
import csv
import glob
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_job(name, B, world):
    import torch
    import bench
    w = bench.WORKLOADS[name]
    return bench.Job(w, torch.device("cuda:0"), B, w["T"], world=world, rank=0, ring_slots=200_000)


class Stepper:
    def __init__(self, job, workgroups, passes, mode, hold_us=0, lds=0):
        import torch
        from fastdeepqlearning_amd import _native as nat
        self.torch, self.nat, self.lib = torch, nat, nat.load()
        self.job, self.W, self.P, self.mode, self.hold, self.lds = job, workgroups, passes, mode, hold_us, lds
        self.dev = job.dev
        self.side = torch.cuda.Stream(self.dev)
        self.ev = [torch.cuda.Event() for _ in range(3)]
        self.bucket = job.agent.grad_bucket()
        self.scratch = torch.empty_like(job.agent.grads)

    def copy(self, lo, hi, stream):
        if self.mode == "none":
            return
        g, d = self.job.agent.grads, self.scratch
        lo4 = (lo + 3) // 4 * 4
        n = (hi - lo4) // 4 * 4
        self.nat.check(self.lib.fdql_debug_side_copy(C.c_void_p(g.data_ptr() + 4 * lo4), C.c_void_p(d.data_ptr() + 4 * lo4), n,
                                                     self.W, self.P, self.hold, self.lds, C.c_void_p(stream.cuda_stream)))

    def step(self, i):
        torch, nat, job, agent = self.torch, self.nat, self.job, self.job.agent
        main = torch.cuda.current_stream(self.dev)
        side = main if self.mode == "serial" else self.side
        e_a, e_b, e_red = self.ev
        n = agent.grads.numel()
        k = job.sample(i)
        agent.update(job.xps[k], seed=job.seed, phase=nat.PHASE_GRAD_CRITICS)
        e_a.record(main)
        side.wait_event(e_a)
        self.copy(self.bucket, n, side)
        agent.update(None, phase=nat.PHASE_GRAD_REST)
        e_b.record(main)
        side.wait_event(e_b)
        self.copy(0, self.bucket, side)
        e_red.record(side)
        main.wait_event(e_red)
        agent.update(None, phase=nat.PHASE_APPLY)

    def timed(self, steps, warmup=10):
        torch = self.torch
        for i in range(warmup):
            self.step(i)
        torch.cuda.synchronize(self.dev)
        t0 = time.perf_counter()
        for i in range(steps):
            self.step(warmup + i)
        torch.cuda.synchronize(self.dev)
        return 1e3 * (time.perf_counter() - t0) / steps


def standalone_copy_ms(job, W, P, lo, hi, hold_us=0, lds=0):
    """the stand-in alone on an idle chip"""
    import torch
    st = Stepper(job, W, P, "side", hold_us, lds)
    main = torch.cuda.current_stream(job.dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5):
        st.copy(lo, hi, main)
    torch.cuda.synchronize(job.dev)
    e0.record()
    for _ in range(50):
        st.copy(lo, hi, main)
    e1.record()
    torch.cuda.synchronize(job.dev)
    return e0.elapsed_time(e1) / 50


def table(args):
    """Per workload: compute only; then the stand-in in its two characters - bandwidth-bound (copy passes, no hold) and
    latency-bound (one pass, then resident for hold_us: what a ring all-reduce over xGMI mostly is) - beside / serial."""
    import json
    import torch
    for name, B in (("config2", 256), ("config4", 128), ("config4", 512)):
        job = make_job(name, B, 2)
        n, b = job.agent.grads.numel(), job.agent.grad_bucket()
        base = Stepper(job, 0, 1, "none").timed(args.steps)
        line = {"workload": f"{name} B={B}/GPU", "arena_MB": round(4 * n / 1e6, 2),
                "early_bucket_MB": round(4 * (n - b) / 1e6, 2), "none_ms": round(base, 4)}
        for tag, W, P, hold, lds in (("copy6_W32", 32, 6, 0, 0), ("hold100us_W16", 16, 1, 100, 0), ("hold100us_W32", 32, 1, 100, 0),
                                     ("hold100us_W32_lds16k", 32, 1, 100, 16384), ("hold200us_W32", 32, 1, 200, 0), ("hold100us_W8", 8, 1, 100, 0)):
            alone = standalone_copy_ms(job, W, P, b, n, hold, lds) + standalone_copy_ms(job, W, P, 0, b, hold, lds)
            side = Stepper(job, W, P, "side", hold, lds).timed(args.steps)
            serial = Stepper(job, W, P, "serial", hold, lds).timed(args.steps)
            line[tag] = {"standin_alone_ms": round(alone, 4), "side_ms": round(side, 4), "serial_ms": round(serial, 4),
                         "cost_beside_ms": round(side - base, 4), "cost_serial_ms": round(serial - base, 4)}
        print(json.dumps(line), flush=True)
        del job
        torch.cuda.empty_cache()


def trace_run(args):
    job = make_job("config4", 128, 2)
    st = Stepper(job, 32, 1, "side", 100, 0)
    st.timed(20, warmup=5)


def summarize(d):
    files = sorted(glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)
    if not files:
        print("no kernel trace under", d)
        return 1
    rows = list(csv.DictReader(open(files[-1])))
    ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
    ks.sort()
    side = [k for k in ks if "k_side_copy" in k[2]]
    side = side[len(side) // 2:]         # the steady-state half
    tot = ov = 0
    beside = {}
    for s0, s1, _ in side:
        tot += s1 - s0
        for k0, k1, nm in ks:
            if "k_side_copy" in nm or k1 <= s0 or k0 >= s1:
                continue
            o = min(s1, k1) - max(s0, k0)
            ov += o
            short = nm.replace("(anonymous namespace)::", "").split("(")[0][:70]
            beside[short] = beside.get(short, 0) + o
    print(f"{len(side)} stand-in launches, mean duration {tot / max(len(side), 1) / 1e3:.1f} us; "
          f"{100.0 * ov / max(tot, 1):.1f} % of their time a kernel of the main stream was running as well")
    for nm, o in sorted(beside.items(), key=lambda kv: -kv[1])[:12]:
        print(f"  {100.0 * o / max(tot, 1):5.1f} %  {nm}")
    return 0


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--passes", type=int, default=6, help="copy passes of the stand-in (6 passes of a 3.4 MB bucket on 32 workgroups ~ 0.1 ms)")
    ap.add_argument("--trace-run", action="store_true")
    ap.add_argument("--summarize", default=None)
    a = ap.parse_args()
    if a.summarize:
        sys.exit(summarize(a.summarize))
    if a.trace_run:
        trace_run(a)
    else:
        table(a)
