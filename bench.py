#!/usr/bin/env python3
"""Headline benchmark: gradient-steps/sec of the SAC/TQC hot path (BASELINE.json).

One "step" = one full train_step of the reference (franQ/Agent/deepQlearning.py:105-127) on one shard: windowed
minibatch sample from the HBM ring + loss + backward + Adam + polyak.  Synthetic data resident in HBM, random-init
weights; nothing here reads /root/reference.

N = 1 (the driver's BENCH line): BASELINE config 2 - obs 17, act 6, TQC 5 critics x 2 quantiles, MLPs of 256, ring
  1,000,000, B = 256 windows of temporal_len T = 50 (the reference default, conf.py:38).  `value` = steps/s over
  exactly --steps steps; extra keys: `sustained` (>= 2 s of back-to-back steps), `roofline` (dominant kernel, HIP
  events on the launch stream), `cpu_baseline` (the CPU oracle timed on this host), `sampler_roofline`,
  `other_configs` (configs 3, 4 at B=1024 on one GPU, 5), `facade_path` (the franQ-shaped objects end to end).
N > 1 (one process per GPU - started by torch.distributed.run, or by this script itself when WORLD_SIZE is
  not set: the parent touches no GPU, starts N fresh `python bench.py` children with RANK / LOCAL_RANK / WORLD_SIZE /
  MASTER_* set, relays rank 0's JSON line and exits non-zero if any child does).  Every rank keeps a full replica and its own
  ring shard and runs its backward in two phases: FDQL_PHASE_GRAD_CRITICS leaves the critics' share of the gradient arena
  final, its RCCL all-reduce starts on a side stream while FDQL_PHASE_GRAD_REST (encoder / joiner / actor gradients) runs, a
  second all-reduce takes the rest (the row weights already carry 1/(B_global), so the sums are the global-batch gradient),
  then FDQL_PHASE_APPLY: identical Adam on every rank.  No barrier inside a timed region.  Three workloads per line:
  * `value` - the metric's literal reading, "gradient-steps/sec (1M buffer, batch=256) at 1/2/4/8": BASELINE config 2 with the
    GLOBAL batch fixed at 256 windows, 256/N per GPU, a 1M-slot ring shard per rank ("scaling": "strong"; the N = 1 point IS
    the single-GPU line).  `value` = optimiser iterations per second: one all-reduced Adam update is ONE gradient step.
  * `config2_weak` - 256 windows PER GPU (global batch 256 N): `optimizer_iterations_per_s` and `minibatches_per_s` (= N x it).
  * `config4_strong` - SURVEY 8(d)'s multi-GPU workload: BASELINE config 4 (obs 376, act 17, TQC 5 x 25, 2M-slot ring shard
    per rank), GLOBAL batch 1024 windows split 1024/N, with `same_workload_1gpu` (rank 0 alone on the whole batch, measured
    in the same process) so that the strong-scaling speed-up is computable from the line itself.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python bench.py --gpus N --steps K --warmup W            # starts its own N rank processes
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import glob
import hashlib
import json
import os
import re
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HID = 256
MFMA_F32_PEAK_TFLOPS = 157.3                   # MI355X_MICROARCH.md: dense fp32 matrix peak (2.4 GHz)
HBM_PEAK_GBS = 8000.0

# name -> workload.  dims/keys: ring layout; rowbytes follow from it.
WORKLOADS = {
    "config1": dict(obs=3, act=1, goal=0, C=2, Q=1, distributional=False, ring=50_000, B=256, T=50, ep_len=200,
                    text="BASELINE config 1: Pendulum dims (obs=3 act=1), SAC-min 2 critics, MLP(256,256), 50k-transition HBM ring"),
    "config2": dict(obs=17, act=6, goal=0, C=5, Q=2, ring=1_000_000, B=256, T=50, ep_len=1000,
                    text="BASELINE config 2: TQC 5x2 quantile critics, obs=17 act=6, MLP(256), 1M-transition HBM ring"),
    "config3": dict(obs=28, act=6, goal=10, C=5, Q=2, ring=1_000_000, B=256, T=50, ep_len=50,
                    text="BASELINE config 3: config 2 + goal-conditioned rows (obs 28, achieved/desired goal 10), HER-relabelled "
                         "episodes of 50 in a 1M ring"),
    "config4": dict(obs=376, act=17, goal=0, C=5, Q=25, ring=2_000_000, B=1024, T=50, ep_len=1000,
                    text="BASELINE config 4: TQC 5x25 quantile critics, obs=376 act=17 (Humanoid dims), MLP(256), 2M-slot ring"),
}


def csrc_hash():
    """sha256 over the HIP sources: profiles collected on another revision are not quoted as this one's."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "fastdeepqlearning_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def layout(w):
    keys = ["obs_1d"] + (["achieved_goal", "desired_goal"] if w["goal"] else []) + \
           ["action", "reward", "mc_return", "task_done", "episode_done", "episode_step", "idx"]
    dims = [w["obs"]] + ([w["goal"], w["goal"]] if w["goal"] else []) + [w["act"], 1, 1, 1, 1, 1, 1]
    return keys, dims


def synth_rows(w, n, seed, device, gamma=0.99):
    """Synthetic transitions: obs~N(0,1), goals~U(-1,1), action~U(-1,1), reward~N(0,1), fixed-length episodes,
    task_done~Bernoulli(1e-3), mc_return = discounted reward-to-go (what NStepReturn attaches at write time)."""
    g = torch.Generator(device=device).manual_seed(seed)
    ep_len = w["ep_len"]
    n_ep = (n + ep_len - 1) // ep_len
    tot = n_ep * ep_len
    cols = [torch.randn(tot, w["obs"], generator=g, device=device)]
    if w["goal"]:
        cols.append(torch.rand(tot, w["goal"], generator=g, device=device) * 2 - 1)
        cols.append((torch.rand(n_ep, 1, w["goal"], generator=g, device=device) * 2 - 1).expand(n_ep, ep_len, w["goal"]).reshape(tot, -1))
    cols.append(torch.rand(tot, w["act"], generator=g, device=device) * 2 - 1)
    rew = torch.randn(n_ep, ep_len, generator=g, device=device)
    ret = torch.empty_like(rew)
    acc = torch.zeros(n_ep, device=device)
    for t in range(ep_len - 1, -1, -1):
        acc = rew[:, t] + gamma * acc
        ret[:, t] = acc
    td = (torch.rand(tot, 1, generator=g, device=device) < 1e-3).float()
    step = torch.arange(ep_len, device=device, dtype=torch.float32).repeat(n_ep).view(-1, 1)
    cols += [rew.reshape(-1, 1), ret.reshape(-1, 1), td, (step == ep_len - 1).float(), step, torch.zeros(tot, 1, device=device)]
    return torch.cat(cols, dim=1)[:n].contiguous()


class Job:
    """Ring + agent + persistent sample buffers of one workload on one device."""

    def __init__(self, w, dev, B, T, world=1, rank=0, ring_slots=None, fill_chunk=250_000):
        from fastdeepqlearning_amd.core import NativeAgent, NativeRing, make_config
        self.w, self.dev, self.B, self.T = w, dev, B, T
        self.keys, self.dims = layout(w)
        slots = ring_slots or w["ring"]
        cfg = make_config(w["obs"], w["act"], T, B, goal_dim=w["goal"], n_critics=w["C"], n_quantiles=w["Q"], latent=HID,
                          enc_features=HID, enc_hidden=(HID,), joint_hidden=(HID,), pi_hidden=(HID,), critic_hidden=(HID, HID),
                          distributional=w.get("distributional", True), world_size=world, keep_frozen_copy=True)
        self.agent = NativeAgent(cfg, dev)
        self.agent.init_weights(seed=0)                        # same weights on every rank
        # the ring is filled last: the step that follows finds the device as a running job would (busy, not idling
        # behind host-side set-up)
        self.ring = NativeRing(slots, self.dims, dev)
        fill_chunk = min(fill_chunk, max(slots // 2, 1))       # (one add stays well inside the ring)
        for c0 in range(0, slots + 1000, fill_chunk):          # wraps once: len = slots - 1 (quirk q1)
            n = min(fill_chunk, slots + 1000 - c0)
            self.ring.add_rows(synth_rows(w, n, 1000 * rank + c0 // fill_chunk, dev))
        assert len(self.ring) == slots - 1
        # three persistent sample buffers: the batch in use, the one being gathered ahead, the one the previous update read
        # (what Replay.make()'s shards give the facade: sample_buffers=3; the agent caches one launch plan per buffer set)
        self.pool = [[torch.empty(T, B, d, device=dev) for d in self.dims] for _ in range(3)]
        self.xps = [dict(zip(self.keys, outs)) for outs in self.pool]
        self.outs, self.xp = self.pool[0], self.xps[0]
        self.seed = 1234 + rank
        self.rowbytes = 4 * sum(self.dims)
        # FDQL_BENCH_PREFETCH=1: one batch ahead, like the reference's loader thread (torch_dataloader.py:40-50) - step i's
        # update runs beside the gather of step i + 1 on a side stream.  Measured SLOWER on MI355X (797-803 against 810 steps/s,
        # round 4: the gather's workgroups delay the persistent one-per-CU launches by more than the 12 us they hide), so the
        # default is gather and update back to back on one stream.
        self.prefetch = os.environ.get("FDQL_BENCH_PREFETCH") == "1"
        self.side = torch.cuda.Stream(dev) if self.prefetch else None
        self.ev_main = torch.cuda.Event()
        self.ev_done = [torch.cuda.Event() for _ in range(3)]
        self.pending_for = None

    def sample(self, i):
        """The batch of step i (index into pool / xps); with prefetch also issues the gather of step i + 1."""
        k = i % 3
        if not self.prefetch:
            self.ring.sample_windows(self.T, self.B, seed=self.seed, counter=i, outs=self.pool[k])
            return k
        main = torch.cuda.current_stream(self.dev)
        if self.pending_for == i:
            main.wait_event(self.ev_done[k])
        else:                                   # first step of a run: drawn in place
            self.ring.sample_windows(self.T, self.B, seed=self.seed, counter=i, outs=self.pool[k])
        k1 = (i + 1) % 3
        self.ev_main.record(main)               # the update that last read pool[k1] (step i - 2) is behind this point
        self.side.wait_event(self.ev_main)
        with torch.cuda.stream(self.side):
            self.ring.sample_windows(self.T, self.B, seed=self.seed, counter=i + 1, outs=self.pool[k1])
            self.ev_done[k1].record(self.side)
        self.pending_for = i + 1
        return k

    def step(self, i):
        k = self.sample(i)
        self.agent.update(self.xps[k], seed=self.seed)

    def timed(self, steps, warmup, first=0):
        for i in range(warmup):
            self.step(first + i)
        torch.cuda.synchronize(self.dev)
        t0 = time.perf_counter()
        for i in range(steps):
            self.step(first + warmup + i)
        torch.cuda.synchronize(self.dev)
        return time.perf_counter() - t0

    def timed_with_events(self, steps, warmup, first=0):
        """timed() plus the longest single step of the window (HIP events between the steps; recording one costs the host
        about a microsecond and the device nothing)."""
        for i in range(warmup):
            self.step(first + i)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        torch.cuda.synchronize(self.dev)
        t0 = time.perf_counter()
        evs[0].record()
        for i in range(steps):
            self.step(first + warmup + i)
            evs[i + 1].record()
        self.last_host_issue_ms = 1e3 * (time.perf_counter() - t0) / max(steps, 1)   # the loop without the synchronize
        torch.cuda.synchronize(self.dev)
        el = time.perf_counter() - t0
        self.last_device_ms = evs[0].elapsed_time(evs[steps]) / max(steps, 1)
        return el, max(evs[i].elapsed_time(evs[i + 1]) for i in range(steps)) if steps else 0.0

    def timed_windows(self, steps, warmup, windows=3):
        """`windows` back-to-back timed windows of `steps` steps after `warmup`: per-window steps/s (wall clock around a
        synchronised window) and the longest single step of all windows (HIP events between steps): one stalled step
        in a window shows up as max_step_ms instead of silently halving the figure."""
        for i in range(warmup):
            self.step(i)
        torch.cuda.synchronize(self.dev)
        rates, max_ms, first = [], 0.0, warmup
        for _ in range(windows):
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
            t0 = time.perf_counter()
            evs[0].record()
            for i in range(steps):
                self.step(first + i)
                evs[i + 1].record()
            torch.cuda.synchronize(self.dev)
            rates.append(steps / (time.perf_counter() - t0))
            max_ms = max(max_ms, max(evs[i].elapsed_time(evs[i + 1]) for i in range(steps)))
            first += steps
        rates.sort()
        return rates[len(rates) // 2], rates, max_ms

    def timed_split(self, steps, warmup, windows=3):
        """timed_windows() that also says WHERE the time goes: per window the wall time of the step loop WITHOUT the trailing
        synchronize (`host_issue_ms` per step: how long the host needs to enqueue one step - an upper bound, the runtime may
        block the host when its queue is full) and the device's own first-to-last time (`device_ms` per step, HIP events on the
        launch stream around the window).  A launch-bound step (17-23 dependent kernels) that is slow on one box shows here
        whether the host or the chip was slow.  Returns the median window's figures + every window's rate + max_step_ms."""
        for i in range(warmup):
            self.step(i)
        torch.cuda.synchronize(self.dev)
        rows, max_ms, first = [], 0.0, warmup
        for _ in range(windows):
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
            t0 = time.perf_counter()
            evs[0].record()
            for i in range(steps):
                self.step(first + i)
                evs[i + 1].record()
            t_issue = time.perf_counter() - t0
            torch.cuda.synchronize(self.dev)
            t_all = time.perf_counter() - t0
            rows.append((steps / t_all, 1e3 * t_issue / steps, evs[0].elapsed_time(evs[steps]) / steps))
            max_ms = max(max_ms, max(evs[i].elapsed_time(evs[i + 1]) for i in range(steps)))
            first += steps
        med = sorted(rows)[len(rows) // 2]
        return {"rate": med[0], "host_issue_ms": round(med[1], 4), "device_ms": round(med[2], 4),
                "windows": [round(r[0], 2) for r in rows], "max_step_ms": round(max_ms, 4)}

    def calibrate_launch_mode(self, max_rows=6272, steps=30):
        """Eager launches or hipGraph replay for this plan, by measurement on this host (plans with <= max_rows TD rows: the
        launch-bound regime); bigger plans stay eager (kernel-bound: replay measured 0.5-1.5 % slower)."""
        if (self.T - 1) * self.B > max_rows or self.prefetch:
            return {"chosen": "eager", "calibration": None}
        self._ci = getattr(self, "_ci", 900_000)

        def one():
            self.step(self._ci)
            self._ci += 1
        try:
            return self.agent.calibrate_launch_mode(one, steps=steps)
        except Exception as e:   # noqa: BLE001 - a failed capture must never take a figure of the line down: eager launches always work
            self.agent.set_launch_mode(False)
            return {"chosen": "eager", "calibration": None, "error": f"{type(e).__name__}: {e}"[:200]}

    def kernel_profile(self, reps=3):
        """name -> (ms, flops, bytes, launches) per step, HIP events on the launch stream."""
        acc = {}
        for _ in range(reps):
            for name, ms, fl, by in self.agent.profile_update(self.xp, seed=self.seed):
                a = acc.setdefault(name, [0.0, fl, by, 0])
                a[0] += ms
                a[3] += 1
        return {k: (v[0] / reps, v[1], v[2], v[3] // reps) for k, v in acc.items()}


def dominant(prof):
    """The kernel instantiation with the largest total time among the MFMA kernels (tile shapes of the grouped GEMM,
    the row-block chain kernel): name, ms/step, flops/step, launches/step, and the all-MFMA-kernel totals."""
    groups = {}
    for k, v in prof.items():
        if k.startswith(("gemm", "chain", "fwd3", "rows", "rowd", "wstat", "wgstat", "conv:")):
            groups.setdefault(k.split(":")[0], []).append(v)
    name, rows = max(groups.items(), key=lambda kv: sum(r[0] for r in kv[1]))
    ms, fl, n = sum(r[0] for r in rows), sum(r[1] * r[3] for r in rows), sum(r[3] for r in rows)
    all_ms = sum(r[0] for g in groups.values() for r in g)
    all_fl = sum(r[1] * r[3] for g in groups.values() for r in g)
    total_ms = sum(v[0] for v in prof.values())
    return name, ms, fl, n, all_ms, all_fl, total_ms, sum(r[2] * r[3] for r in rows)


def kernel_label(name):
    if name.startswith("conv:"):
        return "k_conv_* (implicit-GEMM convolution, image groups resident in LDS, fp32 v_mfma_f32_16x16x4_f32)"
    if name.startswith("chain"):
        return "k_chain (row-block MLP chain, fp32 v_mfma_f32_32x32x2_f32)"
    if name.startswith("fwd3"):
        return "k_fwd3 (small-block forward chain: encoder -> joiner -> actors on 16- / 32-row blocks, fp32 v_mfma_f32_16x16x4_f32)"
    if name.startswith("wstat"):
        return rocprof_tag(name) + "> (weight-stationary persistent row-block GEMM, " + ("dgrad" if name.startswith("wstatg") else "forward") + \
               " form, fp32 v_mfma_f32_32x32x2_f32)"
    if name.startswith("wgstat"):
        return "k_wgrad_stat (output-stationary weight-gradient blocks, fp32 v_mfma_f32_32x32x2_f32)"
    if name.startswith("rowdot"):
        return "k_rowdot (narrow-output dgrads, fp32 v_mfma_f32_16x16x4_f32)"
    if name.startswith("rowdchain"):
        return "k_rowdgrad_chain (three row-block dgrads of the same rows, activations resident in LDS, fp32 v_mfma_f32_16x16x4_f32)"
    if name.startswith("rowd"):
        return "k_rowdgrad (row-block dgrad, K-strided weights, fp32 v_mfma_f32_16x16x4_f32)"
    if name.startswith("rows"):
        return "k_rowgemm (persistent row-block GEMM, " + ("dgrad" if "KS" in name else "forward") + " form, fp32 v_mfma_f32_32x32x2_f32)"
    return f"k_gemm_grouped<{name[4:]} tile> (fp32 v_mfma_f32_32x32x2_f32)"


def rocprof_tag(name):
    """Substring of the rocprofv3 kernel name behind a profile entry (profiles/*: which kernel a PMC file is about)."""
    if name.startswith("conv:"):
        return "k_conv_"
    if name.startswith("chain"):
        return "k_chain"
    if name.startswith("fwd3"):
        return "k_fwd3"
    if name.startswith("wgstat"):
        return "k_wgrad_stat"
    if name.startswith("rowdot<"):
        return "k_rowdot"
    if name.startswith("rowdchain"):
        return "k_rowdgrad_chain"
    if name.startswith("rowd<"):
        return "k_rowdgrad"
    if name.startswith("wstatg<"):       # "wstatg<fuse,plain,ns>:stage" -> k_wstat_grad<true, false, 1>
        f, pl, ns = name[7:name.index(">")].split(",")
        return f"k_wstat_grad<{'true' if f == '1' else 'false'}, {'true' if pl == '1' else 'false'}, {ns}"   # (+ ", MASK>": prefix match)
    if name.startswith("wstat<"):        # "wstat<nsl,nst,hfq>:stage" -> k_wstat<0, 0, 2>
        n, t, q = name[6:name.index(">")].split(",")
        return f"k_wstat<{n}, {t}, {q}"   # (+ ", GM>" since round 4: the profile entries are matched by this prefix)
    if name.startswith("rows"):
        return "k_rowgemm"
    shape_id = {"128x128": 0, "128x32": 1, "32x128": 2, "64x128": 3, "64x64dual": 4, "64x64": 5, "64x64hf": 6}
    return f"k_gemm_grouped<{shape_id.get(name[4:], 5)},"


def roofline_of(job, committed_pmc=True):
    prof = job.kernel_profile()
    name, ms, fl, n, all_ms, all_fl, total_ms, by = dominant(prof)
    tf = fl / max(ms * 1e-3, 1e-12) / 1e12
    r = {"bound": "mfma", "kernel": kernel_label(name), "achieved": round(tf, 2), "peak": MFMA_F32_PEAK_TFLOPS,
         "unit": "TFLOP/s", "frac": round(tf / MFMA_F32_PEAK_TFLOPS, 4), "traffic": None,
         "launches_per_step": int(n), "flops_per_launch": fl / max(n, 1), "avg_launch_ms": round(ms / max(n, 1), 4),
         "share_of_step": round(ms / total_ms, 3),
         "all_mfma_kernels": {"flops_per_step": all_fl, "ms_per_step": round(all_ms, 4),
                              "achieved": round(all_fl / (all_ms * 1e-3) / 1e12, 2), "share_of_step": round(all_ms / total_ms, 3)}}
    top = {k: round(v[0], 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])[:12]}
    if not committed_pmc:
        return r, top
    # PMC figures (HBM traffic, MFMA pipe utilisation, clock) cannot be measured from inside this process: they come from
    # the committed rocprofv3 --pmc passes of the SAME workload, and only if those were collected on THIS csrc revision
    tj = os.path.join(ROOT, "profiles", "r06_dominant_kernel_traffic.json")
    if os.path.exists(tj):
        tr = json.load(open(tj))
        ent = None
        if tr.get("csrc_sha") == csrc_hash():
            if rocprof_tag(name) in tr.get("kernel", ""):
                ent = tr
            else:   # (two launches of a step are equally long: which one is "dominant" changes from run to run)
                hit = [v for k, v in tr.get("kernels", {}).items() if rocprof_tag(name) in k]
                ent = dict(hit[0], source=tr["source"]) if len(hit) == 1 else None
        if ent is not None:
            tr = ent
            r["traffic"] = tr["hbm_bytes_per_launch"]
            r["traffic_unit"] = "bytes/launch (L2<->fabric read+write, PMC FETCH_SIZE x2 + WRITE_SIZE)"
            r["traffic_source"] = "profiles/r06_hbm_traffic_pmc.txt (" + tr["source"] + ")"
            r["algorithmic_bytes_per_launch"] = by / max(n, 1)
            for k in ("mfma_pipe_busy_frac", "shader_clock_ghz_under_load"):
                if k in tr:
                    r[k] = tr[k]
        else:
            r["stale_profile"] = True      # profiles/ were collected on another revision of csrc/: not quoted
    return r, top


def sampler_roofline(job, reps=50):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(job.dev)
    e0.record()
    for i in range(reps):
        job.ring.sample_windows(job.T, job.B, seed=job.seed, counter=10_000 + i, outs=job.outs)
    e1.record()
    torch.cuda.synchronize(job.dev)
    ms = e0.elapsed_time(e1) / reps
    nbytes = 2.0 * job.T * job.B * job.rowbytes + 8 * job.B
    return {"bound": "hbm", "kernel": "k_gather_windows (window starts drawn in-kernel)", "ms": round(ms, 4),
            "achieved": round(nbytes / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "algorithmic_bytes": nbytes}


def cpu_baseline(w, T, B, seconds_budget=25.0, max_steps=200, thread_sets=None):
    """The CPU oracle (oracle/: numpy ring + eager-torch update, a port of the reference path) timed on this host:
    sample + update, steady state, bounded sample of the same workload."""
    from oracle import update as oup
    from oracle.replay import RingOracle, cast_like_loader
    keys, dims = layout(w)
    threads = torch.get_num_threads()
    spec = oup.Spec(obs=w["obs"], act=w["act"], C=w["C"], Q=w["Q"], latent=HID, enc_features=HID, enc_hidden=(HID,),
                    joint_hidden=(HID,), pi_hidden=(HID,), critic_hidden=(HID, HID), distributional=w.get("distributional", True),
                    T=T, B=B)
    st = oup.new_state(spec, oup.init_params(spec, seed=0))
    ring = RingOracle(w["ring"], B, T)
    rows = synth_rows(w, min(200_000, w["ring"] - 1), 0, "cpu").numpy()   # a <= 200k-row slice of the ring is enough for timing
    off = np.cumsum([0] + dims)
    ring.memory = {k: np.zeros((w["ring"], d), np.float32) for k, d in zip(keys, dims)}
    for j, k in enumerate(keys):
        ring.memory[k][:rows.shape[0]] = rows[:, off[j]:off[j + 1]]
    ring.top, ring.len = rows.shape[0], rows.shape[0]
    rng = np.random.RandomState(0)
    g = torch.Generator().manual_seed(0)

    def sample_only():
        xp = cast_like_loader(ring.temporal_sample(rng=rng))
        return {k: torch.from_numpy(v) for k, v in xp.items()}

    def one():
        xp = sample_only()
        nt, na = torch.randn(T - 1, B, w["act"], generator=g), torch.randn(T - 1, B, w["act"], generator=g)
        oup.train_step(st, spec, xp, nt, na)

    def timed(nthreads, budget):
        torch.set_num_threads(nthreads)
        one()  # warm-up
        t0 = time.perf_counter()
        n = 0
        while True:
            one()
            n += 1
            el = time.perf_counter() - t0
            if el > budget or n >= max_steps:
                break
        return n / el, n, el

    # eager torch-CPU on small GEMMs does not scale to a whole socket: time all cores and 16 threads, report the faster
    # (the secondaries' short baselines run the 16-thread setting only: the faster one at config 2 on every box so far)
    sets = thread_sets or ([threads, 16] if threads > 16 else [threads])
    sets = sorted({min(t, threads) for t in sets}, reverse=True)
    results = [(timed(t, seconds_budget / len(sets)), t) for t in sets]
    torch.set_num_threads(threads)
    t0 = time.perf_counter()
    for _ in range(20):
        sample_only()
    sample_ms = (time.perf_counter() - t0) / 20 * 1e3
    (rate, n, el), used = max(results, key=lambda r: r[0][0])
    detail = "; ".join(f"{t} threads: {r[0]:.3f} steps/s" for r, t in results)
    return {"value": rate, "unit": "steps/s", "cores": used, "kind": "port",
            "sample_only_ms": round(sample_ms, 3), "update_only_ms": round(1e3 / rate - sample_ms, 3),
            "sample": f"{n} train_steps (numpy ring sample + torch-CPU update) of the same config, T={T}, B={B}, "
                      f"{el:.1f} s [{detail}]"}


def secondary(name, dev, steps=200, warmup=10, **kw):
    """steps/s of another BASELINE config on this one GPU + its dominant kernel's fraction (never `value`): the median of
    three windows of `steps` steps each, every window's figure and the longest single step listed beside it."""
    w = WORKLOADS[name]
    cpu_budget, cpu_steps = kw.pop("cpu_budget", 4.0), kw.pop("cpu_steps", 50)
    job = Job(w, dev, kw.pop("B", w["B"]), kw.pop("T", w["T"]), **kw)
    mode = job.calibrate_launch_mode()
    plans0 = job.agent.stats()["plans_built"]
    sp = job.timed_split(steps, warmup)
    med = sp["rate"]
    r, _ = roofline_of(job, committed_pmc=False)
    smp = sampler_roofline(job, 20)
    out = {"workload": w["text"] + f", B={job.B} x T={job.T}", "value": round(med, 2), "unit": "steps/s",
           "ms_per_step": round(1e3 / med, 4), "steps": steps, "windows": sp["windows"],
           "max_step_ms": sp["max_step_ms"], "host_issue_ms": sp["host_issue_ms"], "device_ms": sp["device_ms"],
           "launch_mode": mode, "plans_built_in_windows": job.agent.stats()["plans_built"] - plans0,
           "dominant_kernel": r["kernel"], "dominant_kernel_tflops": r["achieved"], "dominant_kernel_frac": r["frac"],
           "all_mfma_kernels_tflops": r["all_mfma_kernels"]["achieved"],
           "sampler_gbs": smp["achieved"], "sampler_roofline": smp}
    B, T = job.B, job.T
    del job
    torch.cuda.empty_cache()
    if cpu_budget:
        try:
            out["cpu_baseline"] = cpu_baseline(w, T, B, seconds_budget=cpu_budget, max_steps=cpu_steps, thread_sets=[16])
        except Exception as e:   # noqa: BLE001
            out["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"[:200]}
    return out


def temporal_len_2(w, dev, B, steps=300):
    """The plain 1-step-minibatch reading of batch=256 (temporal_len 2): 23 dependent launches of ~11 us.  Median of three
    windows, the longest single step, host issue time against device time, and the launch mode chosen by calibration."""
    job = Job(w, dev, B, 2, ring_slots=min(200_000, w["ring"]))
    mode = job.calibrate_launch_mode()
    sp = job.timed_split(steps, 30)
    out = {"temporal_len": 2, "value": round(sp["rate"], 1), "unit": "steps/s", "ms_per_step": round(1e3 / sp["rate"], 4),
           "steps": steps, "windows": sp["windows"], "max_step_ms": sp["max_step_ms"], "host_issue_ms": sp["host_issue_ms"],
           "device_ms": sp["device_ms"], "launch_mode": mode, "launches_per_step": job.agent.stats()["n_launches"] + 1,
           "transitions_per_step": 2 * B,
           "note": "the plain 1-step-minibatch reading of batch=256: launch/latency-bound, 2.6 GFLOP per step; host_issue_ms = wall time "
                   "of the step loop without the trailing synchronize, device_ms = HIP events first to last"}
    del job
    torch.cuda.empty_cache()
    try:
        cb = cpu_baseline(w, 2, B, seconds_budget=3.0, max_steps=100, thread_sets=[16])
        out["cpu_baseline"] = cb
    except Exception as e:   # noqa: BLE001
        out["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"[:200]}
    return out


def config1_pendulum(dev):
    """BASELINE config 1 (the reference's own CPU-runnable case: franQ/Agent/components/soft_actor_critic.py:63-134 on
    experiments/train/pendulum.py's dims): SAC-min, 2 critics, 50k ring, B=256; T=50 (reference default) and T=2."""
    w = WORKLOADS["config1"]
    out = secondary("config1", dev, steps=200, warmup=10, cpu_budget=3.0)
    out["temporal_len_2"] = temporal_len_2(w, dev, w["B"], steps=200)
    return out


def config4_secondary(dev):
    """Config 4 at the full batch on one GPU; its gather is the one bandwidth-bound sampler launch of the BASELINE configs, so
    it carries a sampler_roofline with the PMC traffic of the committed rocprofv3 passes (quoted only on the same csrc revision);
    the CPU baseline is a few oracle steps (one step is ~1.5 s of CPU work)."""
    out = secondary("config4", dev, steps=100, warmup=5, cpu_budget=5.0, cpu_steps=3)
    tj = os.path.join(ROOT, "profiles", "r06_sampler_traffic.json")
    smp = out.get("sampler_roofline")
    if smp and os.path.exists(tj):
        tr = json.load(open(tj))
        if tr.get("csrc_sha") == csrc_hash() and "config4" in tr:
            smp["traffic"] = tr["config4"]["hbm_bytes_per_launch"]
            smp["traffic_source"] = "profiles/r06_sampler_pmc.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"
        else:
            smp["stale_profile"] = True
    return out


def config4_per_rank(dev, full_ms):
    """What ONE rank of the N-GPU job computes per step (config 4, global B = 1024 split B/N), measured on this one GPU
    without the collective: a compute-only bound on the strong-scaling curve (speed-up <= t(B) / t(B/N))."""
    w = WORKLOADS["config4"]
    out = {"what": "config 4 per-rank batch on one GPU, no all-reduce: ms/step and the compute-only bound on the N-GPU speed-up",
           "B1024_ms": round(full_ms, 4)}
    for n in (2, 4, 8):
        job = Job(w, dev, w["B"] // n, w["T"], ring_slots=200_000)
        mode = job.calibrate_launch_mode()
        sp = job.timed_split(100, 10)
        out[f"N{n}_B{w['B'] // n}_ms"] = round(1e3 / sp["rate"], 4)
        out[f"N{n}_speedup_bound"] = round(full_ms / (1e3 / sp["rate"]), 2)
        out[f"N{n}_detail"] = {"windows_steps_per_s": sp["windows"], "max_step_ms": sp["max_step_ms"], "host_issue_ms": sp["host_issue_ms"],
                               "device_ms": sp["device_ms"], "launch_mode": mode}
        del job
        torch.cuda.empty_cache()
    return out


def config2_per_rank(dev, full_ms, T):
    """What ONE rank of the N-GPU job computes per step at the metric's literal reading (config 2, GLOBAL batch 256 windows
    split 256/N), measured on this one GPU without the collective: the compute-only bound on `value`'s strong-scaling curve.
    `two_bucket` = the same batch on the data-parallel plan's launch list (FDQL_FORCE_BUCKETS: what a rank really runs)."""
    w = WORKLOADS["config2"]
    out = {"what": "config 2 per-rank batch on one GPU, no all-reduce: ms/step and the compute-only bound on the N-GPU speed-up "
                   "(global batch 256 windows)", "B256_ms": round(full_ms, 4)}
    for n in (2, 4, 8):
        job = Job(w, dev, w["B"] // n, T, ring_slots=200_000)
        mode = job.calibrate_launch_mode()
        sp = job.timed_split(200, 20)
        out[f"N{n}_B{w['B'] // n}_ms"] = round(1e3 / sp["rate"], 4)
        out[f"N{n}_speedup_bound"] = round(full_ms / (1e3 / sp["rate"]), 2)
        out[f"N{n}_detail"] = {"windows_steps_per_s": sp["windows"], "max_step_ms": sp["max_step_ms"], "host_issue_ms": sp["host_issue_ms"],
                               "device_ms": sp["device_ms"], "launch_mode": mode}
        del job
        torch.cuda.empty_cache()
    return out


def config3_her_ingest(dev, episodes=400):
    """Config 3's write path: episodes of 50 through fdql_ring_append_episode with the hindsight copy and the n-step
    return computed on the device (her.py:55-95, nstep_return.py:36-72); records/s including the host packing."""
    from fastdeepqlearning_amd.Replay import ReplayMemory
    from fastdeepqlearning_amd.Replay.wrappers import SparseL2Reward
    w = WORKLOADS["config3"]
    shard = ReplayMemory(200_000, 256, 50, device=dev)
    rng = np.random.RandomState(0)
    fn = SparseL2Reward(0.05, -1.0)
    eps = []
    for e in range(episodes):
        dg = rng.uniform(-1, 1, w["goal"]).astype(np.float32)
        eps.append([{"obs_1d": rng.standard_normal(w["obs"]).astype(np.float32),
                     "achieved_goal": rng.uniform(-1, 1, w["goal"]).astype(np.float32), "desired_goal": dg,
                     "action": rng.uniform(-1, 1, w["act"]).astype(np.float32), "reward": -1.0, "task_done": False,
                     "episode_done": i == 49, "episode_step": i} for i in range(50)])
    shard.append_episode(eps[0], return_name="mc_return", n_step=1000, discount=0.99, her=(49, fn))
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    n = 0
    for ep in eps[1:]:
        n += shard.append_episode(ep, return_name="mc_return", n_step=1000, discount=0.99, her=(49, fn))
    torch.cuda.synchronize(dev)
    return round(n / (time.perf_counter() - t0), 0)


def config3_her_vmap(dev, episodes=8000, K=32, steps=100):
    """Config 3 with sample-time relabelling (SURVEY 8d "per-window sample-time relabel", franQ/Replay/wrappers/her_vmap.py):
    K = 32 virtual goals + the real one stored per step of a 1M-slot ring through the facade's write stack
    (HindsightVmapWrite over NStepReturnVmap: relabel and per-column returns on the device, one append per episode),
    HindsightVmapRead's column select inside the gather kernel, DeepQLearning.train_step() on that read head."""
    import random
    from fastdeepqlearning_amd import Agent, Replay
    from fastdeepqlearning_amd.Agent import AgentConf
    from fastdeepqlearning_amd.Replay.wrappers import SparseL2Reward
    w = WORKLOADS["config3"]
    conf = AgentConf()
    conf.obs_space = _Space(spaces={"obs_1d": _Space(shape=(w["obs"],)), "achieved_goal": _Space(shape=(w["goal"],)),
                                    "desired_goal": _Space(shape=(w["goal"],))})
    conf.action_space = _Space(shape=(w["act"],))
    conf.discrete = False
    conf.training_device = conf.inference_device = dev
    conf.batch_size, conf.temporal_len, conf.replay_size = w["B"], w["T"], w["ring"]
    conf.num_critics, conf.num_q_predictions = w["C"], w["Q"]
    conf.use_async_train, conf.num_instances = False, 1
    conf.use_HER, conf.her_mode, conf.use_nStep_lowerbounds = True, "vmap", True
    read_heads, write_heads = Replay.make(conf, compute_reward=SparseL2Reward(0.05, -1.0))
    wh = write_heads[0]
    wh.num_virtual_goals = K
    rng = np.random.RandomState(0)
    np.random.seed(0)
    random.seed(0)
    ep_len = w["ep_len"]

    def episode():
        dg = rng.uniform(-1, 1, w["goal"]).astype(np.float32)
        obs = rng.standard_normal((ep_len, w["obs"])).astype(np.float32)
        ag = rng.uniform(-1, 1, (ep_len, w["goal"])).astype(np.float32)
        act = rng.uniform(-1, 1, (ep_len, w["act"])).astype(np.float32)
        return [{"obs_1d": obs[i], "achieved_goal": ag[i], "desired_goal": dg, "action": act[i], "reward": -1.0, "task_done": False,
                 "episode_done": i == ep_len - 1, "episode_step": i} for i in range(ep_len)]

    eps = [episode() for _ in range(200)]          # a pool of host episodes, appended round-robin (the relabel differs per append)
    keys_ep = list(eps[0][0])
    stacked = [{k: np.stack([np.asarray(r[k]) for r in ep]) for k in keys_ep} for ep in eps]   # what a vectorised actor holds
    for rec in eps[0]:
        wh.add(rec)
    # three forms of the same write: per-record add() (the reference's only one), add_episode(list of records),
    # add_episode(stacked columns: no per-record Python); the ring is filled by the last
    n_a = max(episodes // 40, 20)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for e in range(1, 1 + n_a):
        for rec in eps[e % len(eps)]:
            wh.add(rec)
    torch.cuda.synchronize(dev)
    ingest_per_record = n_a * ep_len / (time.perf_counter() - t0)
    t0 = time.perf_counter()
    for e in range(n_a):
        wh.add_episode(eps[e % len(eps)])
    torch.cuda.synchronize(dev)
    ingest_list = n_a * ep_len / (time.perf_counter() - t0)
    t0 = time.perf_counter()
    n_rec = 0
    for e in range(1 + 2 * n_a, episodes):
        wh.add_episode(stacked[e % len(stacked)])
        n_rec += ep_len
    torch.cuda.synchronize(dev)
    ingest = n_rec / (time.perf_counter() - t0)
    rh = read_heads[0]
    # sampler alone: T x B windows with the virtual column selected inside the gather
    for _ in range(5):
        xp = rh.temporal_sample()
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        xp = rh.temporal_sample()
    e1.record()
    torch.cuda.synchronize(dev)
    ms = e0.elapsed_time(e1) / 50
    row_floats = sum(int(np.prod(v.shape[2:])) for v in xp.values())
    nbytes = 2.0 * w["T"] * w["B"] * 4 * row_floats + 8 * w["B"]
    agent = Agent.make(conf)
    agent.enable_training(read_heads)
    for _ in range(15):
        agent.train_step()
    torch.cuda.synchronize(dev)
    rates = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(steps):
            agent.train_step()
        torch.cuda.synchronize(dev)
        rates.append(steps / (time.perf_counter() - t0))
    rates.sort()
    ring = rh.replay_buffer._ring if hasattr(rh, "replay_buffer") else None
    out = {"workload": f"BASELINE config 3 with sample-time relabelling: K={K} virtual goals + the real one per step "
                       f"({ring.row_floats if ring else '?'} floats = {4 * ring.row_floats if ring else '?'} B per slot) in a {w['ring']}-slot ring holding "
                       f"{len(rh)} records, HindsightVmapRead column select inside the gather, B={w['B']} x T={w['T']}",
           "value": round(rates[1], 2), "unit": "steps/s", "ms_per_step": round(1e3 / rates[1], 4), "steps": steps,
           "windows": [round(r, 2) for r in rates], "plans_built": agent.native.stats()["plans_built"],
           "sampler_ms": round(ms, 4), "sampler_gbs": round(nbytes / (ms * 1e-3) / 1e9, 1), "sampler_algorithmic_bytes": nbytes,
           "sampler_note": "algorithmic bytes = 2 x T x B x the SELECTED row (one goal column of 33); latency-bound at 6 MB like config 2's gather",
           "her_vmap_ingest_records_per_s": round(ingest, 0),
           "her_vmap_ingest_records_per_s_list_of_records": round(ingest_list, 0),
           "her_vmap_ingest_records_per_s_per_record_add": round(ingest_per_record, 0),
           "ingest_what": f"{n_rec // ep_len} episodes of {ep_len} through HindsightVmapWrite.add_episode(stacked columns) -> NStepReturnVmap -> ring "
                          f"(one call per episode: one H2D copy, relabel + per-column returns + append on the device; no per-record Python); "
                          f"beside it the same episodes as lists of records ({n_a} episodes) and through per-record add() ({n_a} episodes)"}
    del agent, read_heads, write_heads
    torch.cuda.empty_cache()
    return out


def config5_secondary(dev, ring=1_000_000, B=512, T=50, steps=20):
    """BASELINE config 5 (discrete SAC on 4x84x84 uint8 frame stacks, conv encoder of this build - no reference exists,
    SURVEY 8d) at the full batch on the 1M-frame uint8 ring BASELINE states (28 GB of HBM).  Per step: windowed sample of the
    scalar keys + the window slots, then the update - the first conv layer reads the uint8 frames of the sampled windows
    straight from the ring (no float32 frame batch, no column matrices: csrc/conv.hip)."""
    from fastdeepqlearning_amd.core import NativeAgent, NativeRing, make_config
    IMG, ACT = (4, 84, 84), 6
    dims = [IMG[0] * IMG[1] * IMG[2], 1, 1, 1, 1, 1]
    keys = ["obs_2d", "action", "reward", "mc_return", "task_done", "episode_step"]
    r = NativeRing(ring, dims, dev, dtypes=["u8", "f32", "f32", "f32", "f32", "f32"])
    g = torch.Generator(device=dev).manual_seed(0)
    done = 0
    while done < ring:
        n = min(8192, ring - done)
        rows = torch.empty(n, sum(dims), device=dev)
        rows[:, :dims[0]] = torch.randint(0, 256, (n, dims[0]), device=dev, generator=g).float()
        rows[:, dims[0]] = torch.randint(0, ACT, (n,), device=dev, generator=g).float()
        rows[:, dims[0] + 1:dims[0] + 3] = torch.randn(n, 2, device=dev, generator=g)
        rows[:, dims[0] + 3] = (torch.rand(n, device=dev, generator=g) < 0.001).float()
        rows[:, dims[0] + 4] = ((torch.arange(n, device=dev) + done) % 1000).float()
        r.add_rows(rows)
        done += n
    del rows
    cfg = make_config(0, ACT, T, B, discrete=True, n_critics=5, n_quantiles=2, img=IMG, conv=((32, 8, 4), (64, 4, 2), (64, 3, 1)),
                      obs_2d_u8=True)
    agent = NativeAgent(cfg, dev)
    agent.init_weights(0)
    in_place = agent.conv_reads_ring()
    outs = [None if (k == "obs_2d" and in_place) else torch.empty((T, B, d), device=dev) for k, d in zip(keys, dims)]
    starts = torch.empty(B, dtype=torch.int64, device=dev)
    slots = torch.empty((T, B), dtype=torch.int32, device=dev)
    xp = {k: o for k, o in zip(keys, outs) if o is not None}
    xp["obs_2d"], xp["obs_2d_slots"] = r.key_block_u8(0), slots

    def step(i):
        r.sample_windows(T, B, seed=7, counter=i, outs=outs, select={0: None}, starts_out=starts)
        r.window_slots(T, B, starts, out=slots)
        agent.update(xp, seed=7)

    for i in range(2):
        step(i)
    torch.cuda.synchronize(dev)
    dts = []
    for wnd in range(3):                      # median of three windows
        t0 = time.perf_counter()
        for i in range(steps):
            step(10 + wnd * steps + i)
        torch.cuda.synchronize(dev)
        dts.append((time.perf_counter() - t0) / steps)
    dt = sorted(dts)[1]
    fl = agent.stats()["gemm_flops"]
    prof = agent.profile_update(xp, seed=7)
    conv = [(n, ms, f) for n, ms, f, _ in prof if n.startswith("conv:")]
    out = {"workload": f"BASELINE config 5: discrete SAC (6 actions), 4x84x84 uint8 frame stacks, conv 32x8/4-64x4/2-64x3/1, "
                       f"B={B} x T={T}, {ring}-frame uint8 ring, frames read from the ring in place by the first conv layer "
                       f"(conv encoder: no reference exists, throughput only)",
           "value": round(1 / dt, 2), "unit": "steps/s", "ms_per_step": round(dt * 1e3, 3), "steps": steps,
           "windows": [round(1 / d, 2) for d in dts],
           "frames_per_s": round(T * B / dt, 0), "all_mfma_kernels_tflops_over_step": round(fl / dt / 1e12, 1),
           "workspace_GiB": round(agent.workspace.numel() / 2 ** 30, 1), "ring_GiB": round(ring * dims[0] / 2 ** 30, 1),
           "plans_built": agent.stats()["plans_built"],
           "conv_kernels": {n[5:]: {"ms": round(ms, 3), "tflops": round(f / ms / 1e9, 1) if ms else 0.0} for n, ms, f in conv}}
    del agent, r, outs, xp
    torch.cuda.empty_cache()
    return out


class _Space:
    def __init__(self, shape=None, spaces=None):
        if shape is not None:
            self.shape = tuple(shape)
        if spaces is not None:
            self.spaces = spaces


def facade_path(dev, steps=100):
    """Config 2 through the franQ-shaped objects a Runner would hold (Replay.make + Agent.make + train_step()): what the
    drop-in user gets, Python wrapper and per-step buffer bookkeeping included."""
    from fastdeepqlearning_amd import Agent, Replay
    from fastdeepqlearning_amd.Agent import AgentConf
    w = WORKLOADS["config2"]
    conf = AgentConf()
    conf.obs_space = _Space(spaces={"obs_1d": _Space(shape=(w["obs"],))})
    conf.action_space = _Space(shape=(w["act"],))
    conf.discrete = False
    conf.training_device = conf.inference_device = dev
    conf.batch_size, conf.temporal_len, conf.replay_size = w["B"], w["T"], 200_000
    conf.num_critics, conf.num_q_predictions = w["C"], w["Q"]
    conf.use_async_train, conf.num_instances = False, 1
    read_heads, _ = Replay.make(conf)
    keys, dims = layout(w)
    rows = synth_rows(w, 150_000, 5, dev)
    template = {k: (np.zeros(d, np.float32) if d > 1 else 0.0) for k, d in zip(keys, dims)}
    read_heads[0]._ensure_ring(template)
    read_heads[0].add_rows(rows)
    read_heads[0]._len = 150_000
    agent = Agent.make(conf)
    agent.enable_training(read_heads)
    for _ in range(10):
        agent.train_step()
    rates = []
    for _ in range(3):      # three windows, the median reported: the host does more per step here (one stalled window of a
        torch.cuda.synchronize(dev)   # busy box halved the single-window figure once, 516 against 850 steps/s)
        t0 = time.perf_counter()
        for _ in range(steps):
            agent.train_step()
        torch.cuda.synchronize(dev)
        rates.append(steps / (time.perf_counter() - t0))
    rates.sort()
    return {"value": round(rates[1], 2), "unit": "steps/s", "ms_per_step": round(1e3 / rates[1], 4), "windows": [round(r, 2) for r in rates],
            "plans_built": agent.native.stats()["plans_built"],
            "what": "DeepQLearning.train_step() on Replay.make()'s shard, config 2 dims, T=50, B=256, 200k-slot ring"}


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: this (parent) process never touches a GPU; it starts N FRESH children
    of the same command line with the rendezvous variables torch.distributed.run would set, relays rank 0's stdout (the JSON
    line) and returns non-zero if any rank fails (the others are then stopped: they would wait in a collective forever)."""
    backend = os.environ.get("FDQL_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()            # (counting devices does not initialise the GPU runtime)
    if backend == "nccl" and ndev < n:
        print(f"bench.py --gpus {n}: {ndev} GPU(s) visible; RCCL needs one per rank (FDQL_BENCH_BACKEND=gloo rehearses the N>1 "
              f"code path with ranks sharing a card)", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True if r == 0 else None))
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)   # drains rank 0's pipe
    reader.start()
    rc = 0
    try:
        while any(p.poll() is None for p in procs):
            bad = [p for p in procs if p.poll() not in (None, 0)]
            if bad:
                rc = bad[0].returncode or 1
                break
            time.sleep(0.2)
    finally:
        for p in procs:                       # exact children only, never a pattern
            if p.poll() is None and rc:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=30)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    reader.join(timeout=10)
    rc = rc or next((p.returncode for p in procs if p.returncode), 0)
    if chunks and chunks[0]:
        # stdout carries the ONE JSON line; anything else a rank's libraries wrote there (gloo announces its peers on stdout) goes
        # to stderr
        for line in chunks[0].splitlines(keepends=True):
            (sys.stdout if line.lstrip().startswith("{") else sys.stderr).write(line)
        sys.stdout.flush()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--temporal-len", type=int, default=50)
    ap.add_argument("--batch", type=int, default=0, help="windows per step (default: the workload's; N>1: the GLOBAL batch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip other_configs / facade_path / temporal_len 2 (profiling runs)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))              # parent: no GPU call before or after this
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: one process per GPU")
    import torch.distributed as dist
    # FDQL_BENCH_BACKEND=gloo: rehearsal of the N>1 code path on a box with fewer GPUs than ranks (ranks share devices,
    # the gradient all-reduce goes through the host); the driver uses nccl (= RCCL)
    backend = os.environ.get("FDQL_BENCH_BACKEND", "nccl")
    dev = torch.device(f"cuda:{local_rank % max(torch.cuda.device_count(), 1)}")
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    from fastdeepqlearning_amd import _native as nat
    T = args.temporal_len

    if world == 1:
        out = bench_single(args, dev, T)
    else:
        out = bench_distributed(args, dev, T, rank, world, backend, dist, nat)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


def bench_single(args, dev, T):
    w = WORKLOADS["config2"]
    B = args.batch or w["B"]
    job = Job(w, dev, B, T)
    # Steady state first ("steady state, excluding warm-up"): a fresh process rides the chip's clock ramp for its
    # first ~15 steps (profiles/r03_step_time_trend_after_start.txt), so >= 2 s of back-to-back steps run BEFORE the W warm-up
    # steps and the K timed ones - they are also the `sustained` figure
    probe = job.timed(20, 5, first=200_000)
    n_sus = max(200, int(2.2 / max(probe / 20, 1e-6)))
    el_sus = job.timed(n_sus, 0, first=100_000)
    el, max_step_ms = job.timed_with_events(args.steps, args.warmup)
    ms_per_step = 1e3 * el / max(args.steps, 1)
    host_issue_ms, device_ms = job.last_host_issue_ms, job.last_device_ms
    # two more windows of the same length right behind it: one stalled step in a short window shows up here
    more = [job.timed_with_events(args.steps, 0, first=300_000 + 1000 * k) for k in range(2)]
    roofline, top = roofline_of(job)
    sampler = sampler_roofline(job)
    extras, t2, facade = None, None, None
    if not args.no_extras:
        try:
            t2 = temporal_len_2(w, dev, B)
        except Exception as e:   # noqa: BLE001 - a secondary figure must never take the headline line down
            t2 = {"error": f"{type(e).__name__}: {e}"[:300]}
        facade = facade_path(dev)
        extras = {}
        for name, fn in (("config1_pendulum", lambda: config1_pendulum(dev)),
                         ("config3_her", lambda: dict(secondary("config3", dev), her_ingest_records_per_s=config3_her_ingest(dev))),
                         ("config3_her_vmap", lambda: config3_her_vmap(dev)),
                         ("config4_1gpu_B1024", lambda: config4_secondary(dev)),
                         ("config5_B512", lambda: config5_secondary(dev))):
            try:
                extras[name] = fn()
            except Exception as e:   # noqa: BLE001 - a secondary figure must never take the headline line down
                extras[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
            torch.cuda.empty_cache()
        try:
            extras["config2_per_rank"] = config2_per_rank(dev, ms_per_step, T)
        except Exception as e:   # noqa: BLE001
            extras["config2_per_rank"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        try:
            if "ms_per_step" in extras.get("config4_1gpu_B1024", {}):
                extras["config4_per_rank"] = config4_per_rank(dev, extras["config4_1gpu_B1024"]["ms_per_step"])
        except Exception as e:   # noqa: BLE001
            extras["config4_per_rank"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    cpu = None if args.no_cpu_baseline else cpu_baseline(w, T, B)
    return {
        "metric": "gradient-steps/sec", "value": round(args.steps / el, 2), "unit": "steps/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "none", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{w['text']}, B={B} windows x temporal_len T={T} (reference default), "
                               f"sample+loss+backward+Adam+polyak per step",
                   "global_batch_windows": B, "temporal_len": T, "transitions_per_step": B * T, "ring": w["ring"],
                   "parallelism": "dp1"},
        "roofline": roofline, "cpu_baseline": cpu,
        "max_step_ms": round(max_step_ms, 4), "host_issue_ms": round(host_issue_ms, 4), "device_ms": round(device_ms, 4),
        "windows_after": [{"value": round(args.steps / e, 2), "max_step_ms": round(m, 4)} for e, m in more],
        "sustained": {"value": round(n_sus / el_sus, 2), "unit": "steps/s", "steps": n_sus, "seconds": round(el_sus, 2)},
        "sampler_roofline": sampler, "kernel_ms_top": top, "also_temporal_len_2": t2, "facade_path": facade,
        "other_configs": extras, "csrc_sha": csrc_hash(),
    }


class DPRun:
    """One data-parallel workload on this rank: Job + the bucketed step (the facade's
    DeepQLearning._distributed_step runs the same three calls)."""

    def __init__(self, w, dev, B_local, T, rank, world, backend, dist, nat, ring_slots=None):
        self.job = Job(w, dev, B_local, T, world=world, rank=rank, ring_slots=ring_slots)
        self.dev, self.backend, self.dist, self.nat = dev, backend, dist, nat
        self.side = torch.cuda.Stream(dev)
        self.ev = [torch.cuda.Event() for _ in range(3)]
        self.bucket = self.job.agent.grad_bucket()   # grads[bucket:] = critics + log_alpha, final after PHASE_GRAD_CRITICS

    def all_reduce(self, g):
        if g.numel() == 0:                           # (FDQL_NO_BUCKETS: the early bucket is empty)
            return
        if self.backend == "nccl":
            self.dist.all_reduce(g)                  # RCCL sum over xGMI; the row weights already carry 1/B_global
        else:                                        # gloo rehearsal: through the host
            h = g.cpu()
            self.dist.all_reduce(h)
            g.copy_(h)

    def step(self, i):
        # two buckets: the critics' gradients are all-reduced on the side stream while the actor / encoder backward runs
        job, agent, nat, side = self.job, self.job.agent, self.nat, self.side
        main_stream = torch.cuda.current_stream(self.dev)
        ev_a, ev_b, ev_red = self.ev
        grads = agent.grads
        k = job.sample(i)
        agent.update(job.xps[k], seed=job.seed, phase=nat.PHASE_GRAD_CRITICS)
        ev_a.record(main_stream)
        with torch.cuda.stream(side):
            side.wait_event(ev_a)
            self.all_reduce(grads[self.bucket:])
        agent.update(None, phase=nat.PHASE_GRAD_REST)
        ev_b.record(main_stream)
        with torch.cuda.stream(side):
            side.wait_event(ev_b)
            self.all_reduce(grads[:self.bucket])
            ev_red.record(side)
        main_stream.wait_event(ev_red)
        agent.update(None, phase=nat.PHASE_APPLY)

    def timed(self, steps, warmup):
        """W untimed steps, then exactly `steps` steps between barrier + synchronize on both sides; MAX over the ranks."""
        dist, dev = self.dist, self.dev
        for i in range(warmup):
            self.step(i)
        torch.cuda.synchronize(dev)                  # (the collectives of a step are this rank's barrier)
        dist.barrier()                               # ranks start the timed region together; no barrier inside it
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in range(steps):
            self.step(warmup + i)
        torch.cuda.synchronize(dev)
        dist.barrier()
        el = time.perf_counter() - t0
        t = torch.tensor([el], device=dev if self.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())


def dp_run_best(w, dev, B_local, T, rank, world, backend, dist, nat, ring_slots=None, calib_steps=30):
    """The launch list a rank runs is a plan decision like the others, taken by measurement where it cannot be known ahead: the
    two-bucket plan (critics' gradients all-reduced beside the actor / encoder backward; three launches more, the dense weight
    gradients in two launches) against the one-bucket plan (the single-GPU launch list, one all-reduce behind the last
    gradient).  Which wins depends on the collective's latency on the node at hand and on the rank's batch (a rank's share of
    a strong-scaled step is launch-bound).  Both are timed for `calib_steps` steps (MAX over ranks: every rank sees the same
    two numbers), the faster one runs the timed region.  FDQL_NO_BUCKETS / FDQL_FORCE_BUCKETS set by the caller: no choice."""
    if os.environ.get("FDQL_NO_BUCKETS") or os.environ.get("FDQL_FORCE_BUCKETS"):
        run = DPRun(w, dev, B_local, T, rank, world, backend, dist, nat, ring_slots=ring_slots)
        return run, {"chosen": "one_bucket" if run.bucket >= run.job.agent.grads.numel() else "two_bucket", "calibration": None}
    runs, ms = {}, {}
    for mode in ("two_bucket", "one_bucket"):
        if mode == "one_bucket":
            os.environ["FDQL_NO_BUCKETS"] = "1"          # read once, when the agent is created
        try:
            runs[mode] = DPRun(w, dev, B_local, T, rank, world, backend, dist, nat, ring_slots=ring_slots)
        finally:
            os.environ.pop("FDQL_NO_BUCKETS", None)
        ms[mode] = round(1e3 * runs[mode].timed(calib_steps, 10) / calib_steps, 4)
    best = min(ms, key=ms.get)
    for mode in list(runs):
        if mode != best:
            del runs[mode]
    torch.cuda.empty_cache()
    return runs[best], {"chosen": best, "calibration": {"steps": calib_steps, "ms_per_step": ms}}


def bench_distributed(args, dev, T, rank, world, backend, dist, nat):
    """N > 1.  `value` = the metric's literal reading: BASELINE config 2 with the GLOBAL batch fixed at 256 windows, split
    256/N per GPU - optimiser iterations (= gradient steps of the 256-window minibatch) per second, "scaling": "strong".
    Beside it: config 2 weak (256 windows PER GPU; optimiser iterations/s and, separately, minibatches/s = N x that - a
    gradient step is never counted N times) and config 4 strong (global batch 1024)."""
    w2, w4 = WORKLOADS["config2"], WORKLOADS["config4"]
    K, W = args.steps, args.warmup
    shrink = os.environ.get("FDQL_BENCH_RING")                           # rehearsal knob (ranks sharing one card)
    ring2 = int(shrink) if shrink else w2["ring"]
    ring4 = int(shrink) if shrink else w4["ring"]
    extra = os.environ.get("FDQL_BENCH_SKIP_EXTRA") is None

    # ---- headline: config 2, GLOBAL batch 256 windows, 256/N per GPU, 1M-slot shard per rank
    B2 = args.batch or w2["B"]
    if B2 % world:
        raise SystemExit(f"global batch {B2} windows does not split over {world} ranks")
    run, plan2 = dp_run_best(w2, dev, B2 // world, T, rank, world, backend, dist, nat, ring_slots=ring2)
    el = run.timed(K, W)
    it_s = K / el
    roofline, top = (roofline_of(run.job, committed_pmc=False) if rank == 0 else (None, None))
    arena_mb = run.job.agent.grads.numel() * 4 / 1e6
    bucket_frac = 1.0 - run.bucket / max(run.job.agent.grads.numel(), 1)
    del run
    torch.cuda.empty_cache()

    weak2, strong4, single4 = None, None, None
    if extra:
        # ---- config 2, 256 windows PER GPU (global batch 256 N): one optimiser iteration consumes N minibatches
        rw, planw = dp_run_best(w2, dev, B2, T, rank, world, backend, dist, nat, ring_slots=ring2)
        nw = max(K, 50)
        ew = rw.timed(nw, max(W, 5))
        weak2 = {"workload": f"{w2['text']} per rank, B={B2} windows PER GPU x T={T} (global batch {B2 * world})", "scaling": "weak",
                 "optimizer_iterations_per_s": round(nw / ew, 2), "minibatches_per_s": round(world * nw / ew, 2),
                 "transitions_per_s": round(nw / ew * B2 * world * T, 0), "ms_per_step": round(1e3 * ew / nw, 4), "steps": nw, "dp_plan": planw,
                 "note": "one all-reduced Adam update = ONE gradient step whatever N is; minibatches_per_s = N x optimizer_iterations_per_s "
                         "counts the 256-window minibatches differentiated per second"}
        del rw
        torch.cuda.empty_cache()
        # ---- config 4, global batch 1024 windows: rank 0 alone on the whole batch first, then the split
        B4 = w4["B"]
        if B4 % world == 0:
            if rank == 0 and os.environ.get("FDQL_BENCH_SKIP_1GPU") is None:
                j1 = Job(w4, dev, B4, T, ring_slots=ring4)
                n1 = max(8, K // 2)
                e1 = j1.timed(n1, 3)
                single4 = {"value": round(n1 / e1, 2), "unit": "steps/s", "global_batch_windows": B4, "steps": n1,
                           "what": "rank 0 alone on the whole batch, same process, before the distributed phase"}
                del j1
                torch.cuda.empty_cache()
            dist.barrier()
            r4, plan4 = dp_run_best(w4, dev, B4 // world, T, rank, world, backend, dist, nat, ring_slots=ring4, calib_steps=20)
            n4 = max(K, 30)
            e4 = r4.timed(n4, max(W, 5))
            strong4 = {"workload": f"{w4['text']} per rank, GLOBAL batch {B4} windows split {B4 // world} per GPU x T={T}",
                       "scaling": "strong", "value": round(n4 / e4, 2), "unit": "steps/s", "ms_per_step": round(1e3 * e4 / n4, 4),
                       "steps": n4, "transitions_per_s": round(n4 / e4 * B4 * T, 0),
                       "grad_arena_MB": round(r4.job.agent.grads.numel() * 4 / 1e6, 2), "same_workload_1gpu": single4, "dp_plan": plan4}
            if single4:
                strong4["speedup_vs_1gpu"] = round(strong4["value"] / single4["value"], 3)
            del r4
            torch.cuda.empty_cache()
    return {
        "metric": "gradient-steps/sec", "value": round(it_s, 2), "unit": "steps/s", "n_gpus": world,
        "steps": K, "warmup": W, "ms_per_step": round(1e3 * el / max(K, 1), 4),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{w2['text']} per rank, GLOBAL batch {B2} windows split {B2 // world} per GPU x temporal_len T={T}, "
                               f"sample+loss+backward, all-reduce of the {arena_mb:.2f} MB gradient arena "
                               + (f"in two buckets ({backend}; {bucket_frac:.0%} of it beside the actor / encoder backward)" if bucket_frac > 0
                                  else f"in one bucket behind the last gradient ({backend})") + ", Adam+polyak on every rank",
                   "windows_per_gpu": B2 // world, "global_batch_windows": B2, "temporal_len": T,
                   "transitions_per_step": B2 * T, "ring": ring2, "parallelism": f"dp{world}",
                   "collective_ranks": dist.get_world_size(), "backend": backend},
        "value_is": "optimiser iterations per second = gradient steps of the 256-window global minibatch (the metric's 'batch=256'); "
                    "the N = 1 point is the single-GPU BENCH line (same global batch)",
        "optimizer_iterations_per_s": round(it_s, 2), "transitions_per_s": round(it_s * B2 * T, 0),
        "dp_plan": plan2, "config2_weak": weak2, "config4_strong": strong4,
        "roofline": roofline,
        "cpu_baseline": {"value": None, "unit": "steps/s", "cores": None, "kind": "port", "sample": "not timed at N > 1",
                         "see": "the N = 1 line of the same bench.py (BENCH_rNN.json): the CPU oracle is timed on rank 0 at N = 1 only, as the "
                                "measurement contract asks; it does not depend on N"},
        "kernel_ms_top": top, "csrc_sha": csrc_hash(),
    }


if __name__ == "__main__":
    main()
