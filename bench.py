#!/usr/bin/env python3
"""Headline benchmark: gradient-steps/sec of the SAC/TQC hot path (BASELINE.json).

One "step" = one full train_step of the reference (franQ/Agent/deepQlearning.py:105-127) on
one shard: windowed minibatch sample from the 1M-transition HBM ring + TQC loss + backward +
Adam + polyak.  Workload at every N: BASELINE config 2 — obs 17, act 6, TQC 5 critics x 2
quantiles, MLPs of 256, ring 1,000,000, batch B=256 windows PER GPU of temporal_len T=50
(the reference default, conf.py:38), synthetic data resident in HBM, random-init weights.
N>1: one process per GPU (torchrun), each rank its own ring shard and B windows, gradient
arena all-reduced with RCCL, identical Adam on every rank ("weak" scaling: global batch = N*B).

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` for the
dominant kernel (the grouped fp32-MFMA GEMM, timed with HIP events on the launch stream)
and `cpu_baseline` (the CPU oracle = port of the reference path, timed on this host).
"""
import argparse
import json
import os
import re
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

OBS, ACT, C, Q, HID = 17, 6, 5, 2, 256
RING = 1_000_000
KEYS = ["obs_1d", "action", "reward", "mc_return", "task_done", "episode_done", "episode_step", "idx"]
DIMS = [OBS, ACT, 1, 1, 1, 1, 1, 1]          # 29 f32 = 116 B per transition
MFMA_F32_PEAK_TFLOPS = 157.3                   # MI355X_MICROARCH.md: dense fp32 matrix peak
HBM_PEAK_GBS = 8000.0


def synth_rows(n, seed, device, ep_len=1000, gamma=0.99):
    """Synthetic transitions: obs~N(0,1), action~U(-1,1), reward~N(0,1),
    episodes of 1000 steps, task_done~Bernoulli(1e-3), mc_return = discounted reward-to-go."""
    g = torch.Generator(device=device).manual_seed(seed)
    n_ep = (n + ep_len - 1) // ep_len
    tot = n_ep * ep_len
    obs = torch.randn(tot, OBS, generator=g, device=device)
    act = torch.rand(tot, ACT, generator=g, device=device) * 2 - 1
    rew = torch.randn(n_ep, ep_len, generator=g, device=device)
    ret = torch.empty_like(rew)
    acc = torch.zeros(n_ep, device=device)
    for t in range(ep_len - 1, -1, -1):
        acc = rew[:, t] + gamma * acc
        ret[:, t] = acc
    td = (torch.rand(tot, 1, generator=g, device=device) < 1e-3).float()
    step = torch.arange(ep_len, device=device, dtype=torch.float32).repeat(n_ep).view(-1, 1)
    edone = (step == ep_len - 1).float()
    idx = torch.zeros(tot, 1, device=device)
    rows = torch.cat([obs, act, rew.reshape(-1, 1), ret.reshape(-1, 1), td, edone, step, idx], dim=1)
    return rows[:n].contiguous()


def cpu_baseline(T, B, seconds_budget=25.0):
    """The CPU oracle (oracle/: numpy ring + eager-torch update, a port of the reference path)
    timed on this host: sample + update, steady state, bounded sample of the same workload."""
    from oracle import update as oup
    from oracle.replay import RingOracle, cast_like_loader
    threads = torch.get_num_threads()
    spec = oup.Spec(obs=OBS, act=ACT, C=C, Q=Q, latent=HID, enc_features=HID, enc_hidden=(HID,), joint_hidden=(HID,),
                    pi_hidden=(HID,), critic_hidden=(HID, HID), T=T, B=B)
    st = oup.new_state(spec, oup.init_params(spec, seed=0))
    ring = RingOracle(RING, B, T)
    rows = synth_rows(200_000, 0, "cpu").numpy()          # a 200k-row slice of the ring is enough for timing
    off = np.cumsum([0] + DIMS)
    ring.memory = {k: np.zeros((RING, d), np.float32) for k, d in zip(KEYS, DIMS)}
    for j, k in enumerate(KEYS):
        ring.memory[k][:rows.shape[0]] = rows[:, off[j]:off[j + 1]]
    ring.top, ring.len = rows.shape[0], rows.shape[0]
    rng = np.random.RandomState(0)
    g = torch.Generator().manual_seed(0)

    def sample_only():
        xp = cast_like_loader(ring.temporal_sample(rng=rng))
        return {k: torch.from_numpy(v) for k, v in xp.items()}

    def one():
        xp = sample_only()
        nt, na = torch.randn(T - 1, B, ACT, generator=g), torch.randn(T - 1, B, ACT, generator=g)
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
            if el > budget or n >= 200:
                break
        return n / el, n, el

    # eager torch-CPU on small GEMMs does not scale to a whole socket: time all cores and 16 threads, report the faster
    results = [(timed(threads, seconds_budget / 2), threads)]
    if threads > 16:
        results.append((timed(16, seconds_budget / 2), 16))
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--temporal-len", type=int, default=50)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--also-t2", action="store_true",
                    help="add the temporal_len=2 figure (off by default so that a rocprofv3 summary of the "
                         "default command holds launches of ONE workload only)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with torch.distributed.run (one process per GPU)")
    import torch.distributed as dist
    # FDQL_BENCH_BACKEND=gloo: rehearsal of the N>1 code path on a box with fewer GPUs than ranks
    # (ranks share devices, the gradient all-reduce goes through the host); the driver uses nccl (= RCCL)
    backend = os.environ.get("FDQL_BENCH_BACKEND", "nccl")
    dev = torch.device(f"cuda:{local_rank % max(torch.cuda.device_count(), 1)}")
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from fastdeepqlearning_amd.core import NativeAgent, NativeRing, make_config
    from fastdeepqlearning_amd import _native as nat

    T, B = args.temporal_len, args.batch
    ring = NativeRing(RING, DIMS, dev)
    for c0 in range(0, RING + 1000, 250_000):               # wraps once: len = RING - 1 (quirk q1)
        n = min(250_000, RING + 1000 - c0)
        ring.add_rows(synth_rows(n, 1000 * rank + c0 // 250_000, dev))
    assert len(ring) == RING - 1
    cfg = make_config(OBS, ACT, T, B, n_critics=C, n_quantiles=Q, latent=HID, enc_features=HID, enc_hidden=(HID,),
                      joint_hidden=(HID,), pi_hidden=(HID,), critic_hidden=(HID, HID), world_size=world,
                      keep_frozen_copy=True)
    agent = NativeAgent(cfg, dev)
    agent.init_weights(seed=0)                               # same weights on every rank
    outs = [torch.empty(T, B, d, device=dev) for d in DIMS]
    xp = dict(zip(KEYS, outs))
    seed = 1234 + rank

    def step(i):
        ring.sample_windows(T, B, seed=seed, counter=i, outs=outs)
        if world == 1:
            agent.update(xp, seed=seed)
        else:
            agent.update(xp, seed=seed, phase=nat.PHASE_GRAD)
            if backend == "nccl":
                dist.all_reduce(agent.grads)                 # RCCL sum over xGMI; loss already carries 1/(B*world)
            else:
                g = agent.grads.cpu()
                dist.all_reduce(g)
                agent.grads.copy_(g)
            agent.update(None, phase=nat.PHASE_APPLY)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for i in range(args.warmup):
        step(i)
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    sync()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    ms_per_step = 1e3 * el / max(args.steps, 1)
    value = world * args.steps / el                          # every rank completes one train_step per step

    # ---- per-kernel timing with HIP events on the launch stream (rank 0), after the timed region
    roofline, breakdown, sampler = None, None, None
    if rank == 0:
        acc = {}
        reps = 5
        for r in range(reps):
            for name, ms, fl, by in agent.profile_update(xp, seed=seed):
                a = acc.setdefault(name, [0.0, fl, by, 0])
                a[0] += ms
                a[3] += 1
        # dominant kernel = the tile-shape instantiation of the grouped fp32-MFMA GEMM with the largest total
        # time (rocprofv3 agrees: profiles/*kernel_stats*.csv); the narrow shapes run bandwidth-bound problems
        allg = {k: v for k, v in acc.items() if k.startswith("gemm")}
        by_shape = {}
        for k, v in allg.items():
            by_shape.setdefault(k.split(":")[0], {})[k] = v
        dom_name, dom = max(by_shape.items(), key=lambda kv: sum(v[0] for v in kv[1].values()))
        gemm_ms = sum(v[0] for v in dom.values()) / reps
        gemm_fl = sum(v[1] * v[3] for v in dom.values()) / reps
        n_gemm = sum(v[3] for v in dom.values()) // reps
        all_ms = sum(v[0] for v in allg.values()) / reps
        all_fl = sum(v[1] * v[3] for v in allg.values()) / reps
        total_ms = sum(v[0] for v in acc.values()) / reps
        tf = gemm_fl / max(gemm_ms * 1e-3, 1e-12) / 1e12
        roofline = {"bound": "mfma", "kernel": f"k_gemm_grouped<{dom_name[4:]} tile> (fp32 v_mfma_f32_32x32x2_f32)",
                    "achieved": round(tf, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(tf / MFMA_F32_PEAK_TFLOPS, 4), "traffic": None,
                    "launches_per_step": int(n_gemm), "flops_per_launch": gemm_fl / max(n_gemm, 1),
                    "avg_launch_ms": round(gemm_ms / max(n_gemm, 1), 4), "share_of_step": round(gemm_ms / total_ms, 3),
                    "all_gemm_tile_shapes": {"flops_per_step": all_fl, "ms_per_step": round(all_ms, 4),
                                             "achieved": round(all_fl / (all_ms * 1e-3) / 1e12, 2),
                                             "share_of_step": round(all_ms / total_ms, 3)}}
        # HBM traffic of the dominant kernel cannot be measured from inside this process (PMC needs rocprofv3):
        # it is taken from the committed rocprofv3 --pmc summary of the SAME workload when one is present
        tj = os.path.join(ROOT, "profiles", "r01_dominant_kernel_traffic.json")
        if os.path.exists(tj) and T == 50 and B == 256:
            tr = json.load(open(tj))
            roofline["traffic"] = tr["hbm_bytes_per_launch"]
            roofline["traffic_unit"] = "bytes/launch (HBM read+write, PMC FETCH_SIZE x2 + WRITE_SIZE)"
            roofline["traffic_source"] = "profiles/r01_hbm_traffic_pmc.txt (" + tr["source"] + ")"
            roofline["algorithmic_bytes_per_launch"] = sum(v[2] * v[3] for v in dom.values()) / reps / max(n_gemm, 1)
        # matrix-pipe utilisation and shader clock of the same kernel, from the committed SQ-counter pass
        uj = os.path.join(ROOT, "profiles", "r01_mfma_utilisation_pmc.txt")
        if os.path.exists(uj) and T == 50 and B == 256:
            shape_id = {"128x128": 0, "128x32": 1, "32x128": 2, "64x128": 3, "64x128dual": 4, "64x64": 5, "64x64hf": 6}
            tag = f"k_gemm_grouped<{shape_id.get(dom_name[4:], -1)},"
            wsum = usum = csum = 0.0
            for line in open(uj):
                if tag in line:
                    f = dict(re.findall(r"(\w+)=\s*([\d.]+)", line))
                    w = float(f["dur_us"]) * float(f["n"])      # time-weighted over that kernel's launches
                    wsum += w; usum += w * float(f["mfma_util"]); csum += w * float(f["clock_GHz"])
            if wsum > 0:
                roofline["mfma_pipe_busy_frac"] = round(usum / wsum, 3)
                roofline["shader_clock_ghz_under_load"] = round(csum / wsum, 2)
                roofline["pmc_source"] = ("profiles/r01_mfma_utilisation_pmc.txt (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES "
                                          "SQ_BUSY_CYCLES; peak 157.3 TFLOP/s assumes 2.4 GHz)")
        breakdown = {k: round(v[0] / reps, 4) for k, v in sorted(acc.items(), key=lambda kv: -kv[1][0])[:12]}
        # sampler on its own: HBM-bound gather, algorithmic bytes = 2*T*B*rowbytes + 8*B
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(dev)
        e0.record()
        for i in range(50):
            ring.sample_windows(T, B, seed=seed, counter=10_000 + i, outs=outs)
        e1.record()
        torch.cuda.synchronize(dev)
        s_ms = e0.elapsed_time(e1) / 50
        s_bytes = 2.0 * T * B * sum(DIMS) * 4 + 8 * B
        sampler = {"bound": "hbm", "kernel": "k_draw_starts + k_gather_windows", "ms": round(s_ms, 4),
                   "achieved": round(s_bytes / (s_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": round(s_bytes / (s_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "algorithmic_bytes": s_bytes}

    # ---- secondary figure ("reports both T=50 and T=2"): the same workload as plain 1-step
    # minibatches (temporal_len 2 = one TD pair per window), rank 0 at N=1 only; never `value`
    t2 = None
    if rank == 0 and world == 1 and T != 2 and args.also_t2:
        cfg2 = make_config(OBS, ACT, 2, B, n_critics=C, n_quantiles=Q, latent=HID, enc_features=HID, enc_hidden=(HID,),
                           joint_hidden=(HID,), pi_hidden=(HID,), critic_hidden=(HID, HID), world_size=1,
                           keep_frozen_copy=True)
        ag2 = NativeAgent(cfg2, dev)
        ag2.init_weights(seed=0)
        outs2 = [torch.empty(2, B, d, device=dev) for d in DIMS]
        xp2 = dict(zip(KEYS, outs2))
        for i in range(20):
            ring.sample_windows(2, B, seed=seed, counter=20_000 + i, outs=outs2)
            ag2.update(xp2, seed=seed)
        torch.cuda.synchronize(dev)
        n2 = 200
        t0 = time.perf_counter()
        for i in range(n2):
            ring.sample_windows(2, B, seed=seed, counter=30_000 + i, outs=outs2)
            ag2.update(xp2, seed=seed)
        torch.cuda.synchronize(dev)
        e2 = time.perf_counter() - t0
        t2 = {"temporal_len": 2, "value": round(n2 / e2, 1), "unit": "steps/s", "ms_per_step": round(1e3 * e2 / n2, 4),
              "transitions_per_step": 2 * B, "note": "launch/latency-bound: 2.6 GFLOP per step"}
        del ag2

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(T, B)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        out = {
            "metric": "gradient-steps/sec", "value": round(value, 2), "unit": "steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE config 2: TQC 5x2 quantile critics, obs=17 act=6, MLP(256), "
                                   f"1M-transition HBM ring, B={B} windows/GPU x temporal_len T={T} "
                                   f"(reference default), sample+loss+backward+Adam+polyak per step",
                       "global_batch_windows": B * world, "temporal_len": T, "transitions_per_step": B * world * T,
                       "ring": RING, "parallelism": f"dp{world}"},
            "roofline": roofline, "cpu_baseline": cpu,
            "sampler_roofline": sampler, "kernel_ms_top": breakdown, "also_temporal_len_2": t2,
        }
        print(json.dumps(out))


if __name__ == "__main__":
    main()
