"""The product's N > 1 path in fresh processes: two ranks (torch.multiprocessing spawn, gloo process group, both on
cuda:0) run the facade's DeepQLearning.train_step() with world_size = 2 on their halves of a batch; the result is
compared with one process stepping on the whole batch (franQ/Agent/deepQlearning.py:105-127 is the step,
the reference itself has no multi-GPU path).  bench.py --gpus N runs the same three calls (FDQL_PHASE_GRAD,
all-reduce of the gradient arena, FDQL_PHASE_APPLY) with RCCL."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
T, B, OBS, ACT = 6, 32, 9, 3


class _Space:
    def __init__(self, shape=None, spaces=None):
        if shape is not None:
            self.shape = tuple(shape)
        if spaces is not None:
            self.spaces = spaces


class FixedBatch:
    """A replay read head in the reference's duck-typed sense (wrapper_base_class.py:17-39): always the same batch."""

    def __init__(self, xp):
        self.xp = xp

    def temporal_sample(self):
        return self.xp

    def ready(self):
        return True


def _conf(dev, batch, world):
    from fastdeepqlearning_amd.Agent import AgentConf
    conf = AgentConf()
    conf.obs_space = _Space(spaces={"obs_1d": _Space(shape=(OBS,))})
    conf.action_space = _Space(shape=(ACT,))
    conf.discrete = False
    conf.training_device = conf.inference_device = dev
    conf.batch_size, conf.temporal_len = batch, T
    conf.num_critics, conf.num_q_predictions, conf.latent_state_dim = 3, 4, 64
    conf.pi_hidden_dims, conf.critic_hidden_dims = [64], [64, 64]
    conf.encoder_conf.hidden_features = 48
    conf.encoder_conf.obs_1d_hidden_dims, conf.encoder_conf.joint_hidden_dims = (64,), (48,)
    conf.use_async_train = False
    conf.world_size = world
    return conf


def _global_batch():
    g = torch.Generator().manual_seed(4)
    xp = {"obs_1d": torch.randn(T, B, OBS, generator=g), "action": torch.rand(T, B, ACT, generator=g) * 2 - 1,
          "reward": torch.randn(T, B, 1, generator=g), "mc_return": torch.randn(T, B, 1, generator=g),
          "task_done": (torch.rand(T, B, 1, generator=g) < 0.1).float(),
          "episode_step": (torch.arange(T).view(T, 1, 1) + torch.randint(0, 30, (1, B, 1), generator=g)).float()}
    nt, na = torch.randn(T - 1, B, ACT, generator=g), torch.randn(T - 1, B, ACT, generator=g)
    return xp, nt, na


def _one_step(conf, dev, xp, nt, na, steps=2):
    from fastdeepqlearning_amd import Agent
    agent = Agent.make(conf)                               # seed 0 in every process: identical initial weights
    agent.enable_training([FixedBatch({k: v.to(dev).contiguous() for k, v in xp.items()})])
    for _ in range(steps):
        agent.train_step(noise=(nt.to(dev).contiguous(), na.to(dev).contiguous()))
    torch.cuda.synchronize(dev)
    return agent


def _rank_main(rank, world, port, out_dir):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    xp, nt, na = _global_batch()
    per = B // world
    sl = slice(rank * per, (rank + 1) * per)
    agent = _one_step(_conf(dev, per, world), dev, {k: v[:, sl] for k, v in xp.items()}, nt[:, sl], na[:, sl])
    torch.save({k: v.cpu() for k, v in agent.state_dict().items()}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_multi_process_data_parallel_train_step(tmp_path, world):
    """world ranks (fresh processes, gloo, all on cuda:0) through the facade's bucketed step: PHASE_GRAD_CRITICS, all-reduce
    of arena[bucket:] on a side stream beside PHASE_GRAD_REST, all-reduce of arena[:bucket], PHASE_APPLY."""
    import torch.multiprocessing as mp
    assert torch.cuda.is_available()
    port = 29500 + (os.getpid() % 400) + world
    mp.spawn(_rank_main, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    rs = [torch.load(tmp_path / f"rank{r}.pt") for r in range(world)]
    r0, r1 = rs[0], rs[-1]
    for r in rs[1:]:
        for k in r0:
            assert torch.equal(r0[k], r[k]), k
    dev = torch.device("cuda:0")
    xp, nt, na = _global_batch()
    whole = _one_step(_conf(dev, B, 1), dev, xp, nt, na)
    ref = {k: v.cpu() for k, v in whole.state_dict().items()}
    lr = float(whole.conf.learning_rate)
    assert set(r0) == set(ref)
    close = []
    for k in ref:
        assert torch.equal(r0[k], r1[k]), k                               # the replicas stay bit-identical
        d = float((r0[k] - ref[k]).abs().max())
        assert d <= 2 * 2.1 * lr, (k, d)                                  # two Adam steps, each sign-like where g ~ 0
        close.append(d < 2e-6)
    assert sum(close) >= 0.9 * len(close)                                 # and fp32 summation order elsewhere
    moved = [not torch.equal(ref[k], v.cpu()) for k, v in _one_step(_conf(dev, B, 1), dev, xp, nt, na, steps=0).state_dict().items()]
    assert any(moved)


def _nccl_rank_main(rank, world, port, out_dir):
    """ONE rank over an nccl (= RCCL) process group: the facade's bucketed step exactly as a multi-GPU job runs it."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["FDQL_FORCE_BUCKETS"] = "1"                 # the two-bucket plan at world_size 1 (latched at agent create)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    assert dist.get_backend() == "nccl"
    xp, nt, na = _global_batch()
    conf = _conf(dev, B, 1)
    conf.force_distributed_step = True
    stepped = _one_step(conf, dev, xp, nt, na)             # GRAD_CRITICS | all_reduce(g[b:]) on the side stream | GRAD_REST | all_reduce(g[:b]) | APPLY
    b, n = stepped.native.grad_bucket(), stepped.native.grads.numel()
    assert 0 < b < n, (b, n)                               # both buckets are non-empty: both all-reduces ran through RCCL
    whole = _one_step(_conf(dev, B, 1), dev, xp, nt, na)   # same (bucketed) plan, one FDQL_PHASE_ALL call per step
    # RCCL really executed a collective on this device (a sum over one rank is the identity, so the weights cannot show it)
    probe = torch.arange(1024, device=dev, dtype=torch.float32)
    dist.all_reduce(probe)
    torch.cuda.synchronize(dev)
    torch.save({"dp": {k: v.cpu() for k, v in stepped.state_dict().items()}, "all": {k: v.cpu() for k, v in whole.state_dict().items()},
                "bucket": (b, n), "probe_ok": bool(torch.equal(probe.cpu(), torch.arange(1024, dtype=torch.float32)))},
               os.path.join(out_dir, "nccl.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_step_over_rccl_process_group(tmp_path):
    """The nccl branch of DeepQLearning._all_reduce / bench.py (RCCL; world_size 1 - this box has one GPU): RCCL loads, both
    buckets are all-reduced on the side stream inside the event chain, and the weights after two steps are BIT-equal to the same
    plan stepped with FDQL_PHASE_ALL."""
    import torch.multiprocessing as mp
    assert torch.cuda.is_available()
    port = 29500 + (os.getpid() % 400) + 17
    mp.spawn(_nccl_rank_main, args=(1, port, str(tmp_path)), nprocs=1, join=True)
    r = torch.load(tmp_path / "nccl.pt")
    assert r["probe_ok"]
    assert set(r["dp"]) == set(r["all"])
    for k in r["dp"]:
        assert torch.equal(r["dp"][k], r["all"][k]), k
