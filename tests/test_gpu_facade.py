"""GPU tests of the franQ-shaped facade (Replay / Agent) — the reference's own replay tests
restated against the native objects, the write wrappers against the reference's emitted
sequences, and the agent object's life cycle."""
import random

import numpy as np
import pytest
import torch

from golden_io import load, spec_from_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def test_reference_test_size(dev):
    """tests/test_replays.py:36-57 (AsyncReplayMemory length counts up to and saturates at maxlen)."""
    from fastdeepqlearning_amd.Replay import AsyncReplayMemory
    maxlen = 500
    r = AsyncReplayMemory(maxlen=maxlen, batch_size=32, temporal_len=10, device=dev)
    for i in range(maxlen * 2):
        r.add({"obs": np.random.uniform(size=[10]), "action": 2})
        if i < maxlen:
            assert len(r) == i + 1
        assert (1 + i) - maxlen <= len(r) <= i + 1
    for i in range(maxlen):
        r.add({"obs": np.random.uniform(size=[10]), "action": 2})
    assert len(r) == maxlen


def test_reference_test_temporal_consistency(dev):
    """tests/test_replays.py:60-84: obs[1:] == obs[:-1] + 1 on every window while the ring has not wrapped."""
    from fastdeepqlearning_amd.Replay import AsyncReplayMemory
    from fastdeepqlearning_amd.Replay.wrappers import TorchDataLoader
    maxlen, B, T, obs_size = 5000, 256, 10, 10
    r = AsyncReplayMemory(maxlen=maxlen, batch_size=B, temporal_len=T, device=dev)
    for i in range(maxlen // 2):
        r.add({"obs": np.ones([obs_size]) * i, "action": 2})
    l = TorchDataLoader(r, dev)
    for _ in range(20):
        xp = l.temporal_sample()
        obs = xp["obs"]
        assert tuple(obs.shape) == (T, B, obs_size) and obs.dtype == torch.float32 and obs.is_cuda
        assert (obs[1:] == (obs[:-1] + 1)).all()
        assert tuple(xp["action"].shape) == (T, B, 1)


def test_reference_test_nstep_return(dev):
    """tests/test_replays.py:16-33 through the native wrapper + ring."""
    from fastdeepqlearning_amd import Replay
    discount, n_step = 0.99, 1000
    replay = Replay.ReplayMemory(1001, batch_size=128, temporal_len=1, device=dev)
    replay = Replay.wrappers.NStepReturn(replay, n_step=n_step, discount=discount)
    for i in range(n_step):
        replay.add({"reward": float(i == (n_step - 1)), "episode_done": i == (n_step - 1), "step": i})
    for j in range(20):
        s = replay.sample()
        assert np.allclose(s["mc_return"].cpu().numpy(), discount ** (n_step - 1 - s["step"].cpu().numpy()))


class Sink:
    def __init__(self, dev):
        self.rows, self.device = [], dev

    def add(self, d):
        self.rows.append(dict(d))

    def stacked(self):
        keys = sorted(self.rows[0].keys())
        return {k: np.stack([np.asarray(r[k], np.float64).reshape(-1) for r in self.rows]) for k in keys}


@pytest.mark.parametrize("case", ["sparse_1000", "dense_two_eps", "pop_quirk", "single_step"])
def test_nstep_wrapper_matches_reference_sequence(dev, case):
    from fastdeepqlearning_amd.Replay.wrappers import NStepReturn
    g = load("nstep")[case]
    sink = Sink(dev)
    w = NStepReturn(sink, int(g["n_step"]), float(g["gamma"]))
    inp = g["in"]
    for i in range(inp["reward"].shape[0]):
        w.add({"reward": float(inp["reward"][i, 0]), "episode_done": bool(inp["episode_done"][i, 0]),
               "episode_step": int(inp["episode_step"][i, 0]), "obs_1d": inp["obs_1d"][i]})
    out = sink.stacked()
    for k, v in g["out"].items():
        np.testing.assert_array_equal(out[k].astype(np.float32), np.asarray(v, np.float32), err_msg=k)


@pytest.mark.parametrize("case", ["final", "random", "final_nstep"])
def test_her_wrapper_matches_reference_sequence(dev, case):
    from fastdeepqlearning_amd.Replay.wrappers import HindsightNStepReplay, NStepReturn, SparseL2Reward
    g = load("her")[case]
    sink = Sink(dev)
    inner = NStepReturn(sink, 1000, float(g["gamma"])) if int(g["nstep"]) else sink
    random.seed(3)
    w = HindsightNStepReplay(inner, SparseL2Reward(float(g["thr"]), -1.0), mode=str(g["mode"]), device=dev)
    inp = g["in"]
    for i in range(inp["reward"].shape[0]):
        w.add({"obs_1d": inp["obs_1d"][i], "achieved_goal": inp["achieved_goal"][i], "desired_goal": inp["desired_goal"][i],
               "action": inp["action"][i], "reward": float(inp["reward"][i, 0]), "task_done": bool(inp["task_done"][i, 0]),
               "episode_done": bool(inp["episode_done"][i, 0]), "episode_step": int(inp["episode_step"][i, 0]), "info": {}})
    out = sink.stacked()
    for k, v in g["out"].items():
        np.testing.assert_allclose(out[k], np.asarray(v, np.float64), rtol=0, atol=1e-6, err_msg=k)


# ---------------------------------------------------------------------------------------
# Write-path ingestion: a finished episode -> ring in one fdql_ring_append_episode
# ---------------------------------------------------------------------------------------
def _ring_contents(mem, n):
    got = mem[np.arange(n)]
    return {k: v.cpu().numpy().reshape(n, -1) for k, v in got.items()}


@pytest.mark.parametrize("case", ["sparse_1000", "dense_two_eps", "pop_quirk", "single_step"])
def test_fused_nstep_append_matches_reference_sequence(dev, case):
    """NStepReturn directly over the ring: the ring then holds exactly the reference's emitted sequence
    (nstep golden, incl. the duplicated _pop record of quirk q3), bit for bit."""
    from fastdeepqlearning_amd.Replay import ReplayMemory
    from fastdeepqlearning_amd.Replay.wrappers import NStepReturn
    g = load("nstep")[case]
    mem = ReplayMemory(4096, 4, 2, device=dev)
    w = NStepReturn(mem, int(g["n_step"]), float(g["gamma"]))
    assert w._fused_target() is mem
    inp = g["in"]
    for i in range(inp["reward"].shape[0]):
        w.add({"reward": float(inp["reward"][i, 0]), "episode_done": bool(inp["episode_done"][i, 0]),
               "episode_step": int(inp["episode_step"][i, 0]), "obs_1d": inp["obs_1d"][i]})
    n_out = g["out"]["reward"].shape[0]
    assert len(mem) == n_out
    got = _ring_contents(mem, n_out)
    for k, v in g["out"].items():
        np.testing.assert_array_equal(got[k], np.asarray(v, np.float32), err_msg=k)


@pytest.mark.parametrize("case", ["final", "random", "final_nstep"])
def test_fused_her_append_matches_reference_sequence(dev, case):
    """HindsightNStepReplay (over NStepReturn) over the ring, device reward function: one call per episode,
    ring contents == the reference's emitted sequence (her golden)."""
    from fastdeepqlearning_amd.Replay import ReplayMemory
    from fastdeepqlearning_amd.Replay.wrappers import HindsightNStepReplay, NStepReturn, SparseL2Reward
    g = load("her")[case]
    mem = ReplayMemory(4096, 4, 2, device=dev)
    inner = NStepReturn(mem, 1000, float(g["gamma"])) if int(g["nstep"]) else mem
    random.seed(3)
    w = HindsightNStepReplay(inner, SparseL2Reward(float(g["thr"]), -1.0), mode=str(g["mode"]), device=dev)
    calls = []
    orig = mem.append_episode
    mem.append_episode = lambda *a, **k: calls.append(1) or orig(*a, **k)
    inp = g["in"]
    for i in range(inp["reward"].shape[0]):
        w.add({"obs_1d": inp["obs_1d"][i], "achieved_goal": inp["achieved_goal"][i], "desired_goal": inp["desired_goal"][i],
               "action": inp["action"][i], "reward": float(inp["reward"][i, 0]), "task_done": bool(inp["task_done"][i, 0]),
               "episode_done": bool(inp["episode_done"][i, 0]), "episode_step": int(inp["episode_step"][i, 0]), "info": {}})
    assert len(calls) == len(g["ep_lens"])          # one fused call per episode
    n_out = g["out"]["reward"].shape[0]
    assert len(mem) == n_out
    got = _ring_contents(mem, n_out)
    for k, v in g["out"].items():
        np.testing.assert_allclose(got[k], np.asarray(v, np.float64), rtol=0, atol=1e-6, err_msg=k)


@pytest.mark.parametrize("mode,n_step", [("final", 3), ("random", 3), ("final", 1), ("random", 50)])
def test_fused_append_equals_per_record_path(dev, mode, n_step):
    """Random goal episodes of mixed lengths (shorter than, equal to and longer than n_step, so _pop's
    duplicate appears in both the real and the hindsight block; ring wrap included): the fused call leaves
    the ring bit-identical to the per-record wrapper path (itself pinned to the reference sequences)."""
    from fastdeepqlearning_amd.Replay import ReplayMemory
    from fastdeepqlearning_amd.Replay.wrappers import HindsightNStepReplay, NStepReturn, SparseL2Reward
    from fastdeepqlearning_amd.Replay.wrappers.wrapper_base_class import ReplayMemoryWrapper
    fn = SparseL2Reward(0.6, -1.0)
    rng = np.random.RandomState(11)
    episodes = []
    for L in (1, 2, 3, 4, 9, 70, 130):
        ag = rng.uniform(-1, 1, (L, 2)).astype(np.float32)
        dg = np.repeat(rng.uniform(-1, 1, (1, 2)).astype(np.float32), L, 0)
        episodes.append([{"obs_1d": rng.standard_normal(3).astype(np.float32), "achieved_goal": ag[i], "desired_goal": dg[i],
                          "action": rng.uniform(-1, 1, 2).astype(np.float32), "reward": float(fn(ag[i], dg[i])[0]) + 0.25 * (i % 3),
                          "task_done": bool(fn(ag[i], dg[i])[1]), "episode_done": i == L - 1, "episode_step": i, "info": {}}
                         for i in range(L)])
    mems = []
    for fused in (True, False):
        mem = ReplayMemory(300, 4, 2, device=dev)           # 2*(219 + pops) rows > 300 slots: wraps
        base = mem if fused else ReplayMemoryWrapper(mem)    # a wrapper in between disables the fused path
        stack = HindsightNStepReplay(NStepReturn(base, n_step, 0.97), fn, mode=mode, device=dev)
        random.seed(5)
        for ep in episodes:
            for rec in ep:
                stack.add(dict(rec))
        mems.append(mem)
    a, b = mems
    assert len(a) == len(b) and a._ring.top == b._ring.top
    ca, cb = _ring_contents(a, len(a)), _ring_contents(b, len(b))   # len caps at maxlen-1 (q1)
    assert set(ca) == set(cb) and "mc_return" in ca
    for k in ca:
        np.testing.assert_array_equal(ca[k], cb[k], err_msg=k)


class _Space:
    def __init__(self, shape=None, n=None, spaces=None):
        if shape is not None:
            self.shape = tuple(shape)
        if n is not None:
            self.n = n
        if spaces is not None:
            self.spaces = spaces


def _conf(dev, T=4, B=16):
    from fastdeepqlearning_amd.Agent import AgentConf
    conf = AgentConf()
    conf.obs_space = _Space(spaces={"obs_1d": _Space(shape=(5,))})
    conf.action_space = _Space(shape=(3,))
    conf.discrete = False
    conf.training_device = conf.inference_device = dev
    conf.batch_size, conf.temporal_len, conf.replay_size = B, T, 4000
    conf.num_critics, conf.num_q_predictions, conf.latent_state_dim = 3, 4, 32
    conf.pi_hidden_dims, conf.critic_hidden_dims = [32], [32, 32]
    conf.encoder_conf.hidden_features = 32
    conf.encoder_conf.obs_1d_hidden_dims, conf.encoder_conf.joint_hidden_dims = (32,), (32,)
    conf.use_async_train = False
    conf.num_instances = 2
    return conf


def test_agent_facade_life_cycle(dev, tmp_path):
    """Replay.make + Agent.make wired like franQ.Runner.__init__ (runner.py:30-42): feed records
    through the write heads, train from the read heads, save / load."""
    from fastdeepqlearning_amd import Agent, Replay
    conf = _conf(dev)
    g = load("update_tqc_small")
    read_heads, write_heads = Replay.make(conf)
    agent = Agent.make(conf)
    assert set(agent.state_dict().keys()) == set(g["init"].keys())      # SURVEY a22 names
    for k, v in agent.state_dict().items():
        assert tuple(v.shape) == tuple(g["init"][k].shape), k
    agent.enable_training(read_heads)
    rng = np.random.RandomState(0)
    for ep in range(6):
        for i in range(120):
            row = {"obs_1d": rng.standard_normal(5), "action": rng.uniform(-1, 1, 3).astype(np.float32),
                   "reward": float(rng.standard_normal()), "task_done": bool(rng.rand() < 0.01),
                   "episode_done": i == 119, "episode_step": i, "idx": 0}
            for w in write_heads:
                w.add(dict(row))
    before = {k: v.clone() for k, v in agent.state_dict().items()}
    for _ in range(3):
        agent.train_step()
    assert agent.iteration == 3 * len(read_heads)
    sc = agent.native.scalars()
    assert np.isfinite(sc["loss"]) and sc["step"] == 6
    # two shards x a 3-buffer sample pool = 6 batch address sets: all of them stay in the agent's plan cache
    built = agent.native.stats()["plans_built"]
    assert built == 3 * len(read_heads)
    for _ in range(6):
        agent.train_step()
    assert agent.native.stats()["plans_built"] == built
    assert agent.native.scalars()["step"] == 18
    after = agent.state_dict()
    assert any(not torch.equal(before[k], after[k]) for k in before)
    action, hidden, info = agent.act({"obs_1d": torch.randn(7, 5), "exploit_mask": torch.zeros(7, 1, dtype=torch.bool)})
    assert tuple(action.shape) == (7, 3) and hidden is None and float(action.abs().max()) <= 1.0
    agent.save(tmp_path)
    agent2 = type(agent).load_from_file(tmp_path)
    for k, v in agent.state_dict().items():
        assert torch.equal(v, agent2.state_dict()[k]), k
    assert agent2.iteration == agent.iteration


def test_trainer_summaries_and_hook(dev):
    """What franQ's trainer logs every log_interval steps (deepQlearning.py:114-122, 231-247; q_pred_var:
    distributional_soft_actor_critic.py:65-67): fdql_agent_summaries against the same expressions in torch on the step's own
    buffers, and the facade's summary_hook firing on the reference's schedule."""
    from fastdeepqlearning_amd import Agent, Replay
    conf = _conf(dev)
    conf.log_interval = 2
    seen = []
    conf.summary_hook = lambda step, d: seen.append((step, dict(d)))
    read_heads, write_heads = Replay.make(conf)
    agent = Agent.make(conf)
    agent.enable_training(read_heads[:1])
    rng = np.random.RandomState(1)
    for ep in range(6):
        for i in range(120):
            write_heads[0].add({"obs_1d": rng.standard_normal(5), "action": rng.uniform(-1, 1, 3).astype(np.float32),
                                "reward": float(rng.standard_normal()), "task_done": bool(rng.rand() < 0.05),
                                "episode_done": i == 119, "episode_step": i, "idx": 0})
    for _ in range(9):
        agent.train_step()
    assert [s for s, _ in seen] == [0, 2, 4, 6, 8]                       # step % log_interval == 0, counted like the reference
    assert any(k.startswith("GradNorms/") for k in seen[0][1]) and any(k.startswith("GradNorms/") for k in seen[4][1])
    assert not any(k.startswith("GradNorms/") for k in seen[1][1])       # every 4 x log_interval only
    nat = agent.native
    T, B = int(nat.cfg.T), int(nat.cfg.B)
    sm = nat.summaries(grad_norms=True)
    q = nat.debug("q_pred", (T - 1, B, nat.cfg.n_critics * nat.cfg.n_quantiles)).double()
    ic = nat.debug("is_contiguous", (T - 1, B)).double()
    assert abs(sm["q_pred_var"] - float(q.var(-1).mean())) <= 1e-5 * max(1.0, float(q.var(-1).mean()))
    assert abs(sm["valid_portion_mean"] - float(ic.mean())) < 1e-6
    assert abs(sm["valid_portion_max"] - float(ic.sum(0).max() / T)) < 1e-6
    assert abs(sm["valid_portion_min"] - float(ic.sum(0).min() / T)) < 1e-6
    for k, v in sm["grad_norms"].items():
        ref = float(nat.grad_views[k].double().norm())
        assert abs(v - ref) <= 1e-5 * max(ref, 1e-6), k


def test_phase_split_equals_single_call(dev):
    """FDQL_PHASE_GRAD + FDQL_PHASE_APPLY (the multi-GPU sequence, all-reduce in between) gives
    bit-identical weights to FDQL_PHASE_ALL."""
    from fastdeepqlearning_amd import _native as nat
    from test_gpu_parity import _agent_for
    g = load("update_tqc_small")
    spec = spec_from_case(g["case"])
    rec = g["step0"]
    xp = {k: torch.tensor(v).to(dev) for k, v in rec["batch"].items()}
    nt, na = torch.tensor(rec["noise_target"]).to(dev), torch.tensor(rec["noise_actor"]).to(dev)
    a1, a2 = _agent_for(spec, dev), _agent_for(spec, dev)
    for a in (a1, a2):
        a.load_tensors({k: torch.tensor(v) for k, v in g["init"].items()})
    a1.update(xp, nt, na)
    a2.update(xp, nt, na, phase=nat.PHASE_GRAD)
    a2.update(None, phase=nat.PHASE_APPLY)
    for k in a1.tensors:
        assert torch.equal(a1.tensors[k], a2.tensors[k]), k


@pytest.mark.parametrize("world", [1, 2])
def test_bucketed_phases_equal_one_grad_call(dev, world):
    """FDQL_PHASE_GRAD_CRITICS + FDQL_PHASE_GRAD_REST (+ APPLY) - the data-parallel sequence whose first bucket
    (arena[bucket:]: critics + log_alpha) is all-reduced beside the second part - leaves bit-identical gradients and
    weights to FDQL_PHASE_GRAD (+ APPLY), and arena[bucket:] is already final after the first part.  world 1: the plan is
    not bucketed (bucket = arena length) and the first part does nothing but bind the batch."""
    import dataclasses
    from fastdeepqlearning_amd import _native as nat
    from test_gpu_parity import _agent_for
    g = load("update_tqc_small")
    spec = dataclasses.replace(spec_from_case(g["case"]), world_size=world)
    rec = g["step0"]
    xp = {k: torch.tensor(v).to(dev) for k, v in rec["batch"].items()}
    nt, na = torch.tensor(rec["noise_target"]).to(dev), torch.tensor(rec["noise_actor"]).to(dev)
    a1, a2 = _agent_for(spec, dev, world_size=world), _agent_for(spec, dev, world_size=world)
    for a in (a1, a2):
        a.load_tensors({k: torch.tensor(v) for k, v in g["init"].items()})
    b = a2.grad_bucket()
    assert (b < a2.grads.numel()) == (world > 1) and b % 4 == 0
    a1.update(xp, nt, na, phase=nat.PHASE_GRAD)
    a2.update(xp, nt, na, phase=nat.PHASE_GRAD_CRITICS)
    torch.cuda.synchronize(dev)
    assert torch.equal(a1.grads[b:], a2.grads[b:])                 # the early bucket is final
    a2.update(None, phase=nat.PHASE_GRAD_REST)
    assert torch.equal(a1.grads, a2.grads)
    a1.update(None, phase=nat.PHASE_APPLY)
    a2.update(None, phase=nat.PHASE_APPLY)
    for k in a1.tensors:
        assert torch.equal(a1.tensors[k], a2.tensors[k]), k


def test_data_parallel_sharding_on_device(dev):
    """The multi-GPU sequence on one card: two agents configured as ranks of a world of 2 take half the windows each
    (FDQL_PHASE_GRAD), their gradient arenas are summed (what the RCCL all-reduce does) and applied
    (FDQL_PHASE_APPLY); the result matches one agent stepping on the whole batch (weights within fp32 summation
    order, every replica identical bit for bit)."""
    import dataclasses
    from fastdeepqlearning_amd import _native as nat
    from oracle import update as oup
    from test_gpu_parity import _agent_for
    spec = oup.Spec(obs=9, act=3, C=3, Q=4, latent=64, enc_features=48, enc_hidden=(64,), joint_hidden=(48,), pi_hidden=(64,),
                    critic_hidden=(64, 64), T=6, B=32)
    half = dataclasses.replace(spec, B=16, world_size=2)
    params = oup.init_params(spec, seed=4)
    g = torch.Generator().manual_seed(4)
    T, B, A = spec.T, spec.B, spec.act
    xp = {"obs_1d": torch.randn(T, B, spec.obs, generator=g), "action": torch.rand(T, B, A, generator=g) * 2 - 1,
          "reward": torch.randn(T, B, 1, generator=g), "mc_return": torch.randn(T, B, 1, generator=g),
          "task_done": (torch.rand(T, B, 1, generator=g) < 0.1).float(),
          "episode_step": (torch.arange(T).view(T, 1, 1) + torch.randint(0, 30, (1, B, 1), generator=g)).float()}
    nt, na = torch.randn(T - 1, B, A, generator=g), torch.randn(T - 1, B, A, generator=g)
    whole = _agent_for(spec, dev)
    ranks = [_agent_for(half, dev, world_size=2), _agent_for(half, dev, world_size=2)]
    for ag in [whole] + ranks:
        ag.load_tensors(params)
    whole.update({k: v.to(dev) for k, v in xp.items()}, nt.to(dev), na.to(dev))
    for r, ag in enumerate(ranks):
        sl = slice(16 * r, 16 * (r + 1))
        ag.update({k: v[:, sl].contiguous().to(dev) for k, v in xp.items()}, nt[:, sl].contiguous().to(dev),
                  na[:, sl].contiguous().to(dev), phase=nat.PHASE_GRAD)
    total = ranks[0].grads + ranks[1].grads            # all_reduce(sum)
    ref = whole.grads
    assert float((total - ref).abs().max() / ref.abs().max()) < 2e-5
    for ag in ranks:
        ag.grads.copy_(total)
        ag.update(None, phase=nat.PHASE_APPLY)
    lr = spec.lr
    for k in whole.tensors:
        assert torch.equal(ranks[0].tensors[k], ranks[1].tensors[k]), k          # replicas stay identical
        d = float((ranks[0].tensors[k] - whole.tensors[k]).abs().max())
        assert d <= 2.1 * lr, (k, d)                                            # Adam's step is sign-like where g ~ 0
    close = [float((ranks[0].tensors[k] - whole.tensors[k]).abs().max()) < 1e-6 for k in whole.tensors]
    assert sum(close) >= 0.9 * len(close)


def test_config4_full_size_two_rank_split_on_device(dev):
    """BASELINE config 4 at its full size - TQC 5 x 25 quantiles, obs 376, act 17, MLP 256, T = 50, GLOBAL batch of 1024
    windows from 2 M-slot ring shards - as bench.py --gpus 2 runs it, but with both ranks on this card: each rank samples
    512 windows from its own shard (fdql_ring_sample_windows, Philox starts), FDQL_PHASE_GRAD, gradient arenas summed,
    FDQL_PHASE_APPLY.  Size-independent properties (the CPU oracle would need minutes here): the summed arena equals the
    gradient of ONE agent stepping on the concatenated 1024 windows (fp32 summation order), both replicas end bit-identical,
    and the sampled windows are bit-exact gathers of the ring rows they name."""
    from fastdeepqlearning_amd import _native as nat
    from fastdeepqlearning_amd.core import NativeAgent, NativeRing, make_config
    T, Bg, obs, act = 50, 1024, 376, 17
    keys = ["obs_1d", "action", "reward", "mc_return", "task_done", "episode_done", "episode_step", "idx"]
    dims = [obs, act, 1, 1, 1, 1, 1, 1]
    slots = 2_000_000
    mk = lambda B, world: NativeAgent(make_config(obs, act, T, B, n_critics=5, n_quantiles=25, latent=256, enc_features=256,
                                                  enc_hidden=(256,), joint_hidden=(256,), pi_hidden=(256,), critic_hidden=(256, 256),
                                                  world_size=world), dev)
    whole, ranks = mk(Bg, 1), [mk(Bg // 2, 2), mk(Bg // 2, 2)]
    whole.init_weights(3)
    for ag in ranks:
        ag.load_tensors({k: v.clone() for k, v in whole.tensors.items()})
    g = torch.Generator(device=dev).manual_seed(5)
    batches = []
    for r in range(2):
        ring = NativeRing(slots, dims, dev)
        n = 600_000   # rows written to this 2 M-slot shard (3.8 GB of f32 rows)
        rows = torch.randn(n, sum(dims), device=dev, generator=g)
        ep = torch.arange(n, device=dev) % 1000
        rows[:, obs + act + 2] = (torch.rand(n, device=dev, generator=g) < 0.001).float()     # task_done
        rows[:, obs + act + 3] = (ep == 999).float()                                            # episode_done
        rows[:, obs + act + 4] = ep.float()                                                     # episode_step
        rows[:, obs + act + 5] = torch.arange(n, device=dev).float()                            # idx
        rows[:, obs:obs + act].clamp_(-0.999, 0.999)
        ring.add_rows(rows)
        outs = [torch.empty(T, Bg // 2, d, device=dev) for d in dims]
        _, starts = ring.sample_windows(T, Bg // 2, seed=11 + r, counter=0, outs=outs, return_starts=True)
        xp = dict(zip(keys, outs))
        # the gather is exact: window b at time t is ring row starts[b] + t
        sel = torch.randint(0, Bg // 2, (16,), device=dev, generator=g)
        for b in sel.tolist():
            want = rows[int(starts[b]):int(starts[b]) + T]
            got = torch.cat([o[:, b] for o in outs], dim=1)
            assert torch.equal(got, want)
        batches.append(xp)
        del ring, rows
    use = ["obs_1d", "action", "reward", "mc_return", "task_done", "episode_step"]
    nz = [torch.randn(T - 1, Bg, act, device=dev, generator=g) for _ in range(2)]
    cat = {k: torch.cat([batches[0][k], batches[1][k]], dim=1).contiguous() for k in use}
    whole.update(cat, nz[0], nz[1], phase=nat.PHASE_GRAD)
    for r, ag in enumerate(ranks):
        sl = slice(512 * r, 512 * (r + 1))
        ag.update({k: batches[r][k] for k in use}, nz[0][:, sl].contiguous(), nz[1][:, sl].contiguous(), phase=nat.PHASE_GRAD)
    total = ranks[0].grads + ranks[1].grads
    ref = whole.grads
    torch.cuda.synchronize()
    assert float((total - ref).abs().max() / ref.abs().max()) < 2e-5
    for ag in ranks:
        ag.grads.copy_(total)
        ag.update(None, phase=nat.PHASE_APPLY)
    whole.update(None, phase=nat.PHASE_APPLY)
    torch.cuda.synchronize()
    for k in whole.tensors:
        assert torch.equal(ranks[0].tensors[k], ranks[1].tensors[k]), k
    assert ranks[0].scalars()["step"] == 1


def test_device_noise_statistics_and_determinism(dev):
    """Philox path (perf runs): same seed/step -> same result; actions inside (-1, 1); noise ~ N(0,1)."""
    from test_gpu_parity import _agent_for
    g = load("update_tqc_small")
    spec = spec_from_case(g["case"])
    xp = {k: torch.tensor(v).to(dev) for k, v in g["step0"]["batch"].items()}
    outs = []
    for rep in range(2):
        a = _agent_for(spec, dev)
        a.load_tensors({k: torch.tensor(v) for k, v in g["init"].items()})
        a.update(xp, seed=77)
        outs.append((a.debug("pi").clone(), a.debug("noise_actor").clone(), a.scalars()["loss"]))
    assert torch.equal(outs[0][0], outs[1][0]) and outs[0][2] == outs[1][2]
    assert float(outs[0][0].abs().max()) <= 1.0   # tanh saturates to exactly 1 in fp32, as on the CPU
    from fastdeepqlearning_amd.core import NativeAgent, make_config
    big = NativeAgent(make_config(5, 3, 50, 256, n_critics=2, n_quantiles=5, latent=32, enc_features=32, enc_hidden=(32,),
                                  joint_hidden=(32,), pi_hidden=(32,), critic_hidden=(32, 32)), dev)
    big.init_weights(0)
    T, B = 50, 256
    xpb = {"obs_1d": torch.randn(T, B, 5, device=dev), "action": torch.rand(T, B, 3, device=dev) * 2 - 1,
           "reward": torch.randn(T, B, 1, device=dev), "mc_return": torch.randn(T, B, 1, device=dev),
           "task_done": torch.zeros(T, B, 1, device=dev),
           "episode_step": torch.arange(T, device=dev).view(T, 1, 1).expand(T, B, 1).float().contiguous()}
    big.update(xpb, seed=5)
    z = big.debug("noise_actor")
    assert abs(float(z.mean())) < 0.02 and abs(float(z.std()) - 1.0) < 0.02


# --------------------------------------------------------------------------- HER-vmap (shim-pinned: the reference files ran on a jax stand-in)
def _l2_callable(ag, dg, thr=0.25):
    d = np.linalg.norm(np.asarray(ag, np.float32) - np.asarray(dg, np.float32))
    reward = np.float32(-1.0) if d > thr else np.float32(0.0)
    return reward, bool(reward == 0)


@pytest.mark.parametrize("case", ["k4", "k32_pop"])
@pytest.mark.parametrize("reward_kind", ["device", "device_per_record", "host_callable", "episode_list", "episode_stacked", "episode_stacked_host_fn"])
def test_her_vmap_stack_matches_reference(dev, case, reward_kind):
    """HindsightVmapWrite -> NStepReturnVmap -> HBM ring -> HindsightVmapRead against what the reference's own
    her_vmap.py / nstep_return_vmap.py emitted and sampled (tests/golden/her_vmap.npz): fdql_episode_her_vmap (or the
    host loop for an arbitrary reward callable), fdql_episode_mc_return_vmap (q10, and the _pop duplicate q3) and the
    column select inside the gather, fdql_ring_sample_windows_sel (q11) - bit for bit."""
    from fastdeepqlearning_amd.Replay import ReplayMemory
    from fastdeepqlearning_amd.Replay.wrappers import HindsightVmapRead, HindsightVmapWrite, NStepReturnVmap, SparseL2Reward
    g = load("her_vmap")[case]
    K, T, B = int(g["K"]), int(g["T"]), int(g["B"])
    ring = ReplayMemory(64, B, T, device=dev)
    inner = NStepReturnVmap(ring, int(g["n_step"]), float(g["gamma"]))
    # "device": the fused path (relabel, returns and packed rows stay on the device, one append per episode);
    # "device_per_record": the same kernels, records re-added one by one; "host_callable": an arbitrary Python reward function
    # "episode_list" / "episode_stacked": HindsightVmapWrite.add_episode() with the finished episode as a list of records / as ONE
    # dict of stacked columns (no per-record Python); "episode_stacked_host_fn": stacked columns with a Python reward callable
    host_fn = reward_kind in ("host_callable", "episode_stacked_host_fn")
    fn = _l2_callable if host_fn else SparseL2Reward(float(g["thr"]), -1.0)
    w = HindsightVmapWrite(inner, fn, num_virtual_goals=K, fused=reward_kind != "device_per_record")
    if reward_kind in ("device", "episode_list", "episode_stacked"):
        assert w._fused_target() is not None
    inp = g["in"]
    np.random.seed(100 + K)                 # the generator's seed: her_vmap.py:75 draws from numpy's global state
    recs = [{"obs_1d": inp["obs_1d"][i], "achieved_goal": inp["achieved_goal"][i], "desired_goal": inp["desired_goal"][i],
             "action": inp["action"][i], "reward": float(inp["reward"][i, 0]), "task_done": bool(inp["task_done"][i, 0]),
             "episode_done": bool(inp["episode_done"][i, 0]), "episode_step": int(inp["episode_step"][i, 0]), "info": {}}
            for i in range(inp["reward"].shape[0])]
    if reward_kind.startswith("episode"):
        ends = [i for i, r in enumerate(recs) if r["episode_done"]]
        assert ends and ends[-1] == len(recs) - 1
        lo, written = 0, 0
        for e in ends:
            ep = recs[lo:e + 1]
            if reward_kind == "episode_list":
                written += w.add_episode(ep)
            else:
                keys = [k for k in ep[0] if k != "info"]
                written += w.add_episode({k: np.stack([np.asarray(r[k]) for r in ep]) for k in keys})
            lo = e + 1
        with pytest.raises(ValueError):
            w.add_episode(recs[:max(ends[0], 1)])          # no episode_done at the end
        if reward_kind != "episode_stacked_host_fn":
            assert written == int(g["ring_len"])           # (incl. the n-step wrapper's one-shot _pop record)
    else:
        for r in recs:
            w.add(r)
    n = int(g["ring_len"])
    assert len(ring) == n
    got = ring[np.arange(n)]
    assert set(got) == set(g["out"])
    for k, v in g["out"].items():
        np.testing.assert_array_equal(got[k].cpu().numpy().reshape(v.shape), v, err_msg=k)
    random.seed(7)                          # the generator's seed for the read-time column (her_vmap.py:107-108)
    xp = HindsightVmapRead(ring).temporal_sample(starts=torch.as_tensor(g["read"]["starts"]))
    assert set(xp) == set(g["read"]["sample"])
    for k, v in g["read"]["sample"].items():
        np.testing.assert_array_equal(xp[k].cpu().numpy().reshape(v.shape), v, err_msg=k)



def test_her_vmap_kernels_match_restatement(dev):
    """fdql_episode_her_vmap / fdql_episode_mc_return_vmap against oracle.replay.vmap_* (her_vmap.py:30-45,
    nstep_return_vmap.py:61-74 restated from text; the reference file itself cannot run without jax)."""
    import ctypes as C
    from fastdeepqlearning_amd import _native as nat
    from fastdeepqlearning_amd.Replay.wrappers import SparseL2Reward
    from oracle import replay as orp
    lib = nat.load()
    rng = np.random.RandomState(2)
    n, g, K = 37, 3, 8
    ag = rng.uniform(-1, 1, (n, g)).astype(np.float32)
    ag[5] = ag[20]; ag[11] = ag[30]                      # revisits so some virtual goals are reached
    dg = np.tile(rng.uniform(-1, 1, (1, g)).astype(np.float32), (n, 1))
    dg[-1] = ag[-1]                                      # real goal reached at the end
    fn = SparseL2Reward(0.3, -1.0)
    reward = np.asarray([fn(ag[i], dg[i])[0] + 0.25 * (i % 3) for i in range(n)], np.float32)
    done = np.asarray([fn(ag[i], dg[i])[1] for i in range(n)], np.float32)
    idx = rng.randint(0, n, K).astype(np.int32)
    want_r, want_d = orp.vmap_virtual_episode(ag[idx], ag, dg, reward, done, fn)
    t = lambda a: torch.tensor(np.ascontiguousarray(a)).to(dev)
    vg, vr, vd = torch.empty(n, (K + 1) * g, device=dev), torch.empty(n, K + 1, device=dev), torch.empty(n, K + 1, device=dev)
    nf = fn.native()
    r_d, d_d, ag_d, dg_d, idx_d = t(reward), t(done), t(ag), t(dg), t(idx)     # keep the device inputs alive
    nat.check(lib.fdql_episode_her_vmap(nat.ptr(r_d), nat.ptr(d_d), nat.ptr(ag_d), nat.ptr(dg_d),
                                        C.c_void_p(idx_d.data_ptr()), n, g, K, C.byref(nf), nat.ptr(vg), nat.ptr(vr),
                                        nat.ptr(vd), nat.current_stream()))
    vr, vd, vg = vr.cpu().numpy(), vd.cpu().numpy(), vg.cpu().numpy().reshape(n, K + 1, g)
    np.testing.assert_allclose(vr[:, :K], want_r.T, rtol=0, atol=1e-6)
    np.testing.assert_array_equal(vd[:, :K] != 0, want_d.T)
    np.testing.assert_array_equal(vr[:, K], reward)
    np.testing.assert_array_equal(vd[:, K], done)
    np.testing.assert_array_equal(vg[:, :K], np.broadcast_to(ag[idx], (n, K, g)))
    np.testing.assert_array_equal(vg[:, K], dg)
    # per-column return, quirk q10 (multiplies by done[i])
    out = torch.empty(n, K + 1, device=dev)
    vr_d, vd_d = t(vr), t(vd)
    nat.check(lib.fdql_episode_mc_return_vmap(nat.ptr(vr_d), nat.ptr(vd_d), nat.ptr(out), n, K + 1, 0.97, nat.current_stream()))
    out = out.cpu().numpy()
    for c in range(K + 1):
        want = orp.vmap_return_newest_first(vr[::-1, c], vd[::-1, c] != 0, 0.97)[::-1]
        np.testing.assert_array_equal(out[:, c], want)


def test_her_vmap_replay_stack(dev):
    """Replay.make(her_mode='vmap'): HindsightVmapWrite -> NStepReturnVmap -> ring; the read head replaces goal /
    reward / done / mc_return by ONE stored column for the whole batch (her_vmap.py:104-123), selected in the gather."""
    from fastdeepqlearning_amd import Replay
    from fastdeepqlearning_amd.Replay.wrappers import SparseL2Reward
    conf = _conf(dev, T=4, B=8)
    conf.use_HER, conf.her_mode, conf.num_instances, conf.replay_size = True, "vmap", 1, 600
    read_heads, write_heads = Replay.make(conf, compute_reward=SparseL2Reward(0.3, -1.0))
    rng = np.random.RandomState(0)
    np.random.seed(0)
    for ep in range(5):
        dg = rng.uniform(-1, 1, 2).astype(np.float32)
        for i in range(40):
            ag = rng.uniform(-1, 1, 2).astype(np.float32)
            write_heads[0].add({"obs_1d": rng.standard_normal(5).astype(np.float32), "achieved_goal": ag, "desired_goal": dg,
                                "action": rng.uniform(-1, 1, 3).astype(np.float32), "reward": -1.0, "task_done": False,
                                "episode_done": i == 39, "episode_step": i, "info": {}})
    rb = read_heads[0].replay_buffer
    assert len(rb) == 200
    K1 = 33
    starts = torch.arange(8) * 20
    full = rb.temporal_sample(starts=starts)
    assert tuple(full["virtual_goals"].shape) == (4, 8, K1, 2) and tuple(full["virtual_mc_return"].shape) == (4, 8, K1)
    for idx in (0, 7, K1 - 1):
        sel = rb.temporal_sample_select({"virtual_goals": (idx * 2, 2), "virtual_rewards": (idx, 1), "virtual_dones": (idx, 1),
                                         "virtual_mc_return": (idx, 1), "desired_goal": None, "reward": None,
                                         "task_done": None}, starts=starts)
        assert "desired_goal" not in sel and "reward" not in sel
        assert torch.equal(sel["virtual_goals"], full["virtual_goals"][:, :, idx])
        assert torch.equal(sel["virtual_rewards"], full["virtual_rewards"][:, :, idx, None])
        assert torch.equal(sel["virtual_dones"], full["virtual_dones"][:, :, idx, None])
        assert torch.equal(sel["virtual_mc_return"], full["virtual_mc_return"][:, :, idx, None])
        assert torch.equal(sel["obs_1d"], full["obs_1d"])
    # the real column (index K) carries the environment's own goal / reward
    assert torch.equal(full["virtual_goals"][:, :, K1 - 1], full["desired_goal"])
    assert torch.equal(full["virtual_rewards"][:, :, K1 - 1, None], full["reward"])
    xp = read_heads[0].temporal_sample()
    assert set(xp) == {"obs_1d", "achieved_goal", "desired_goal", "action", "reward", "task_done", "episode_done",
                       "episode_step", "mc_return"}
    assert tuple(xp["desired_goal"].shape) == (4, 8, 2) and tuple(xp["mc_return"].shape) == (4, 8, 1)


# --------------------------------------------------------------------------- C-ABI error behaviour
def test_c_abi_error_paths(dev):
    """Status codes + fdql_last_error(): bad config, update before bind, short workspace, oversample."""
    import ctypes as C
    from fastdeepqlearning_amd import _native as nat
    from fastdeepqlearning_amd.core import make_config, NativeRing
    lib = nat.load()
    h = C.c_void_p()
    bad = make_config(5, 3, 1, 8)                               # T = 1: no TD pair
    assert lib.fdql_agent_create(C.byref(h), C.byref(bad)) == nat.FDQL_EINVAL
    assert b"T >= 2" in lib.fdql_last_error()
    q6 = make_config(5, 3, 4, 8, n_critics=2, n_quantiles=2)    # int(0.2 * 4) == 0: the reference's [:-0] slice is empty (q6)
    assert lib.fdql_agent_create(C.byref(h), C.byref(q6)) == nat.FDQL_EINVAL
    ok = make_config(5, 3, 4, 8, n_critics=2, n_quantiles=5, latent=32, enc_features=32, enc_hidden=(32,), joint_hidden=(32,),
                     pi_hidden=(32,), critic_hidden=(32, 32))
    nat.check(lib.fdql_agent_create(C.byref(h), C.byref(ok)))
    b = nat.Batch()
    assert lib.fdql_agent_update(h, C.byref(b), None, None, 0, 0, None) == nat.FDQL_ESTATE      # not bound yet
    n = lib.fdql_agent_arena_floats(h, 0)
    bufs = [torch.zeros(n, device=dev) for _ in range(4)] + [torch.zeros(lib.fdql_agent_arena_floats(h, 1), device=dev),
                                                            torch.zeros(lib.fdql_agent_arena_floats(h, 2), device=dev)]
    ws = torch.zeros(1024, dtype=torch.uint8, device=dev)
    rc = lib.fdql_agent_bind(h, *[C.c_void_p(t.data_ptr()) for t in bufs], C.c_void_p(ws.data_ptr()), 1024)
    assert rc == nat.FDQL_EINVAL and b"workspace too small" in lib.fdql_last_error()
    lib.fdql_agent_destroy(h)
    ring = NativeRing(64, [2], dev)
    ring.add_rows(np.zeros((5, 2), np.float32))
    with pytest.raises(nat.OversampleError):
        ring.sample_windows(4, 2)
    with pytest.raises(nat.FdqlError):
        NativeRing(64, [0], dev)


def test_checkpoint_resume_is_exact(dev, tmp_path):
    """Save agent (weights + Adam moments + step + lagged alpha) and ring (contents, write position, length, sample
    counter) mid-run, rebuild both from disk, continue: the next steps are bit-identical to the uninterrupted run."""
    from fastdeepqlearning_amd import Agent, Replay
    conf = _conf(dev, T=4, B=16)
    conf.num_instances = 1
    conf.replay_size = 300          # wraps during the fill below
    read_heads, write_heads = Replay.make(conf)
    agent = Agent.make(conf)
    agent.enable_training(read_heads)
    rng = np.random.RandomState(3)
    for ep in range(5):
        for i in range(90):
            write_heads[0].add({"obs_1d": rng.standard_normal(5), "action": rng.uniform(-1, 1, 3).astype(np.float32),
                                "reward": float(rng.standard_normal()), "task_done": bool(rng.rand() < 0.02),
                                "episode_done": i == 89, "episode_step": i, "idx": 0})
    for _ in range(3):
        agent.train_step()
    agent.save(tmp_path)
    ring = read_heads[0]            # Replay.make: the read heads are the ring shards themselves
    ring.save(tmp_path / "ring0")
    for _ in range(2):
        agent.train_step()
    want = {k: v.clone() for k, v in agent.state_dict().items()}

    conf2 = _conf(dev, T=4, B=16)
    conf2.num_instances, conf2.replay_size = 1, 300
    read2, _ = Replay.make(conf2)
    ring2 = read2[0]
    ring2.load(tmp_path / "ring0")
    assert len(ring2) == len(ring) and ring2._ring.top == ring._ring.top
    agent2 = type(agent).load_from_file(tmp_path)
    agent2.enable_training(read2)
    for _ in range(2):
        agent2.train_step()
    for k, v in agent2.state_dict().items():
        assert torch.equal(v, want[k]), k


def test_gru_agent_life_cycle(dev):
    """EncoderConf.JoinerModeEnum.gru through the facade like franQ.Runner drives it: act() takes the carried
    agent_state and returns the next one (runner.py:103-106, 157), the replay stores it, train_step scans from
    the stored state (RnnLatentStateTrainMode.store)."""
    from fastdeepqlearning_amd import Agent, Replay
    conf = _conf(dev, T=4, B=16)
    conf.num_instances = 1
    ec = conf.encoder_conf
    ec.joiner_mode = ec.JoinerModeEnum.gru
    ec.rnn_latent_state_training_mode = ec.RnnLatentStateTrainMode.store
    try:
        read_heads, write_heads = Replay.make(conf)
        agent = Agent.make(conf)
        assert "encoder.joiner.weight_hh_l0" in agent.state_dict() and "encoder.hidden_state" in agent.state_dict()
        assert tuple(agent.state_dict()["encoder.joiner.weight_ih_l0"].shape) == (96, 32)
        h = agent.get_random_hidden()
        assert tuple(h.shape) == (32,)
        rng = np.random.RandomState(1)
        hidden = h.view(1, -1).to(dev)
        for ep in range(3):
            for i in range(60):
                obs = rng.standard_normal(5).astype(np.float32)
                action, hidden_next, info = agent.act({"obs_1d": torch.tensor(obs).view(1, -1), "agent_state": hidden,
                                                       "exploit_mask": torch.zeros(1, 1, dtype=torch.bool)})
                assert tuple(hidden_next.shape) == (1, 32) and not torch.equal(hidden_next, hidden)
                write_heads[0].add({"obs_1d": obs, "action": action[0].cpu().numpy(), "reward": float(rng.standard_normal()),
                                    "task_done": False, "episode_done": i == 59, "episode_step": i, "idx": 0,
                                    "agent_state": hidden[0].cpu().numpy()})
                hidden = hidden_next
        agent.enable_training(read_heads)      # after collection: in sync mode act() would train as soon as the ring is ready
        before = {k: v.clone() for k, v in agent.state_dict().items()}
        for _ in range(2):
            agent.train_step()
        sc = agent.native.scalars()
        assert np.isfinite(sc["loss"]) and sc["step"] == 2
        after = agent.state_dict()
        assert not torch.equal(before["encoder.joiner.weight_hh_l0"], after["encoder.joiner.weight_hh_l0"])
        assert torch.equal(before["encoder.hidden_state"], after["encoder.hidden_state"])   # unused in store mode
    finally:
        ec.joiner_mode = ec.JoinerModeEnum.feedforward      # EncoderConf attributes are class-level defaults
        ec.rnn_latent_state_training_mode = ec.RnnLatentStateTrainMode.zero


def test_pixel_agent_life_cycle(dev):
    """obs_2d observation space (BASELINE config 5 shape of problem, small): frames enter the replay as uint8 arrays and
    stay one byte per pixel in HBM, act() runs the conv encoder on the current frames, train_step trains it."""
    from fastdeepqlearning_amd import Agent, Replay
    conf = _conf(dev, T=3, B=8)
    conf.num_instances = 1
    conf.obs_space = _Space(spaces={"obs_2d": _Space(shape=(2, 12, 12))})
    conf.action_space = _Space(n=4)
    conf.discrete = True
    conf.encoder_conf.conv_layers = ((8, 4, 2), (8, 3, 1))
    read_heads, write_heads = Replay.make(conf)
    agent = Agent.make(conf)
    sd = agent.state_dict()
    assert tuple(sd["encoder.visible_layer_encoders.obs_2d.conv.0.weight"].shape) == (8, 2 * 4 * 4)
    assert tuple(sd["encoder.visible_layer_encoders.obs_2d.conv.1.weight"].shape) == (8, 8 * 3 * 3)
    assert tuple(sd["encoder.visible_layer_encoders.obs_1d.feature_extractor.0.0.weight"].shape) == (32, 8 * 3 * 3)
    rng = np.random.RandomState(2)
    for ep in range(3):
        for i in range(40):
            frame = rng.randint(0, 256, (2, 12, 12)).astype(np.uint8)
            action, hidden, info = agent.act({"obs_2d": torch.tensor(frame).unsqueeze(0),
                                              "exploit_mask": torch.zeros(1, 1, dtype=torch.bool)})
            assert hidden is None and action.dtype == torch.int64 and 0 <= int(action) < 4
            write_heads[0].add({"obs_2d": frame, "action": np.asarray([int(action)]), "reward": float(rng.standard_normal()),
                                "task_done": False, "episode_done": i == 39, "episode_step": i, "idx": 0})
    ring = read_heads[0]
    assert ring._dtypes[ring._keys.index("obs_2d")] == "u8"
    batch = ring.temporal_sample()
    assert tuple(batch["obs_2d"].shape) == (3, 8, 2, 12, 12) and batch["obs_2d"].dtype == torch.float32
    assert float(batch["obs_2d"].max()) > 1.0 and torch.equal(batch["obs_2d"], batch["obs_2d"].round())
    agent.enable_training(read_heads)
    before = {k: v.clone() for k, v in agent.state_dict().items()}
    for _ in range(2):
        agent.train_step()
    sc = agent.native.scalars()
    assert np.isfinite(sc["loss"]) and sc["step"] == 2
    assert not torch.equal(before["encoder.visible_layer_encoders.obs_2d.conv.0.weight"],
                           agent.state_dict()["encoder.visible_layer_encoders.obs_2d.conv.0.weight"])


def test_pixel_agent_reads_uint8_frames_from_the_ring_in_place(dev):
    """BASELINE config 5's frame stacks through the facade: a uint8 obs_2d space makes the agent take the frames as bytes
    (cfg.obs_2d_u8), enable_training() switches its shards to in-place reads - temporal_sample() hands out the ring's uint8
    block and an int32 slot per row instead of a gathered float32 batch - and the gradient equals the one an agent on float32
    frames (first layer on im2col + GEMM) forms from the same windows."""
    from fastdeepqlearning_amd import Agent, Replay
    from fastdeepqlearning_amd import _native as N

    class _U8Space(_Space):
        dtype = np.uint8

    def build(float_frames):
        conf = _conf(dev, T=3, B=4)
        conf.num_instances = 1
        conf.obs_space = _Space(spaces={"obs_2d": _U8Space(shape=(4, 84, 84))})
        conf.action_space = _Space(n=6)
        conf.discrete = True
        conf.encoder_conf.conv_layers = ((32, 8, 4), (64, 4, 2), (64, 3, 1))
        conf.pixel_frames_float32 = float_frames
        return conf

    conf = build(False)
    read_heads, write_heads = Replay.make(conf)
    agent = Agent.make(conf)
    assert agent.native.cfg.obs_2d_u8 == 1 and agent.native.conv_reads_ring()
    rng = np.random.RandomState(4)
    for i in range(40):
        write_heads[0].add({"obs_2d": rng.randint(0, 256, (4, 84, 84)).astype(np.uint8), "action": np.asarray([int(rng.randint(6))]),
                            "reward": float(rng.standard_normal()), "task_done": False, "episode_done": i == 39, "episode_step": i, "idx": 0})
    agent.enable_training(read_heads)
    loader = agent.replays[0]
    xp = loader.temporal_sample()
    ring = read_heads[0]
    assert xp["obs_2d"].dtype == torch.uint8 and tuple(xp["obs_2d"].shape) == (ring._maxlen, 4, 84, 84)
    assert xp["obs_2d_slots"].dtype == torch.int32 and tuple(xp["obs_2d_slots"].shape) == (3, 4)
    slots = xp["obs_2d_slots"].long()
    assert int(slots.max()) < len(ring) and torch.equal(slots[1:], slots[:-1] + 1)      # consecutive slots of a window (no wrap at this fill)
    M, A = 2 * 4, 6
    g = torch.Generator().manual_seed(1)
    nt, na = torch.rand(2, 4, A, generator=g).to(dev), torch.rand(2, 4, A, generator=g).to(dev)
    agent.native.update(xp, nt, na, phase=N.PHASE_GRAD)
    # the same windows as a gathered float32 batch, on an agent whose first layer takes float32 frames
    ref = Agent.make(build(True))
    assert ref.native.cfg.obs_2d_u8 == 0
    ref.load_state_dict(agent.state_dict())
    xp2 = {k: v for k, v in xp.items() if not k.startswith("obs_2d")}
    xp2["obs_2d"] = xp["obs_2d"][slots.reshape(-1)].float().view(3, 4, 4, 84, 84)
    ref.native.update(xp2, nt, na, phase=N.PHASE_GRAD)
    torch.cuda.synchronize(dev)
    for name, gv in agent.native.grad_views.items():
        r = ref.native.grad_views[name]
        scale = float(r.abs().max()) + 1e-30
        assert float((gv - r).abs().max()) / scale < 1e-4, name
    for _ in range(2):
        agent.train_step()
    sc = agent.native.scalars()
    assert np.isfinite(sc["loss"]) and agent.native.stats()["plans_built"] <= 4


def test_in_place_batches_follow_a_restored_ring(dev):
    """load_state_dict() builds a new HBM ring; the uint8 block, the slot / start buffers and the persistent
    sample buffers handed out before must go with the old one.  Sample in place, restore a checkpoint whose frames differ,
    sample again: the batch's block is the restored ring's (its frames at the sampled slots are the checkpoint's)."""
    from fastdeepqlearning_amd.Replay import ReplayMemory
    T, B, n, shape = 3, 4, 40, (2, 6, 6)

    def fill(shard, offset):
        for i in range(n):
            shard.add({"obs_2d": np.full(shape, (i + offset) % 256, np.uint8), "action": np.asarray([float(i)], np.float32),
                       "reward": 0.0, "task_done": False, "episode_done": False, "episode_step": i})

    a = ReplayMemory(64, B, T, device=dev, sample_buffers=3)
    fill(a, 100)
    sd = a.state_dict()
    b = ReplayMemory(64, B, T, device=dev, sample_buffers=3)
    fill(b, 0)
    b.enable_in_place(("obs_2d",))
    xp0 = b.temporal_sample()
    old_block = xp0["obs_2d"]
    s0 = xp0["obs_2d_slots"].long()
    assert torch.equal(old_block[s0.reshape(-1)][:, 0, 0, 0].float().view(T, B), xp0["action"].view(T, B))
    b.load_state_dict(sd)
    xp1 = b.temporal_sample()
    assert xp1["obs_2d"].data_ptr() != old_block.data_ptr(), "the batch still carries the replaced ring's block"
    s1 = xp1["obs_2d_slots"].long()
    frames = xp1["obs_2d"][s1.reshape(-1)][:, 0, 0, 0].float().view(T, B)
    assert torch.equal(frames, (xp1["action"].view(T, B) + 100) % 256), "frames do not belong to the sampled records"
    # no float32 buffer exists for the key read in place
    assert all(entry[b._keys.index("obs_2d")] is None for entry in b._pool)


def test_launch_mode_auto_calibrates_and_keeps_training(dev):
    """conf.launch_mode = "auto" (round 6): a launch-bound plan (temporal_len 2) is timed eagerly and as a hipGraph replay after
    20 steps - ordinary training steps, counted like any other - and the faster way stays; the weights keep moving and the
    loss stays finite whichever was chosen (replay is bit-identical to the eager list: test_graph_replay_matches_eager_launches)."""
    from fastdeepqlearning_amd import Agent, Replay
    conf = _conf(dev, T=2, B=64)
    conf.num_instances = 1
    conf.launch_mode = "auto"
    read_heads, write_heads = Replay.make(conf)
    rng = np.random.RandomState(0)
    for i in range(400):
        write_heads[0].add({"obs_1d": rng.standard_normal(5).astype(np.float32),
                            "action": rng.uniform(-1, 1, 3).astype(np.float32), "reward": float(rng.standard_normal()),
                            "task_done": False, "episode_done": i % 50 == 49, "episode_step": i % 50, "idx": 0})
    agent = Agent.make(conf)
    agent.enable_training(read_heads)
    for _ in range(19):
        agent.train_step()
    assert agent.launch_mode_choice is None and agent.iteration == 19
    w0 = agent.state_dict()["actor_critic.actor.head.weight"].clone()
    agent.train_step()          # step 20 -> still plain
    agent.train_step()          # iteration 20 reached: calibrates (2 rounds x 2 modes x (9 + 30) steps), then steps once more
    ch = agent.launch_mode_choice
    assert ch is not None and ch["chosen"] in ("eager", "graph") and agent.native.launch_mode == ch["chosen"]
    assert agent.iteration == 21 + 2 * 2 * 39
    for _ in range(5):
        agent.train_step()
    sc = agent.native.scalars()
    assert np.isfinite(sc["loss"]) and not torch.equal(w0, agent.state_dict()["actor_critic.actor.head.weight"])
    if ch["chosen"] == "graph":
        assert agent.native.stats()["graph_launches"] > 0
