"""CPU-only checks: the C-ABI library loads and exports every symbol include/fdql.h declares,
struct mirrors match, the conf translation, loud failure without a GPU, and the data-parallel
identity (world_size 2 over gloo) that the multi-GPU path relies on."""
import os
import re
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported_and_bound():
    from fastdeepqlearning_amd import _native as nat
    hdr = open(os.path.join(ROOT, "include", "fdql.h")).read()
    declared = set(re.findall(r"\b(fdql_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"fdql_agent_config_t", "fdql_batch_t"}
    lib = nat.load()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/fdql.h but not exported"
        assert name in nat.SIGNATURES, f"{name} has no ctypes signature"
    assert set(nat.SIGNATURES) == declared
    assert lib.fdql_version() >= 1


def test_product_path_has_no_cpu_fallback():
    """Without a GPU the ring / agent constructors must raise, not silently run on the host."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from fastdeepqlearning_amd.core import NativeRing
    with pytest.raises(Exception):
        NativeRing(16, [1], "cuda:0")
    import fastdeepqlearning_amd
    src = []
    for dp, _, fs in os.walk(os.path.dirname(fastdeepqlearning_amd.__file__)):
        for f in fs:
            if f.endswith(".py"):
                src.append(open(os.path.join(dp, f)).read())
    assert not any(re.search(r"^\s*(from|import)\s+oracle\b", s, re.M) for s in src), "product code imports the oracle"


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    from fastdeepqlearning_amd import _native as nat
    monkeypatch.setattr(nat, "_lib", None)
    monkeypatch.setattr(nat, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(nat.NativeLibraryMissing):
        nat.load()


class _Space:
    def __init__(self, shape=None, n=None, spaces=None):
        if shape is not None:
            self.shape = tuple(shape)
        if n is not None:
            self.n = n
        if spaces is not None:
            self.spaces = spaces


def test_conf_translation_matches_reference_defaults():
    from fastdeepqlearning_amd.Agent import AgentConf
    from fastdeepqlearning_amd.Agent.deepQlearning import native_config_from_conf
    conf = AgentConf()
    conf.obs_space = _Space(spaces={"obs_1d": _Space(shape=(28,)), "achieved_goal": _Space(shape=(10,)),
                                    "desired_goal": _Space(shape=(10,))})
    conf.action_space = _Space(shape=(6,))
    conf.discrete = False
    conf.num_critics, conf.num_q_predictions = 5, 2
    c = native_config_from_conf(conf)
    assert (c.obs_dim, c.goal_dim, c.act_dim) == (28, 10, 6)
    assert (c.n_critics, c.n_quantiles, c.latent, c.enc_features) == (5, 2, 256, 256)
    assert list(c.critic_hidden)[:c.n_critic_hidden] == [256, 256] and list(c.pi_hidden)[:c.n_pi_hidden] == [256]
    assert (c.T, c.B) == (50, 256)
    assert c.gamma == 0.99 and c.tau == 0.05 and c.lr == 3e-4 and c.beta2 == 0.999 and c.init_log_alpha == -2.0
    assert c.distributional == 1 and c.use_lowerbound == 1 and c.use_max_entropy == 1
    # attribute == item access (franQ/common_utils.py:59-67)
    assert conf["gamma"] == conf.gamma
    # GRU joiner + latent-state training mode (encoder.py:40-42, 78-94)
    conf.encoder_conf.joiner_mode = conf.encoder_conf.JoinerModeEnum.gru
    conf.encoder_conf.rnn_latent_state_training_mode = conf.encoder_conf.RnnLatentStateTrainMode.learned
    c = native_config_from_conf(conf)
    assert c.joiner_gru == 1 and c.gru_state_mode == 2
    conf.encoder_conf.use_burn_in = True
    assert native_config_from_conf(conf).burn_in_steps == int(50 * 0.2)
    conf.use_bootstrap_minibatch_nstep = True      # only defined for SAC-min with lower bounds in the reference
    with pytest.raises(ValueError):
        native_config_from_conf(conf)


def test_replay_make_selects_the_reference_wrapper_stacks():
    """franQ/Replay/__init__.py:20-36: which write / read wrappers the conf flags select (no ring is touched:
    shards allocate lazily)."""
    from fastdeepqlearning_amd import Replay
    from fastdeepqlearning_amd.Agent import AgentConf
    from fastdeepqlearning_amd.Replay import wrappers as W

    def chain(head):
        names = []
        while hasattr(head, "replay_buffer"):
            names.append(type(head).__name__)
            head = head.replay_buffer
        return names + [type(head).__name__]

    conf = AgentConf()
    conf.training_device = "cuda:0"
    conf.num_instances = 2
    r, w = Replay.make(conf)
    assert len(r) == len(w) == 2 and chain(w[0]) == ["NStepReturn", "AsyncReplayMemory"] and r[0] is w[0].replay_buffer
    conf.use_squashed_rewards = True
    assert chain(Replay.make(conf)[1][0]) == ["SquashRewards", "NStepReturn", "AsyncReplayMemory"]
    conf.use_HER = True                       # squash is dropped under HER (:28)
    with pytest.raises(KeyError):
        Replay.make(conf)
    fn = W.SparseL2Reward(0.05)
    assert chain(Replay.make(conf, compute_reward=fn)[1][0]) == ["HindsightNStepReplay", "NStepReturn", "AsyncReplayMemory"]
    conf.her_mode = "vmap"
    r, w = Replay.make(conf, compute_reward=fn)
    assert chain(w[0]) == ["HindsightVmapWrite", "NStepReturnVmap", "AsyncReplayMemory"] and chain(r[0])[0] == "HindsightVmapRead"
    conf.use_nStep_lowerbounds = False
    assert chain(Replay.make(conf, compute_reward=fn)[1][0]) == ["HindsightVmapWrite", "AsyncReplayMemory"]


def test_squash_rewards_known_answers():
    """squash_rewards.py:5-8: h(x) = sign(x)(sqrt(|x|+1)-1) + 0.01 x, applied to the record's reward only."""
    from fastdeepqlearning_amd.Replay.wrappers import SquashRewards

    class Sink:
        def __init__(self):
            self.rows = []

        def add(self, d):
            self.rows.append(d)

    sink = Sink()
    w = SquashRewards(sink)
    xs = [0.0, 3.0, -3.0, 8.0, -0.44, 1e4]
    for x in xs:
        w.add({"reward": x, "obs_1d": np.ones(2)})
    want = [0.0, 1.03, -1.03, 2.08, -(np.sqrt(1.44) - 1) - 0.0044, np.sqrt(10001.0) - 1 + 100.0]
    np.testing.assert_allclose([r["reward"] for r in sink.rows], want, rtol=1e-12, atol=1e-15)
    assert all(np.array_equal(r["obs_1d"], np.ones(2)) for r in sink.rows)


def test_her_host_relabel_matches_golden_with_callable_reward():
    """A plain Python compute_reward (env code) takes the host branch of the HER wrapper; its
    relabel arithmetic is checked against the reference's emitted sequence."""
    from golden_io import load
    from fastdeepqlearning_amd.Replay.wrappers.her import HindsightNStepReplay

    g = load("her")["final"]

    def reward(ag, dg, thr=0.25):
        d = np.linalg.norm(np.asarray(ag, np.float32) - np.asarray(dg, np.float32))
        r = np.float32(-1.0) if d > thr else np.float32(0.0)
        return r, bool(r == 0)

    class Sink:
        device = "cpu"

        def __init__(self):
            self.rows = []

        def add(self, d):
            self.rows.append(dict(d))

    sink = Sink()
    w = HindsightNStepReplay(sink, reward, mode="final", device="cpu")
    inp = g["in"]
    for i in range(inp["reward"].shape[0]):
        w.add({"obs_1d": inp["obs_1d"][i], "achieved_goal": inp["achieved_goal"][i], "desired_goal": inp["desired_goal"][i],
               "action": inp["action"][i], "reward": float(inp["reward"][i, 0]), "task_done": bool(inp["task_done"][i, 0]),
               "episode_done": bool(inp["episode_done"][i, 0]), "episode_step": int(inp["episode_step"][i, 0]), "info": {}})
    for k, v in g["out"].items():
        got = np.stack([np.asarray(r[k], np.float64).reshape(-1) for r in sink.rows])
        np.testing.assert_allclose(got, np.asarray(v, np.float64), rtol=0, atol=1e-6, err_msg=k)


# --------------------------------------------------------------------------- data parallel (gloo, world_size 2)
def _dp_worker(rank, world, port, ret):
    import torch.distributed as dist
    from oracle import update as oup
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    T, Bg = 4, 8
    Bl = Bg // world
    kw = dict(obs=5, act=3, C=3, Q=4, latent=16, enc_features=16, enc_hidden=(16,), joint_hidden=(16,), pi_hidden=(16,),
              critic_hidden=(16, 16), T=T)
    g = torch.Generator().manual_seed(0)
    xp = {"obs_1d": torch.randn(T, Bg, 5, generator=g), "action": torch.rand(T, Bg, 3, generator=g) * 2 - 1,
          "reward": torch.randn(T, Bg, 1, generator=g), "mc_return": torch.randn(T, Bg, 1, generator=g),
          "task_done": (torch.rand(T, Bg, 1, generator=g) < 0.2).float(),
          "episode_step": torch.arange(T).view(T, 1, 1).expand(T, Bg, 1).float().clone()}
    nt, na = torch.randn(T - 1, Bg, 3, generator=g), torch.randn(T - 1, Bg, 3, generator=g)
    spec_l = oup.Spec(B=Bl, world_size=world, **kw)
    params = oup.init_params(spec_l, seed=1)
    sl = slice(rank * Bl, (rank + 1) * Bl)
    st = oup.new_state(spec_l, params)
    _, aux = oup.train_step(st, spec_l, {k: v[:, sl] for k, v in xp.items()}, nt[:, sl], na[:, sl])
    names = oup.trainable_names(spec_l)
    flat = torch.cat([aux["grad"][n].reshape(-1) for n in names])
    dist.all_reduce(flat)                      # what the RCCL all-reduce of the gradient arena does
    if rank == 0:
        spec_g = oup.Spec(B=Bg, world_size=1, **kw)
        stg = oup.new_state(spec_g, params)
        _, auxg = oup.train_step(stg, spec_g, xp, nt, na)
        ref = torch.cat([auxg["grad"][n].reshape(-1) for n in names])
        ret["err"] = float((flat - ref).abs().max() / ref.abs().max())
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_gradient_identity_gloo():
    """Sharding B over ranks with the loss normalised by B*world and a SUM all-reduce reproduces
    the global-batch gradient.  Two CPU processes over gloo."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, ret)) for r in range(2)]
    [p.start() for p in procs]
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert ret["err"] < 1e-5, ret["err"]


def test_bench_picks_the_faster_data_parallel_launch_list(monkeypatch):
    """bench.py::dp_run_best (N > 1): both launch lists are built - the one-bucket one under FDQL_NO_BUCKETS, which must not
    leak into the environment - each is timed, the faster one is returned; a switch set by the caller takes the choice away."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    built = []

    class FakeAgent:
        def __init__(self, no_buckets):
            self.grads = torch.zeros(100)
            self._b = 100 if no_buckets else 34

    class FakeJob:
        def __init__(self, no_buckets):
            self.agent = FakeAgent(no_buckets)

    class FakeRun:
        ms = {False: 1.30, True: 1.10}   # two-bucket, one-bucket

        def __init__(self, *a, **k):
            self.nb = os.environ.get("FDQL_NO_BUCKETS") is not None
            self.job = FakeJob(self.nb)
            self.bucket = self.job.agent._b
            built.append(self.nb)

        def timed(self, steps, warmup):
            return self.ms[self.nb] * 1e-3 * steps

    monkeypatch.setattr(bench, "DPRun", FakeRun)
    monkeypatch.delenv("FDQL_NO_BUCKETS", raising=False)
    monkeypatch.delenv("FDQL_FORCE_BUCKETS", raising=False)
    run, plan = bench.dp_run_best({}, "cpu", 32, 50, 0, 2, "gloo", None, None)
    assert built == [False, True] and "FDQL_NO_BUCKETS" not in os.environ
    assert plan["chosen"] == "one_bucket" and run.nb and plan["calibration"]["ms_per_step"] == {"two_bucket": 1.3, "one_bucket": 1.1}
    FakeRun.ms = {False: 0.90, True: 1.10}
    run, plan = bench.dp_run_best({}, "cpu", 32, 50, 0, 2, "gloo", None, None)
    assert plan["chosen"] == "two_bucket" and not run.nb
    monkeypatch.setenv("FDQL_NO_BUCKETS", "1")   # the caller decided: one run, no calibration
    del built[:]
    run, plan = bench.dp_run_best({}, "cpu", 32, 50, 0, 2, "gloo", None, None)
    assert built == [True] and plan == {"chosen": "one_bucket", "calibration": None} and os.environ.get("FDQL_NO_BUCKETS") == "1"
