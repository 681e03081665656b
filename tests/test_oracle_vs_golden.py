"""Pin the CPU oracle against vectors produced by running the reference itself
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from golden_io import load, spec_from_case, UPDATE_CASES, ACT_CASES
from oracle import replay as orp
from oracle import update as oup


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30)) if a.size else 0.0


# --------------------------------------------------------------------------- ring
@pytest.mark.parametrize("case", ["wrap8", "nowrap50", "wrap50"])
def test_ring_matches_reference(case):
    g = load("ring")[case]
    r = orp.RingOracle(int(g["maxlen"]), int(g["B"]), int(g["T"]))
    rows = g["rows"]
    n = rows["reward"].shape[0]
    for i in range(n):
        r.add({"obs_1d": rows["obs_1d"][i], "action": rows["action"][i], "reward": float(rows["reward"][i]),
               "task_done": bool(rows["task_done"][i]), "episode_done": bool(rows["episode_done"][i]),
               "episode_step": int(rows["episode_step"][i]), "idx": int(rows["idx"][i])})
    assert len(r) == int(g["len"]) and r.top == int(g["top"])
    for k, v in g["memory"].items():
        assert r.memory[k].dtype == v.dtype
        np.testing.assert_array_equal(r.memory[k], v)
    win = r.temporal_sample(starts=g["starts"])
    for k, v in g["window"].items():
        np.testing.assert_array_equal(win[k], v)
    flat = r.sample(idx=g["flat_idx"])
    for k, v in g["flat"].items():
        np.testing.assert_array_equal(flat[k], v)


def test_ring_len_quirk_q1():
    r = orp.RingOracle(8, 2, 2)
    for i in range(20):
        r.add({"x": np.zeros(1, np.float32)})
    assert len(r) == 7


def test_ring_oversample_thresholds():
    g = load("ring")["oversample"]
    r = orp.RingOracle(100, int(g["B"]), int(g["T"]))
    got = []
    for i in range(len(g["ok_after_n_adds"])):
        r.add({"x": np.zeros(1, np.float32)})
        try:
            r.temporal_sample()
            got.append(1)
        except orp.OversampleError:
            got.append(0)
    np.testing.assert_array_equal(got, g["ok_after_n_adds"])


# --------------------------------------------------------------------------- n-step
class Sink:
    def __init__(self):
        self.rows = []

    def add(self, d):
        self.rows.append(dict(d))

    def stacked(self):
        keys = sorted(self.rows[0].keys())
        return {k: np.stack([np.asarray(r[k]).reshape(-1) for r in self.rows]) for k in keys}


@pytest.mark.parametrize("case", ["sparse_1000", "dense_two_eps", "pop_quirk", "single_step"])
def test_nstep_matches_reference(case):
    g = load("nstep")[case]
    sink = Sink()
    w = orp.NStepOracle(sink, int(g["n_step"]), float(g["gamma"]))
    inp = g["in"]
    for i in range(inp["reward"].shape[0]):
        w.add({"reward": float(inp["reward"][i, 0]), "episode_done": bool(inp["episode_done"][i, 0]),
               "episode_step": int(inp["episode_step"][i, 0]), "obs_1d": inp["obs_1d"][i]})
    out = sink.stacked()
    assert out["reward"].shape == g["out"]["reward"].shape
    for k, v in g["out"].items():
        np.testing.assert_array_equal(np.asarray(out[k], v.dtype), v, err_msg=k)


def _ulps(a, b):
    """distance in float32 units in the last place of b"""
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    return np.abs(a.astype(np.float64) - b.astype(np.float64)) / np.spacing(np.abs(b)).astype(np.float64)


def test_nstep_known_answers_and_numba_arithmetic():
    """Known-answer episodes for the n-step scan (nstep_return.py:60-72).  (a) gamma = 0.5 and rewards of powers of two: every
    partial sum is exactly representable, so the float32 loop (the goldens' identity-njit arithmetic), numba's float64-product
    arithmetic and the closed form agree BIT FOR BIT.  (b) gamma = 0.99 over 1000 dense random rewards: the two arithmetics
    differ by rounding (and by float32(0.99) vs the double 0.99): the distance is reported and bounded - the claim for this
    column is "within 1e-5 of the reference under real numba", not bit-exactness."""
    n = 20
    ones = np.ones(n, np.float32)
    closed = (2.0 - 0.5 ** np.arange(n)).astype(np.float32)              # newest first: 1, 1.5, 1.75, ...
    assert np.array_equal(orp.discounted_return_newest_first(ones, 0.5), closed)
    assert np.array_equal(orp.discounted_return_numba_arithmetic(ones, 0.5), closed)
    sparse = np.zeros(n, np.float32)
    sparse[0] = 8.0                                                      # terminal reward, newest record
    want = (8.0 * 0.5 ** np.arange(n)).astype(np.float32)
    assert np.array_equal(orp.discounted_return_newest_first(sparse, 0.5), want)
    assert np.array_equal(orp.discounted_return_numba_arithmetic(sparse, 0.5), want)
    rng = np.random.RandomState(3)
    r = rng.standard_normal(1000).astype(np.float32)
    f32, nb = orp.discounted_return_newest_first(r, 0.99), orp.discounted_return_numba_arithmetic(r, 0.99)
    rel = np.abs(f32.astype(np.float64) - nb) / np.max(np.abs(nb))
    assert rel.max() < 1e-5, rel.max()
    # and the reference's own test signal (tests/test_replays.py:16-33: reward 1 at the last step): both within its np.allclose
    last = np.zeros(1000, np.float32)
    last[0] = 1.0
    for fn in (orp.discounted_return_newest_first, orp.discounted_return_numba_arithmetic):
        assert np.allclose(fn(last, 0.99), 0.99 ** np.arange(1000))


def test_nstep_reference_own_test():
    """tests/test_replays.py:16-33 restated: mc_return == 0.99 ** (999 - step)."""
    sink = Sink()
    w = orp.NStepOracle(sink, 1000, 0.99)
    for i in range(1000):
        w.add({"reward": float(i == 999), "episode_done": i == 999, "step": i})
    out = sink.stacked()
    assert np.allclose(out["mc_return"], 0.99 ** (999 - out["step"]))


# --------------------------------------------------------------------------- HER
def l2_sparse_reward(ag, dg, thr=0.25):
    d = np.linalg.norm(np.asarray(ag, np.float32) - np.asarray(dg, np.float32))
    reward = np.float32(-1.0) if d > thr else np.float32(0.0)
    return reward, bool(reward == 0)


@pytest.mark.parametrize("case", ["final", "random", "final_nstep"])
def test_her_matches_reference(case):
    import random
    g = load("her")[case]
    sink = Sink()
    inner = orp.NStepOracle(sink, 1000, float(g["gamma"])) if int(g["nstep"]) else sink
    random.seed(3)
    w = orp.HerOracle(inner, l2_sparse_reward, mode=str(g["mode"]))
    inp = g["in"]
    for i in range(inp["reward"].shape[0]):
        w.add({"obs_1d": inp["obs_1d"][i], "achieved_goal": inp["achieved_goal"][i],
               "desired_goal": inp["desired_goal"][i], "action": inp["action"][i],
               "reward": float(inp["reward"][i, 0]), "task_done": bool(inp["task_done"][i, 0]),
               "episode_done": bool(inp["episode_done"][i, 0]), "episode_step": int(inp["episode_step"][i, 0]),
               "info": {}})
    out = sink.stacked()
    for k, v in g["out"].items():
        assert out[k].shape == v.shape, k
        np.testing.assert_allclose(np.asarray(out[k], np.float64), np.asarray(v, np.float64), rtol=0, atol=1e-6,
                                   err_msg=k)


def _vmap_records(inp):
    return [{"obs_1d": inp["obs_1d"][i], "achieved_goal": inp["achieved_goal"][i], "desired_goal": inp["desired_goal"][i],
             "action": inp["action"][i], "reward": float(inp["reward"][i, 0]), "task_done": bool(inp["task_done"][i, 0]),
             "episode_done": bool(inp["episode_done"][i, 0]), "episode_step": int(inp["episode_step"][i, 0]), "info": {}}
            for i in range(inp["reward"].shape[0])]


@pytest.mark.parametrize("case", ["k4", "k32_pop"])
def test_her_vmap_matches_reference(case):
    """her_vmap.py + nstep_return_vmap.py ran on the jax stand-in (shim-pinned): K relabelled columns + the real
    one, per-column returns (q10), the one-shot _pop duplicate (q3), and the read-time single column (q11)."""
    g = load("her_vmap")[case]
    K, T, B = int(g["K"]), int(g["T"]), int(g["B"])
    draws = iter(g["goal_idx_newest_first"])
    sink = Sink()
    w = orp.VmapWriteOracle(sink, l2_sparse_reward, K, int(g["n_step"]), float(g["gamma"]), draw=lambda n, k: next(draws))
    for row in _vmap_records(g["in"]):
        w.add(row)
    out = sink.stacked()
    assert set(out) == set(g["out"])
    for k, v in g["out"].items():
        np.testing.assert_array_equal(np.asarray(out[k], np.float32).reshape(v.shape), v, err_msg=k)
    # read side through the numpy ring
    ring = orp.RingOracle(64, B, T)
    for r in sink.rows:
        ring.add({k: (np.asarray(v) if np.asarray(v).ndim else v) for k, v in r.items()})
    assert len(ring) == int(g["ring_len"])
    sample = orp.vmap_read_select({k: v.reshape(v.shape[:2] + ((K + 1, -1) if k == "virtual_goals" else v.shape[2:]))
                                   for k, v in ring.temporal_sample(starts=g["read"]["starts"]).items()},
                                  int(g["read"]["column"]))
    assert set(sample) == set(g["read"]["sample"])
    for k, v in g["read"]["sample"].items():
        np.testing.assert_array_equal(np.asarray(sample[k], np.float32).reshape(v.shape), v, err_msg=k)


# --------------------------------------------------------------------------- update
TOL = 2e-5  # tensor-normalised max error; both sides are fp32 CPU with different op order


@pytest.mark.parametrize("case", UPDATE_CASES)
def test_update_matches_reference(case):
    torch.set_num_threads(1)
    g = load("update_" + case)
    spec = spec_from_case(g["case"])
    params = {k: torch.tensor(v) for k, v in g["init"].items()}
    assert int(g["n_trainable"]) == sum(params[n].numel() for n in oup.trainable_names(spec))
    st = oup.new_state(spec, params)
    n_steps = len([k for k in g if k.startswith("step")])
    for s in range(n_steps):
        rec = g[f"step{s}"]
        xp = {k: torch.tensor(v) for k, v in rec["batch"].items()}
        assert abs(st.alpha - float(rec["alpha_in"])) <= 1e-6 * abs(st.alpha)
        loss, aux = oup.train_step(st, spec, xp, torch.tensor(rec["noise_target"]), torch.tensor(rec["noise_actor"]))
        assert rel_err(loss, rec["loss"]) < TOL, (s, float(loss), float(rec["loss"]))
        for name in ["state", "next_action", "next_log_pi", "next_z", "q_pred", "pi", "log_pi", "q_loss",
                     "pi_loss", "alpha_loss", "is_contiguous"]:
            assert rel_err(aux[name].detach(), rec[name]) < TOL, (s, name)
        if "grad" in rec:
            for n, gr in rec["grad"].items():
                assert rel_err(aux["grad"][n], gr) < 5e-5, (s, n, rel_err(aux["grad"][n], gr))
        if "after" in rec:
            for n, v in rec["after"].items():
                assert rel_err(st.params[n], v) < TOL, (s, n, rel_err(st.params[n], v))
        if "adam_m" in rec:
            for n, v in rec["adam_m"].items():
                assert rel_err(st.adam_m[n], v) < 5e-5, (s, n)
            for n, v in rec["adam_v"].items():
                assert rel_err(st.adam_v[n], v) < 5e-5, (s, n)


@pytest.mark.parametrize("case", ACT_CASES)
def test_act_matches_reference(case):
    """oracle.update.act == franQ DeepQLearning.act (deepQlearning.py:155-187) on the tapped noise draw."""
    torch.set_num_threads(1)
    g = load("act_" + case)
    spec = spec_from_case(g["case"])
    params = {k: torch.tensor(v) for k, v in g["init"].items()}
    xp = {k: torch.tensor(v) for k, v in g["xp"].items()}
    res = oup.act(params, spec, xp, torch.tensor(g["noise"]))
    action, logp, explore, exploit = res[:4]
    if spec.gru:    # the GRU joiner also returns the next hidden state (deepQlearning.py:166, 187)
        np.testing.assert_allclose(res[4].numpy(), g["hidden_state"], rtol=1e-6, atol=1e-7)
    if spec.discrete:   # integer actions: exact
        for got, key in ((action, "action"), (explore, "explore_action"), (exploit, "exploit_action")):
            assert np.array_equal(got.numpy(), g[key]), key
        assert set(np.unique(g["xp"]["exploit_mask"])) == {False, True}
    else:
        for got, key in ((action, "action"), (explore, "explore_action"), (exploit, "exploit_action")):
            np.testing.assert_allclose(got.numpy(), g[key], rtol=1e-6, atol=1e-7, err_msg=key)
    # log_prob is a sum of A terms of size O(1..10) that partly cancel: 1e-5 of the terms, not of the sum
    np.testing.assert_allclose(logp.numpy(), g["log_prob"], rtol=1e-5, atol=2e-5)
