"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and against the
golden vectors generated from the reference.  Run with `pytest -m gpu` on an MI355X.

Tolerance (BASELINE.json north_star: "within 1e-5 rel fp32"): per tensor,
    max|x - ref| <= TOL * max|ref|        with TOL = 1e-5
for forward quantities; gradients / Adam moments / post-step weights are sums of ~1e4
fp32 products of mixed sign, for which both CPU formulations (reference vs oracle) already
differ by a few 1e-6 — they get 5e-5 on the same normalised measure.
"""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from golden_io import load, spec_from_case, ACT_CASES

pytestmark = pytest.mark.gpu

TOL = 1e-5
GTOL = 5e-5
_THREE_WAY_CACHE = {}   # test_gradient_parity_three_way_fp64: the CPU evaluations of a spec, shared by the cases that differ in switches only


def rel_err(a, b):
    a = a.detach().cpu().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, np.float64)
    b = b.detach().cpu().double().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30)) if a.size else 0.0


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "needs a GPU"
    return torch.device("cuda:0")


# --------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,N,K", [(128, 128, 16), (256, 256, 256), (200, 70, 33), (1000, 262, 258), (64, 6, 300),
                                   (513, 129, 17), (1024, 384, 272)])
@pytest.mark.parametrize("form", ["nt", "nn", "tn"])
@pytest.mark.parametrize("dense", [5, 3, 0, 7])     # 64x64, 64x128, 128x128 tiles; the small-batch kernel
def test_gemm_forms(dev, M, N, K, form, dense):
    from fastdeepqlearning_amd import _native as nat
    lib = nat.load()
    nat.check(lib.fdql_debug_set_gemm_dense_shape(dense))
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A = torch.randn(M, K, generator=g)
    Bm = torch.randn(K, N, generator=g)
    bias = torch.randn(N, generator=g)
    ref = (A.double() @ Bm.double() + bias.double()).float()
    if form == "nt":      # A[M,K] k-contiguous, B stored [N,K] k-contiguous
        a_mem, lda, akc = A.contiguous(), K, 1
        b_mem, ldb, bkc = Bm.t().contiguous(), K, 1
    elif form == "nn":    # A k-contiguous, B stored [K,N] (k-strided)
        a_mem, lda, akc = A.contiguous(), K, 1
        b_mem, ldb, bkc = Bm.contiguous(), N, 0
    else:                 # both k-strided: A stored [K,M]
        a_mem, lda, akc = A.t().contiguous(), M, 0
        b_mem, ldb, bkc = Bm.contiguous(), N, 0
    a_d, b_d, bias_d = a_mem.to(dev), b_mem.to(dev), bias.to(dev)
    c_d = torch.full((M, N), float("nan"), device=dev)
    nat.check(lib.fdql_test_gemm(nat.ptr(a_d), lda, akc, nat.ptr(b_d), ldb, bkc, nat.ptr(bias_d), nat.ptr(c_d), N,
                                 M, N, K, 0, None, 0, 1, nat.current_stream()))
    nat.check(lib.fdql_debug_set_gemm_dense_shape(5))
    assert rel_err(c_d, ref) < 2e-6


@pytest.mark.parametrize("dense", [5, 3])
def test_gemm_epilogues_and_ksplit(dev, dense):
    from fastdeepqlearning_amd import _native as nat
    lib = nat.load()
    nat.check(lib.fdql_debug_set_gemm_dense_shape(dense))
    g = torch.Generator().manual_seed(5)
    M, N, K = 300, 140, 1000
    A, Bm = torch.randn(M, K, generator=g), torch.randn(K, N, generator=g)
    gate = torch.randn(M, N, generator=g)
    a_d, b_d, gate_d = A.to(dev), Bm.to(dev), gate.to(dev)
    base = (A.double() @ Bm.double())
    c_d = torch.empty(M, N, device=dev)
    nat.check(lib.fdql_test_gemm(nat.ptr(a_d), K, 1, nat.ptr(b_d), N, 0, None, nat.ptr(c_d), N, M, N, K, 1, None, 0,
                                 1, nat.current_stream()))
    assert rel_err(c_d, torch.nn.functional.leaky_relu(base, 0.01).float()) < 2e-6
    nat.check(lib.fdql_test_gemm(nat.ptr(a_d), K, 1, nat.ptr(b_d), N, 0, None, nat.ptr(c_d), N, M, N, K, 2,
                                 nat.ptr(gate_d), N, 1, nat.current_stream()))
    assert rel_err(c_d, (base * torch.where(gate > 0, 1.0, 0.01).double()).float()) < 2e-6
    S = 5
    slabs = torch.full((S, M, N), float("nan"), device=dev)
    nat.check(lib.fdql_test_gemm(nat.ptr(a_d), K, 1, nat.ptr(b_d), N, 0, None, nat.ptr(slabs), N, M, N, K, 0, None, 0,
                                 S, nat.current_stream()))
    nat.check(lib.fdql_debug_set_gemm_dense_shape(5))
    assert rel_err(slabs.sum(0), base.float()) < 2e-6


# --------------------------------------------------------------------------- ring
def _pack(rows_by_key, keys):
    cols = [np.asarray(rows_by_key[k], np.float32).reshape(len(rows_by_key[k]), -1) for k in keys]
    return np.concatenate(cols, axis=1)


@pytest.mark.parametrize("case", ["wrap8", "nowrap50", "wrap50"])
def test_ring_matches_golden(dev, case):
    from fastdeepqlearning_amd.core import NativeRing
    g = load("ring")[case]
    keys = ["obs_1d", "action", "reward", "task_done", "episode_done", "episode_step", "idx"]
    dims = [np.asarray(g["rows"][k]).reshape(len(g["rows"]["reward"]), -1).shape[1] for k in keys]
    ring = NativeRing(int(g["maxlen"]), dims, dev)
    rows = _pack(g["rows"], keys)
    for i in range(rows.shape[0]):      # one add per record, like the reference's write path
        ring.add_rows(rows[i:i + 1])
    assert len(ring) == int(g["len"]) and ring.top == int(g["top"])
    T, B = int(g["T"]), int(g["B"])
    outs = ring.sample_windows(T, B, starts=torch.tensor(g["starts"]))
    for k, o in zip(keys, outs):
        want = np.asarray(g["window"][k]).astype(np.float32).reshape(T, B, -1)   # torch_dataloader.py:36 cast
        np.testing.assert_array_equal(o.cpu().numpy(), want, err_msg=k)
    outs = ring.sample_rows(B, idx=torch.tensor(g["flat_idx"]))
    for k, o in zip(keys, outs):
        want = np.asarray(g["flat"][k]).astype(np.float32).reshape(B, -1)
        np.testing.assert_array_equal(o.cpu().numpy(), want, err_msg=k)


def test_ring_oversample_and_len_quirk(dev):
    from fastdeepqlearning_amd.core import NativeRing
    from fastdeepqlearning_amd._native import OversampleError
    g = load("ring")["oversample"]
    ring = NativeRing(100, [1], dev)
    got = []
    for i in range(len(g["ok_after_n_adds"])):
        ring.add_rows(np.zeros((1, 1), np.float32))
        try:
            ring.sample_windows(int(g["T"]), int(g["B"]))
            got.append(1)
        except OversampleError:
            got.append(0)
    np.testing.assert_array_equal(got, g["ok_after_n_adds"])
    ring = NativeRing(8, [1], dev)
    for i in range(20):
        ring.add_rows(np.zeros((1, 1), np.float32))
    assert len(ring) == 7


@pytest.mark.parametrize("dims", [[1, 3, 17, 1], [64, 100, 376], [5]])
@pytest.mark.parametrize("T,B", [(50, 37), (2, 256), (7, 16)])
def test_ring_gather_vs_oracle_bulk(dev, dims, T, B):
    """Bulk device fill + wrap + Philox starts: every window equals the oracle's fancy-index gather."""
    from fastdeepqlearning_amd.core import NativeRing
    from oracle.replay import RingOracle
    maxlen, n = 1000, 2337
    rng = np.random.RandomState(1)
    rows = rng.standard_normal((n, sum(dims))).astype(np.float32)
    ring = NativeRing(maxlen, dims, dev)
    ring.add_rows(torch.tensor(rows[:700]).to(dev))     # device bulk append
    ring.add_rows(rows[700:])                           # host staged append (wraps twice)
    orc = RingOracle(maxlen, B, T)
    off = np.cumsum([0] + dims)
    for i in range(n):
        orc.add({f"k{j}": rows[i, off[j]:off[j + 1]] for j in range(len(dims))})
    assert len(ring) == len(orc) == maxlen - 1 and ring.top == orc.top
    outs, starts = ring.sample_windows(T, B, seed=11, counter=3, return_starts=True)
    starts = starts.cpu().numpy()
    assert starts.min() >= 0 and starts.max() < len(orc) - T
    want = orc.temporal_sample(starts=starts)
    for j, o in enumerate(outs):
        np.testing.assert_array_equal(o.cpu().numpy(), want[f"k{j}"])
    outs2, starts2 = ring.sample_windows(T, B, seed=11, counter=4, return_starts=True)
    assert not np.array_equal(starts2.cpu().numpy(), starts)


# --------------------------------------------------------------------------- episode transforms
@pytest.mark.parametrize("case", ["sparse_1000", "dense_two_eps", "single_step"])
def test_mc_return_matches_golden(dev, case):
    from fastdeepqlearning_amd import _native as nat
    lib = nat.load()
    g = load("nstep")[case]
    ep_lens = [int(x) for x in g["ep_lens"]]
    r_in = g["in"]["reward"][:, 0].astype(np.float32)
    want = g["out"]["mc_return"][:, 0]
    pos = 0
    for L in ep_lens:   # n_step >= episode length in these cases: one flush per episode
        r = torch.tensor(r_in[pos:pos + L]).to(dev)
        out = torch.empty(L, device=dev)
        nat.check(lib.fdql_episode_mc_return(nat.ptr(r), nat.ptr(out), L, float(g["gamma"]), nat.current_stream()))
        np.testing.assert_array_equal(out.cpu().numpy(), want[pos:pos + L])
        pos += L


def test_nstep_kernel_known_answers_and_numba_distance(dev):
    """k_mc_return against hand-derived known answers (gamma = 0.5, power-of-two rewards: exact in every arithmetic) and its
    distance to the arithmetic REAL numba gives the reference's loop (float64 product and sum, one rounding; the goldens are
    identity-njit float32 arithmetic, which the kernel matches bit for bit above).  The distance goes to the parity report."""
    from fastdeepqlearning_amd import _native as nat
    from oracle import replay as orp
    lib = nat.load()

    def kernel(r_oldest_first, gamma):
        r = torch.tensor(np.asarray(r_oldest_first, np.float32)).to(dev)
        out = torch.empty(r.numel(), device=dev)
        nat.check(lib.fdql_episode_mc_return(nat.ptr(r), nat.ptr(out), r.numel(), float(gamma), nat.current_stream()))
        return out.cpu().numpy()

    n = 20
    want = (2.0 - 0.5 ** np.arange(n)).astype(np.float32)[::-1]         # oldest first: reward-to-go of n, n - 1, ... ones
    assert np.array_equal(kernel(np.ones(n), 0.5), want)
    sparse = np.zeros(n, np.float32)
    sparse[-1] = 8.0
    assert np.array_equal(kernel(sparse, 0.5), (8.0 * 0.5 ** np.arange(n)).astype(np.float32)[::-1])
    rng = np.random.RandomState(3)
    lines = []
    for L, gamma in ((50, 0.99), (1000, 0.99), (1000, 0.999)):
        r = rng.standard_normal(L).astype(np.float32)
        got = kernel(r, gamma)
        nb = orp.discounted_return_numba_arithmetic(r[::-1], gamma)[::-1]
        f32 = orp.discounted_return_newest_first(r[::-1], gamma)[::-1]
        assert np.array_equal(got, f32)                                  # the goldens' arithmetic: bit for bit
        rel = float(np.max(np.abs(got.astype(np.float64) - nb)) / np.max(np.abs(nb)))
        ulps = np.abs(got.astype(np.float64) - nb) / np.spacing(np.abs(nb).astype(np.float32)).astype(np.float64)
        lines.append(f"n-step scan, {L} steps, gamma {gamma}: vs numba arithmetic max rel {rel:.2e}, max {ulps.max():.1f} ulp, "
                     f"{float(np.mean(ulps == 0)) * 100:.0f} % of the values identical")
        assert rel < 1e-5
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/parity_report.txt", "a") as f:
        f.write("== n-step return: kernel vs the arithmetic of real numba (nstep_return.py:69-72; goldens are identity-njit)\n")
        for ln in lines:
            f.write(ln + "\n")


@pytest.mark.parametrize("case", ["final", "random"])
def test_her_relabel_matches_golden(dev, case):
    from fastdeepqlearning_amd import _native as nat
    lib = nat.load()
    g = load("her")[case]
    inp, out = g["in"], g["out"]
    pos_in, pos_out = 0, 0
    for L in [int(x) for x in g["ep_lens"]]:
        sl = slice(pos_in, pos_in + L)
        hs = slice(pos_out + L, pos_out + 2 * L)          # hindsight copy follows the real episode
        goal = out["desired_goal"][hs][0]
        t = lambda a: torch.tensor(np.ascontiguousarray(a, np.float32)).to(dev)
        r, st, ag, dg, gl = t(inp["reward"][sl, 0]), t(inp["episode_step"][sl, 0]), t(inp["achieved_goal"][sl]), \
            t(inp["desired_goal"][sl]), t(goal)
        ro, do, so = (torch.empty(L, device=dev) for _ in range(3))
        fn = nat.RewardFn(0, float(g["thr"]), -1.0)
        nat.check(lib.fdql_episode_her_relabel(nat.ptr(r), nat.ptr(st), nat.ptr(ag), nat.ptr(dg), nat.ptr(gl), L, 2,
                                               C.byref(fn), nat.ptr(ro), nat.ptr(do), nat.ptr(so),
                                               nat.current_stream()))
        np.testing.assert_allclose(ro.cpu().numpy(), out["reward"][hs, 0], rtol=0, atol=1e-6)
        np.testing.assert_array_equal(do.cpu().numpy(), out["task_done"][hs, 0].astype(np.float32))
        np.testing.assert_array_equal(so.cpu().numpy(), out["episode_step"][hs, 0].astype(np.float32))
        pos_in += L
        pos_out += 2 * L


# --------------------------------------------------------------------------- update vs golden / oracle
#
# Per-quantity bounds.  All are "max|x - ref| <= TOL * max|ref|" except where the reference's own
# fp32 formula is ill-conditioned, in which case the bound adds the first-order effect of a
# 2-ulp difference in tanh (the CPU path's vector libm is itself only accurate to 1 ulp):
#   logp = ... - log(1 - a^2 + 1e-4),  a = tanh(x)   =>   |d logp| <= sum_j 2|a_j| / (1 - a_j^2 + 1e-4) * |da_j|
# and what follows from it (td target and q_loss via alpha * logp', pi_loss via alpha * logp,
# alpha_loss via log_alpha * logp).  Weights after Adam are compared three ways: the optimiser
# arithmetic exactly (recomputed on the CPU from the GPU's own gradient), the gradient itself
# against the reference, and the weights against the reference with the bound that the first
# steps of Adam allow (update = lr * g / (|g| + eps): a sign-like function of a gradient that
# carries fp32 summation noise where it is near zero).
ULP_HALF_TO_ONE = 5.96e-8


def logp_cond(act):
    a = np.abs(np.asarray(act, np.float64))
    return (2 * a / (1 - a * a + 1e-4) * 2 * ULP_HALF_TO_ONE).sum(-1, keepdims=True)


class Report:
    def __init__(self, tag):
        self.tag, self.rows, self.bad = tag, [], []

    def check(self, name, got, ref, tol=TOL, extra=None):
        """max over elements of (|got-ref| - extra) must be <= tol * max|ref|."""
        got = got.detach().cpu().double().numpy() if isinstance(got, torch.Tensor) else np.asarray(got, np.float64)
        ref = ref.detach().cpu().double().numpy() if isinstance(ref, torch.Tensor) else np.asarray(ref, np.float64)
        assert got.shape == ref.shape, (name, got.shape, ref.shape)
        scale = np.max(np.abs(ref)) + 1e-30
        d = np.abs(got - ref)
        raw = float(d.max() / scale) if d.size else 0.0
        if extra is not None:
            d = np.maximum(d - np.broadcast_to(extra, d.shape), 0)
        e = float(d.max() / scale) if d.size else 0.0
        self.rows.append((name, raw, e, tol))
        if not (e <= tol):
            self.bad.append((name, raw, e, tol))

    def check_frac(self, name, got, ref, tol, min_frac, hard_tol):
        """>= min_frac of the elements within tol * max|ref| and all within hard_tol * max|ref|.
        Used for gradients: LeakyReLU / relu(mc - q) kinks make them discontinuous in the
        activations, so a 1e-7 forward difference that lands on a kink moves one unit's row of a
        weight gradient by O(1/sqrt(rows)) of its size, in ANY two fp32 implementations."""
        got = got.detach().cpu().double().numpy() if isinstance(got, torch.Tensor) else np.asarray(got, np.float64)
        ref = ref.detach().cpu().double().numpy() if isinstance(ref, torch.Tensor) else np.asarray(ref, np.float64)
        assert got.shape == ref.shape, (name, got.shape, ref.shape)
        scale = np.max(np.abs(ref)) + 1e-30
        d = np.abs(got - ref) / scale
        frac = float(np.mean(d <= tol)) if d.size else 1.0
        mx = float(d.max()) if d.size else 0.0
        if d.size < 64:     # a fraction is meaningless for a handful of elements (biases of 4): bound the max instead
            frac, hard_tol = 1.0, min(hard_tol, 4 * tol)
        self.rows.append((name + f" frac_within={frac:.5f}", mx, mx, hard_tol))
        if frac < min_frac or mx > hard_tol:
            self.bad.append((name, frac, mx, hard_tol))

    def finish(self):
        import os
        os.makedirs("gpurun_out", exist_ok=True)
        with open("gpurun_out/parity_report.txt", "a") as f:
            f.write(f"== {self.tag}\n")
            for name, raw, e, tol in self.rows:
                f.write(f"{name:70s} raw={raw:.3e} beyond_cond={e:.3e} tol={tol:.1e}\n")
        assert not self.bad, (self.tag, self.bad[:8])


def _agent_for(spec, dev, **kw):
    from fastdeepqlearning_amd.core import NativeAgent, make_config
    cfg = make_config(spec.obs, spec.act, spec.T, spec.B, goal_dim=spec.goal, discrete=spec.discrete,
                      n_critics=spec.C, n_quantiles=spec.Q, latent=spec.latent, enc_features=spec.enc_features,
                      enc_hidden=spec.enc_hidden, joint_hidden=spec.joint_hidden, pi_hidden=spec.pi_hidden,
                      critic_hidden=spec.critic_hidden, distributional=spec.distributional,
                      use_lowerbound=spec.lowerbound, use_max_entropy=spec.max_entropy,
                      hard_updates=spec.hard_updates, gamma=spec.gamma, tau=spec.tau, lr=spec.lr,
                      init_log_alpha=spec.init_log_alpha, drop_frac=spec.drop, bootstrap_nstep=spec.bootstrap,
                      burn_in_steps=spec.burn_in, joiner_gru=bool(spec.gru), gru_state_mode=spec.gru or 0,
                      img=spec.img, conv=spec.conv, **kw)
    return NativeAgent(cfg, dev)


def _snapshot(ag):
    return ({k: v.clone() for k, v in ag.tensors.items()}, {k: v.clone() for k, v in ag.m_views.items()},
            {k: v.clone() for k, v in ag.v_views.items()})


def _check_step(rep, s, ag, spec, ref, before, alpha, log_alpha, lr_steps, grad_gate=True):
    """ref: dict with the reference/oracle values of this step (any subset of the keys below).
    grad_gate=False: gradients are not compared here (the caller's sizes are covered element by element by
    test_gradient_parity_three_way_fp64, which separates kink flips from rounding with an fp64 evaluation)."""
    from oracle import update as oup
    T, B = spec.T, spec.B
    shapes = {"state": (T, B, spec.latent), "next_action": (T - 1, B, spec.act), "next_log_pi": (T - 1, B, 1),
              "next_z": (T - 1, B, spec.Nq), "q_pred": (T - 1, B, spec.Nq), "pi": (T - 1, B, spec.act),
              "log_pi": (T - 1, B, 1), "q_frozen": (T - 1, B, spec.Nq), "q_loss": (T - 1, B, 1),
              "pi_loss": (T - 1, B, 1), "alpha_loss": (T - 1, B, 1), "is_contiguous": (T - 1, B, 1)}
    cn = logp_cond(ref["next_action"]) if "next_action" in ref and not spec.discrete else 0.0
    cc = logp_cond(ref["pi"]) if "pi" in ref and not spec.discrete else 0.0
    ent = 1.0 if spec.max_entropy else 0.0
    extra = {"log_pi": cc, "next_log_pi": cn, "q_loss": ent * alpha * spec.gamma * cn, "pi_loss": alpha * cc,
             "alpha_loss": abs(log_alpha) * cc}
    for name, shp in shapes.items():
        if name in ref and ref[name] is not None:
            rep.check(f"s{s}.{name}", ag.debug(name, shp), np.asarray(ref[name]).reshape(shp), TOL, extra.get(name))
    if "loss" in ref:
        row = extra["q_loss"] + extra["pi_loss"] + extra["alpha_loss"]
        slack = 2.0 * float(np.mean(row)) if np.ndim(row) else 0.0
        got, want = ag.scalars()["loss"], float(ref["loss"])
        rep.check(f"s{s}.loss", np.asarray([got]), np.asarray([want]), 2 * TOL * max(1.0, 1.0 / max(abs(want), 1e-30)),
                  np.asarray([slack]))
    if "dq_pred" in ref and grad_gate:
        rep.check_frac(f"s{s}.dz", ag.debug("dz", (T - 1, B, spec.Nq)), ref["dq_pred"], GTOL, 0.999, 2e-3)
    if "grad" in ref and grad_gate:
        for n, gr in ref["grad"].items():
            rep.check_frac(f"s{s}.grad.{n}", ag.grad_views[n], gr, GTOL, 0.98, 2e-3)
    # optimiser arithmetic, exactly, from the GPU's own gradient (torch.optim.Adam restated in oracle.update)
    p0, m0, v0 = before
    step = int(ag.scalars()["step"])
    for n in ag.trainable:
        g = ag.grad_views[n].cpu()
        pn, mn, vn = oup.adam_update(p0[n].cpu(), g, m0[n].cpu(), v0[n].cpu(), step, spec.lr)
        rep.check(f"s{s}.adam_p.{n}", ag.tensors[n], pn, 1e-6)
        rep.check(f"s{s}.adam_m.{n}", ag.m_views[n], mn, 1e-6)
        rep.check(f"s{s}.adam_v.{n}", ag.v_views[n], vn, 1e-6)
    for n in ag.tensors:
        if "_target." in n:
            src = n.replace("_target.", ".")
            want = ag.tensors[src].cpu() if spec.hard_updates else p0[n].cpu() * (1.0 - spec.tau) + ag.tensors[src].cpu() * spec.tau
            rep.check(f"s{s}.polyak.{n}", ag.tensors[n], want, 1e-6)
        if "_frozen." in n:   # critic_frozen <- critic before the optimiser step (soft_actor_critic.py:142)
            rep.check(f"s{s}.frozen.{n}", ag.tensors[n], p0[n.replace("_frozen.", ".")], 0.0)
    if "after" in ref:
        for n, v in ref["after"].items():
            got = ag.tensors[n].cpu().double().numpy()
            want = np.asarray(v, np.float64)
            d = np.abs(got - want)
            base = TOL * (np.abs(want).max() + 1e-30)
            assert d.max() <= base + 2.2 * spec.lr * lr_steps, (s, "after", n, d.max())
            frac = float(np.mean(d > base + 0.05 * spec.lr * lr_steps))
            rep.rows.append((f"s{s}.after.{n} frac>0.05lr", frac, frac, 0.02))
            if frac > 0.02:
                rep.bad.append((f"s{s}.after.{n}", frac, frac, 0.02))


CONT_CASES = ["tqc_small", "tqc_c5q2", "tqc_goal", "sac_min", "tqc_nolb", "tqc_discrete", "sac_boot", "tqc_burn", "gru_zero", "gru_learned", "gru_store"]


@pytest.mark.parametrize("case", CONT_CASES)
def test_update_matches_reference_golden(dev, case):
    g = load("update_" + case)
    spec = spec_from_case(g["case"])
    ag = _agent_for(spec, dev)
    ag.load_tensors({k: torch.tensor(v) for k, v in g["init"].items()})
    rep = Report("golden:" + case)
    n_steps = len([k for k in g if k.startswith("step")])
    for s in range(n_steps):
        rec = g[f"step{s}"]
        xp = {k: torch.tensor(v).to(dev) for k, v in rec["batch"].items()}
        before = _snapshot(ag)
        log_alpha = float(ag.tensors["actor_critic.log_alpha"])
        ag.update(xp, torch.tensor(rec["noise_target"]).to(dev), torch.tensor(rec["noise_actor"]).to(dev))
        sc = ag.scalars()
        assert abs(sc["alpha"] - float(rec["alpha_in"])) <= 1e-6 * abs(sc["alpha"])
        _check_step(rep, s, ag, spec, rec, before, sc["alpha"], log_alpha, s + 1)
        if "adam_m" in rec:
            for n, v in rec["adam_m"].items():
                rep.check(f"s{s}.ref_m.{n}", ag.m_views[n], v, GTOL)
            for n, v in rec["adam_v"].items():
                rep.check(f"s{s}.ref_v.{n}", ag.v_views[n], v, GTOL)
    rep.finish()


@pytest.mark.parametrize("T,B,hid", [(4, 64, 256), (50, 256, 256)])
def test_update_matches_oracle_config2(dev, T, B, hid):
    """BASELINE config 2 dims (obs 17, act 6, 5x2-head TQC, MLP 256) against the CPU oracle."""
    from oracle import update as oup
    torch.manual_seed(0)
    spec = oup.Spec(obs=17, act=6, C=5, Q=2, latent=hid, enc_features=hid, enc_hidden=(hid,), joint_hidden=(hid,),
                    pi_hidden=(hid,), critic_hidden=(hid, hid), T=T, B=B)
    params = oup.init_params(spec, seed=3)
    st = oup.new_state(spec, params)
    ag = _agent_for(spec, dev)
    ag.load_tensors(params)
    rep = Report(f"oracle:config2 T={T} B={B}")
    g = torch.Generator().manual_seed(1)
    for step in range(2 if T * B < 10000 else 1):   # (full size: one step - the second costs 8 s of CPU oracle and exercises the same launches)
        xp = {"obs_1d": torch.randn(T, B, 17, generator=g), "action": torch.rand(T, B, 6, generator=g) * 2 - 1,
              "reward": torch.randn(T, B, 1, generator=g), "mc_return": torch.randn(T, B, 1, generator=g) * 2,
              "task_done": (torch.rand(T, B, 1, generator=g) < 0.05).float(),
              "episode_step": (torch.arange(T).view(T, 1, 1) + torch.randint(0, 900, (1, B, 1), generator=g)).float()}
        xp["episode_step"][T // 2:, ::7] = torch.arange(T - T // 2).view(-1, 1, 1).float()   # episode boundaries
        nt, na = torch.randn(T - 1, B, 6, generator=g), torch.randn(T - 1, B, 6, generator=g)
        alpha, log_alpha = st.alpha, float(st.params["actor_critic.log_alpha"])
        # every step starts from the oracle's exact state: after an Adam step the two trajectories
        # legitimately differ by up to 2*lr on near-zero-gradient weights
        ag.load_tensors({k: v for k, v in st.params.items() if "_frozen." not in k})
        ag.load_opt_state(st.adam_m, st.adam_v, st.step, st.alpha)
        loss, aux = oup.train_step(st, spec, xp, nt, na)
        before = _snapshot(ag)
        ag.update({k: v.to(dev) for k, v in xp.items()}, nt.to(dev), na.to(dev))
        ref = {k: (v.detach().numpy() if isinstance(v, torch.Tensor) else v) for k, v in aux.items() if k != "grad"}
        ref["loss"] = float(loss)
        ref["grad"] = aux["grad"]
        ref["after"] = {n: st.params[n] for n in ag.tensors if "_frozen." not in n}
        # at the full size a handful of the 3.2 M units per layer sit on their LeakyReLU kink and pick their branch by
        # rounding, differently in any two fp32 evaluations; the fraction-based gradient gate cannot tell that from an
        # error, the fp64 three-way test below can - it owns the gradient comparison at (T, B) = (50, 256)
        _check_step(rep, step, ag, spec, ref, before, alpha, log_alpha, step + 1, grad_gate=(T * B < 10000))
    rep.finish()


@pytest.mark.parametrize("name,kw", [
    ("config4 dims (Humanoid obs 376, act 17, 5x25 quantiles)", dict(obs=376, act=17, C=5, Q=25, T=3, B=96)),
    ("config3 dims (obs 28 + goals 10, HER-style batch)", dict(obs=28, goal=10, act=6, C=5, Q=2, T=4, B=64)),
    ("config1 dims (Pendulum SAC-min, 2 critics)", dict(obs=3, act=1, C=2, Q=1, T=5, B=128, distributional=False)),
    ("SAC-min + window-long bootstrap lower bound (use_bootstrap_minibatch_nstep), T=50",
     dict(obs=3, act=1, C=2, Q=1, T=50, B=64, distributional=False, bootstrap=True)),
    ("GRU joiner, learned start state, T=50 scan (encoder.py:40-42, 78-94)",
     dict(obs=9, act=3, C=2, Q=3, T=50, B=48, gru="learned", latent=64, enc_features=48, enc_hidden=(64,), joint_hidden=(64,),
          pi_hidden=(64,), critic_hidden=(64, 64))),
    ("GRU joiner, stored start state + burn-in rows", dict(obs=9, act=3, C=2, Q=3, T=12, B=40, gru="store", burn_in=2,
                                                          latent=64, enc_features=48, enc_hidden=(64,), joint_hidden=(64,),
                                                          pi_hidden=(64,), critic_hidden=(64, 64))),
    ("GRU joiner, persistent scans at the default width (latent 256: four waves per workgroup), 30 windows (a partial 4-row tile), "
     "zero start state", dict(obs=9, act=3, C=2, Q=3, T=9, B=30, gru="zero", latent=256, enc_features=64, enc_hidden=(64,),
                              joint_hidden=(64,), pi_hidden=(64,), critic_hidden=(64, 64))),
    ("GRU joiner, persistent scans at latent 128 (two waves), learned start state, T=50",
     dict(obs=9, act=3, C=2, Q=3, T=50, B=20, gru="learned", latent=128, enc_features=48, enc_hidden=(64,), joint_hidden=(64,),
          pi_hidden=(64,), critic_hidden=(64, 64))),
    ("GRU joiner on the step-by-step launches (FDQL_GRU_SCAN=0: T x (recurrent GEMM + gate kernel))",
     dict(obs=9, act=3, C=2, Q=3, T=12, B=40, gru="learned", latent=64, enc_features=48, enc_hidden=(64,), joint_hidden=(64,),
          pi_hidden=(64,), critic_hidden=(64, 64), env={"FDQL_GRU_SCAN": "0"})),
    ("pixel encoder, 2 conv layers on 2x12x12 frames + obs_1d, discrete head (no reference: vs torch conv2d)",
     dict(obs=4, act=5, discrete=True, C=2, Q=3, T=3, B=20, img=(2, 12, 12), conv=((8, 4, 2), (16, 3, 1)), latent=32,
          enc_features=32, enc_hidden=(48,), joint_hidden=(32,), pi_hidden=(32,), critic_hidden=(32, 32))),
    ("pixel encoder alone (obs_dim 0), Atari stack on 4x84x84, config-5 shapes at a small batch",
     dict(obs=0, act=6, discrete=True, C=2, Q=5, T=2, B=6, img=(4, 84, 84), conv=((32, 8, 4), (64, 4, 2), (64, 3, 1)))),
    ("pixel encoder alone, Atari stack, the frames as a uint8 batch (agent created with obs_2d_u8): every conv layer on the "
     "implicit-GEMM kernels (csrc/conv.hip) - forward from the bytes, gather-form data gradients, output-stationary weight gradients",
     dict(obs=0, act=6, discrete=True, C=2, Q=5, T=2, B=6, img=(4, 84, 84), conv=((32, 8, 4), (64, 4, 2), (64, 3, 1)), frames="u8")),
    ("the same with the first layer reading the ring's uint8 block in place through one slot index per row (batch.obs_2d_slots), "
     "obs_1d beside the frames, T=3, B=7 (21 frame stacks: odd image groups in every conv launch)",
     dict(obs=5, act=6, discrete=True, C=2, Q=5, T=3, B=7, img=(4, 84, 84), conv=((32, 8, 4), (64, 4, 2), (64, 3, 1)), frames="ring")),
    ("Atari stack as float32 frames (first layer on im2col + GEMM, layers 1-2 on the implicit-GEMM kernels) with the implicit path "
     "switched off altogether (FDQL_NO_IMPLICIT_CONV: the im2col / col2im path, the checker of the new one)",
     dict(obs=0, act=6, discrete=True, C=2, Q=5, T=2, B=6, img=(4, 84, 84), conv=((32, 8, 4), (64, 4, 2), (64, 3, 1)),
          env={"FDQL_NO_IMPLICIT_CONV": "1"})),
    ("config5 head (discrete SAC, 6 actions, Gumbel-softmax)", dict(obs=64, act=6, discrete=True, C=2, Q=5, T=4, B=128)),
    ("ragged sizes (B=7, odd widths 18/33/21: unaligned rows, partial tiles, M < one tile)",
     dict(obs=3, act=2, C=2, Q=3, T=3, B=7, critic_hidden=(33, 18), pi_hidden=(21,), enc_hidden=(18,), joint_hidden=(33,),
          latent=21, enc_features=18)),
    ("hard target updates, no entropy bonus, no lower bound", dict(obs=6, act=2, C=2, Q=5, T=4, B=32, hard_updates=True,
                                                                     max_entropy=False, lowerbound=False, latent=32,
                                                                     enc_features=32, enc_hidden=(32,), joint_hidden=(32,),
                                                                     pi_hidden=(32,), critic_hidden=(32, 32))),
    ("head-only MLPs (no hidden layers in actor / encoder)", dict(obs=6, act=2, C=2, Q=5, T=4, B=32, latent=32, enc_features=32,
                                                                    enc_hidden=(), joint_hidden=(), pi_hidden=(),
                                                                    critic_hidden=(32,))),
    ("deep nets (3-layer critic, 2-layer actor/encoder)", dict(obs=9, act=4, C=3, Q=5, T=4, B=40, critic_hidden=(64, 96, 64),
                                                               pi_hidden=(64, 48), enc_hidden=(80, 64), joint_hidden=(64, 64),
                                                               latent=64, enc_features=48)),
    ("fused skip-head partials, 4 quantiles, ragged critic widths 96/40 (partial column tiles, 3-layer head sum)",
     dict(obs=7, act=3, C=3, Q=4, T=5, B=52, critic_hidden=(96, 40, 70), pi_hidden=(64,), enc_hidden=(64,), joint_hidden=(64,),
          latent=64, enc_features=48)),
    ("fused skip-head partials, 8 quantiles (one row per butterfly group)",
     dict(obs=7, act=3, C=2, Q=8, T=4, B=33, critic_hidden=(128, 64), pi_hidden=(64,), enc_hidden=(64,), joint_hidden=(64,),
          latent=64, enc_features=48)),
    ("config 2 dims with the head fusion switched off (FDQL_NO_HEAD_FUSE: the head streams every activation)",
     dict(obs=17, act=6, C=5, Q=2, T=4, B=64, env={"FDQL_NO_HEAD_FUSE": "1"})),
    ("config 2 dims without the dual-output first layer (FDQL_NO_DUAL)",
     dict(obs=17, act=6, C=5, Q=2, T=4, B=64, env={"FDQL_NO_DUAL": "1"})),
    ("config 2 dims, critic layers forced onto the weight-stationary row-block kernel at a small batch (FDQL_ROWGEMM=all: "
     "8 tiles per instance, prologue + flush paths, per-workgroup column sums)",
     dict(obs=17, act=6, C=5, Q=2, T=5, B=64, env={"FDQL_ROWGEMM": "all"})),
    ("3-layer 256-wide critics on the weight-stationary kernel (non-fused dgrad form on the middle layer, fused on the last); "
     "gradients by the fp64 three-way test: with this case's data ONE unit of critic 2's middle layer sits 1.0e-7 of the layer's "
     "scale from its LeakyReLU kink and the forward chain kernels round it to different sides (tools/diag in DESIGN section 2)",
     dict(obs=17, act=6, C=3, Q=2, T=5, B=64, critic_hidden=(256, 256, 256), env={"FDQL_ROWGEMM": "all"}, grad_gate=False)),
    ("config 2 dims, every MLP forward through the row-block chain kernel (FDQL_CHAIN=all: encoder/joiner/actors in one "
     "program, each critic instance in one)", dict(obs=17, act=6, C=5, Q=2, T=4, B=64, env={"FDQL_CHAIN": "all"})),
    ("config 2 dims on per-layer launches only (FDQL_CHAIN=0)", dict(obs=17, act=6, C=5, Q=2, T=6, B=192, env={"FDQL_CHAIN": "0"})),
    ("chain kernel, goal-conditioned input (three K-segments in the load), 25-quantile heads (two 16-column head tiles)",
     dict(obs=28, goal=10, act=6, C=3, Q=25, T=4, B=70, env={"FDQL_CHAIN": "all"})),
    ("chain kernel, ragged sizes (B=7, odd widths 18/33/21, partial row block)",
     dict(obs=3, act=2, C=2, Q=3, T=3, B=7, critic_hidden=(33, 18), pi_hidden=(21,), enc_hidden=(18,), joint_hidden=(33,),
          latent=21, enc_features=18, env={"FDQL_CHAIN": "all"})),
    ("chain kernel, deep nets (3-layer critic, 2-layer actor / encoder: wide heads over three feature blocks)",
     dict(obs=9, act=4, C=3, Q=5, T=4, B=40, critic_hidden=(64, 96, 64), pi_hidden=(64, 48), enc_hidden=(80, 64),
          joint_hidden=(64, 64), latent=64, enc_features=48, env={"FDQL_CHAIN": "all"})),
    ("chain kernel, discrete SAC head (Gumbel-softmax, one-hot critic input)",
     dict(obs=64, act=6, discrete=True, C=2, Q=5, T=4, B=128, env={"FDQL_CHAIN": "all"})),
    ("config 4 dims (376 observation columns: only 32-row blocks fit the chain kernel's LDS), encoder -> joiner -> actors chained "
     "(FDQL_CHAIN=enc), 5x25 quantiles", dict(obs=376, act=17, C=5, Q=25, T=3, B=80, env={"FDQL_CHAIN": "enc"})),
    ("config 2 dims with an odd row count (T=4, B=33: 99 gradient rows, one K-split slab), every narrow weight gradient on the "
     "streaming kernel (FDQL_STREAM_WGRAD=2): the last row pair is half empty",
     dict(obs=17, act=6, C=5, Q=2, T=4, B=33, env={"FDQL_STREAM_WGRAD": "2"})),
    ("config 2 dims, T=9, B=100 (800 gradient rows, 2 slabs of 400): riders on the output-stationary launch forced at a small "
     "size (FDQL_ROWGEMM=all FDQL_WGRAD_STAT_FACTOR=1), streaming kernel for the rest",
     dict(obs=17, act=6, C=5, Q=2, T=9, B=100, env={"FDQL_ROWGEMM": "all", "FDQL_STREAM_WGRAD": "2"})),
    ("config 2 dims, T=5, B=64: the row-block dgrad kernel forced at 4 blocks (FDQL_ROWDGRAD_MIN_BLOCKS=1: gated one-segment "
     "form, plain two-segment form, column sums)", dict(obs=17, act=6, C=5, Q=2, T=5, B=64, env={"FDQL_ROWGEMM": "all"})),
    ("TQC loss, one wave per row with ONE atom per lane (5 x 8 = 40 pooled atoms, 32 kept: k_loss_wave<1>)",
     dict(obs=9, act=3, C=5, Q=8, T=4, B=50)),
    ("TQC loss, one wave per row, 3 x 25 = 75 atoms over two slots per lane with a ragged second slot (k_loss_wave<2>), odd row count",
     dict(obs=9, act=3, C=3, Q=25, T=3, B=33)),
    ("config4 dims with the thread-per-atom loss kernel (FDQL_LOSS_WAVE=0: 125 atoms on 128-thread groups)",
     dict(obs=376, act=17, C=5, Q=25, T=3, B=96, env={"FDQL_LOSS_WAVE": "0"})),
    ("temporal_len 2 at config 2 dims, B=256 (512 rows): every GEMM stage that is not a row-block launch on the small-batch "
     "kernel (smallgemm.hip: K split over the waves of a workgroup; forward with K-segments, gated dgrads + column sums, "
     "weight gradients, d state shares)", dict(obs=17, act=6, C=5, Q=2, T=2, B=256)),
    ("temporal_len 2, B=7, goal-conditioned rows, 25-quantile heads (partial 64-row tiles, 17-wide and 10-wide K-segments, "
     "unaligned head rows on the small-batch kernel)", dict(obs=28, goal=10, act=17, C=3, Q=25, T=2, B=7)),
    ("config 2 dims at T=6, B=64 with the small-batch kernel forced on every problem that has its form (dense shape 7: "
     "320-row problems, the dual / head-fusion problems stay on their tile shapes)", dict(obs=17, act=6, C=5, Q=2, T=6, B=64, dense_shape=7)),
    ("temporal_len 2 at config 2 dims on the tile kernels (FDQL_SMALL_GEMM=0: the path the small-batch kernel replaces)",
     dict(obs=17, act=6, C=5, Q=2, T=2, B=256, env={"FDQL_SMALL_GEMM": "0"})),
    ("ragged sizes on the tile kernels (FDQL_SMALL_GEMM=0; B=7, odd widths 18/33/21: unaligned rows, partial tiles, M < one tile)",
     dict(obs=3, act=2, C=2, Q=4, T=3, B=7, critic_hidden=(33, 18), pi_hidden=(21,), enc_hidden=(18,), joint_hidden=(33,),
          latent=21, enc_features=18, env={"FDQL_SMALL_GEMM": "0"})),
    ("policy backward and the actor's last-layer gradient in one launch (k_policy_bwd_dpre): 8 actions (2A = 16, the widest head it "
     "takes), 99 gradient rows (partial 64-row block)", dict(obs=17, act=8, C=3, Q=4, T=4, B=33)),
    ("config 2 dims with the actor's last-layer gradient as a GEMM stage of its own (FDQL_NO_POLICY_DPRE_FUSE: the path the fused "
     "launch replaces)", dict(obs=17, act=6, C=5, Q=2, T=4, B=64, env={"FDQL_NO_POLICY_DPRE_FUSE": "1"})),
    ("config 2 dims with the bias gradients' column sums in a launch of their own (FDQL_NO_COLSUM_STREAM: k_skinny_wgrad, the path "
     "the streaming launch's column-sum forms replace)", dict(obs=17, act=6, C=5, Q=2, T=6, B=192, env={"FDQL_NO_COLSUM_STREAM": "1"})),
    ("12 action columns, 8 quantiles per critic (column sums of d logits over 24 columns on the streaming launch's 40-column "
     "lanes-over-rows form, d z's 8 columns on the 8-column one, 256-wide partial rows in row groups), 99 rows",
     dict(obs=11, act=12, C=3, Q=8, T=4, B=33)),
    ("config 4 dims (5x25 quantiles) with the weight-stationary launches forced at T=3, B=96 (FDQL_ROWGEMM=all): the last hidden "
     "layer's rank-25 gradient gated by the forward launches' masks (k_head_dgrad_masked), the layer-0 dgrad on the masked "
     "weight-stationary form with 4 narrow steps", dict(obs=376, act=17, C=5, Q=25, T=3, B=96, env={"FDQL_ROWGEMM": "all"})),
    ("the same with the rank-25 gradient as a GEMM stage (FDQL_NO_HEAD_DGRAD_MASKED)",
     dict(obs=376, act=17, C=5, Q=25, T=3, B=96, env={"FDQL_ROWGEMM": "all", "FDQL_NO_HEAD_DGRAD_MASKED": "1"})),
    ("config 2 dims at T=5, B=64 with the weight-stationary and the row-block dgrad kernels forced (FDQL_ROWGEMM=all, "
     "FDQL_ROWDGRAD_MIN_BLOCKS=1): the sum of the d state shares inside the joiner's dgrad launch (RowDgradArgs::sum_*), column sums "
     "per 64 rows", dict(obs=17, act=6, C=5, Q=2, T=5, B=64, env={"FDQL_ROWGEMM": "all"})),
    ("the same with the three dgrad launches behind d state kept apart (FDQL_NO_ROWDGRAD_CHAIN: the sum folded into the first one "
     "only; the default runs them as one k_rowdgrad_chain launch)",
     dict(obs=17, act=6, C=5, Q=2, T=5, B=64, env={"FDQL_ROWGEMM": "all", "FDQL_NO_ROWDGRAD_CHAIN": "1"})),
    ("config 2 dims, weight-stationary launches forced, gates from the activations themselves (FDQL_NO_GATE_MASKS: the path the "
     "forward launches' sign masks replace)", dict(obs=17, act=6, C=5, Q=2, T=5, B=64, env={"FDQL_ROWGEMM": "all", "FDQL_NO_GATE_MASKS": "1"})),
    ("config 2 dims on the weight-stationary kernel writing every head plane (FDQL_NO_HEAD_PRESUM: plane-sum launch + finish, the "
     "path the in-kernel plane sum replaces)", dict(obs=17, act=6, C=5, Q=2, T=5, B=64, env={"FDQL_ROWGEMM": "all", "FDQL_NO_HEAD_PRESUM": "1"})),
    ("config 2 dims at T=50, B=32: one rank's share of the 256-window global batch at N = 8 (1 568 gradient rows: d state as per-network "
     "problems + a summing pass, the tile / small-batch kernels for the single networks, weight-stationary critics layer 1)",
     dict(obs=17, act=6, C=5, Q=2, T=50, B=32)),
    ("config 2 dims at T=50, B=128: one rank's share at N = 2 (6 272 gradient rows: encoder -> joiner -> actors as 200 chain blocks of "
     "32 rows, k_chain<1>, the output-stationary weight-gradient launch at 11 tiles per workgroup)", dict(obs=17, act=6, C=5, Q=2, T=50, B=128)),
    ("config 2 dims at T=50, B=64: one rank's share at N = 4 (3 136 gradient rows: 100 chain blocks of 32 rows, the dense weight "
     "gradients as K-split problems riding in the dgrad launches)", dict(obs=17, act=6, C=5, Q=2, T=50, B=64)),
    ("config 2 dims at T=50, B=384: 18 816 gradient rows = 588 blocks of 32 - k_rowdgrad_chain<2> (d state sum + joiner / d enc / "
     "encoder dgrads in one launch) with MORE blocks than one round of workgroups; 600 forward blocks of 32: past k_fwd3's one round, "
     "k_chain<2>", dict(obs=17, act=6, C=5, Q=2, T=50, B=384)),
    ("config 2 dims at T=50, B=32 on the launches the small-block kernels replace (FDQL_NO_ROWDGRAD_CHAIN: summing launch + three "
     "dgrad launches; FDQL_CHAIN=0: six per-layer forward launches)", dict(obs=17, act=6, C=5, Q=2, T=50, B=32,
                                                                           env={"FDQL_NO_ROWDGRAD_CHAIN": "1", "FDQL_CHAIN": "0"})),
    ("config 2 dims at T=50, B=64 with encoder -> joiner -> actors on k_chain<1> (FDQL_CHAIN=enc: 100 blocks of 32 rows, the launch "
     "k_fwd3<1> replaces at this size)", dict(obs=17, act=6, C=5, Q=2, T=50, B=64, env={"FDQL_CHAIN": "enc"})),
    ("discrete head (6 logits) and 40 windows (B % 16 != 0: no small-block forward kernel; 1 960 gradient rows, not a multiple of 16: "
     "no small-block dgrad chain)", dict(obs=17, act=6, discrete=True, C=3, Q=4, T=50, B=40)),
])
def test_update_matches_oracle_other_configs(dev, name, kw, monkeypatch):
    """The remaining BASELINE configs' shapes (and non-default depths) against the CPU oracle, one step."""
    from oracle import update as oup
    kw = dict(kw)
    for k, v in kw.pop("env", {}).items():      # plan switches are read when the agent is created
        monkeypatch.setenv(k, v)
    dense_shape = kw.pop("dense_shape", None)   # tile shape of the dense GEMMs (process-wide setting: restored below)
    if dense_shape is not None:
        from fastdeepqlearning_amd import _native as nat
        nat.check(nat.load().fdql_debug_set_gemm_dense_shape(dense_shape))
    try:
        _run_other_config(dev, name, kw)
    finally:
        if dense_shape is not None:
            nat.check(nat.load().fdql_debug_set_gemm_dense_shape(5))


def _run_other_config(dev, name, kw):
    from oracle import update as oup
    T, B = kw.pop("T"), kw.pop("B")
    grad_gate = kw.pop("grad_gate", True)   # False: the case's gradients are compared by test_gradient_parity_three_way_fp64 (a unit on its kink)
    frames = kw.pop("frames", None)   # "u8": uint8 frame batch; "ring": the ring's uint8 block read in place (obs_2d_slots)
    base = dict(latent=256, enc_features=256, enc_hidden=(256,), joint_hidden=(256,), pi_hidden=(256,), critic_hidden=(256, 256))
    base.update(kw)
    spec = oup.Spec(T=T, B=B, **base)
    params = oup.init_params(spec, seed=5)
    st = oup.new_state(spec, params)
    ag = _agent_for(spec, dev, obs_2d_u8=frames is not None)
    if frames is not None:
        assert ag.conv_reads_ring()
    ag.load_tensors(params)
    g = torch.Generator().manual_seed(9)
    A = spec.act
    xp = {"obs_1d": torch.randn(T, B, max(spec.obs, 1), generator=g), "action": torch.rand(T, B, A, generator=g) * 2 - 1,
          "reward": torch.randn(T, B, 1, generator=g), "mc_return": torch.randn(T, B, 1, generator=g) * 2,
          "task_done": (torch.rand(T, B, 1, generator=g) < (0.002 if spec.bootstrap else 0.1)).float(),
          "episode_step": (torch.arange(T).view(T, 1, 1) + torch.randint(0, 50, (1, B, 1), generator=g)).float()}
    if spec.discrete:
        xp["action"] = torch.randint(0, A, (T, B, 1), generator=g).float()
    if spec.goal:
        xp["achieved_goal"] = torch.randn(T, B, spec.goal, generator=g)
        xp["desired_goal"] = torch.randn(T, B, spec.goal, generator=g)
    xp["episode_step"][T // 2:, ::5] = 0.0
    if spec.gru == "store":
        xp["agent_state"] = torch.rand(T, B, spec.latent, generator=g)
    if spec.img:
        xp["obs_2d"] = torch.randint(0, 256, (T, B) + tuple(spec.img), generator=g).float()
    if not spec.obs:
        del xp["obs_1d"]
    nt, na = torch.randn(T - 1, B, A, generator=g), torch.randn(T - 1, B, A, generator=g)
    if spec.discrete:
        nt, na = torch.rand(T - 1, B, A, generator=g), torch.rand(T - 1, B, A, generator=g)
    alpha, log_alpha = st.alpha, float(st.params["actor_critic.log_alpha"])
    loss, aux = oup.train_step(st, spec, xp, nt, na)
    if spec.bootstrap:   # the window-long bound must actually bind somewhere, or the case tests nothing
        assert int(((aux["bootstrap_lowerbound"] * aux["is_contiguous"].prod(0)) > 0).sum()) >= 8
    before = _snapshot(ag)
    xd = {k: v.to(dev) for k, v in xp.items()}
    if frames == "u8":
        xd["obs_2d"] = xp["obs_2d"].to(torch.uint8).to(dev)
    elif frames == "ring":   # a ring block of T B + 9 slots holding the batch's frames at scattered slots
        nslot = T * B + 9
        perm = torch.randperm(nslot, generator=g)[:T * B]
        block = torch.randint(0, 256, (nslot,) + tuple(spec.img), generator=g, dtype=torch.uint8)
        block[perm] = xp["obs_2d"].to(torch.uint8).reshape((T * B,) + tuple(spec.img))
        xd["obs_2d"] = block.to(dev)
        xd["obs_2d_slots"] = perm.to(torch.int32).view(T, B).to(dev)
    ag.update(xd, nt.to(dev), na.to(dev))
    rep = Report("oracle:" + name)
    ref = {k: (v.detach().numpy() if isinstance(v, torch.Tensor) else v) for k, v in aux.items() if k != "grad"}
    ref["loss"], ref["grad"] = float(loss), aux["grad"]
    ref["after"] = {n: st.params[n] for n in ag.tensors if "_frozen." not in n}
    _check_step(rep, 0, ag, spec, ref, before, alpha, log_alpha, 1, grad_gate=grad_gate)
    rep.finish()


def test_ring_uint8_keys(dev):
    """Pixel keys stored as one byte per element (replay_memory.py:26-35 keeps uint8; torch_dataloader.py:36 casts at
    read time): windows, single rows, sub-row selection and checkpoint round trip are bit-exact against the numpy
    ring; the HBM block really is maxlen * dim bytes."""
    from fastdeepqlearning_amd.core import NativeRing
    from oracle.replay import RingOracle
    maxlen, T, B = 300, 5, 12
    dims, dtypes = [4 * 6 * 6, 3, 1, 5], ["u8", "f32", "f32", "u8"]
    keys = ["frames", "action", "episode_step", "odd_u8"]
    rng = np.random.RandomState(0)
    n = 470   # wraps
    cols = [rng.randint(0, 256, (n, dims[0])).astype(np.float32), rng.standard_normal((n, 3)).astype(np.float32),
            (np.arange(n) % 40).astype(np.float32).reshape(n, 1), rng.randint(0, 256, (n, 5)).astype(np.float32)]
    rows = np.concatenate(cols, 1)
    free0 = torch.cuda.mem_get_info(dev)[0]
    big = NativeRing(4_000_000, [dims[0]], dev, dtypes=["u8"])
    used = free0 - torch.cuda.mem_get_info(dev)[0]
    assert used < 1.25 * 4_000_000 * dims[0] + (64 << 20), used     # ~0.58 GB as bytes; 2.3 GB if it were float32
    del big
    ring = NativeRing(maxlen, dims, dev, dtypes=dtypes)
    ring.add_rows(rows[:200])
    ring.add_rows(torch.tensor(rows[200:]).to(dev))       # device append path
    orc = RingOracle(maxlen, B, T)
    for i in range(n):
        orc.add({k: c[i] for k, c in zip(keys, cols)})
    assert len(ring) == len(orc)
    outs, starts = ring.sample_windows(T, B, seed=3, counter=1, return_starts=True)
    want = orc.temporal_sample(starts=starts.cpu().numpy())
    for k, o in zip(keys, outs):
        assert np.array_equal(o.cpu().numpy(), want[k].astype(np.float32)), k
    idx = torch.tensor(rng.randint(0, len(ring), 17))
    got = ring.sample_rows(17, idx=idx)
    for k, o in zip(keys, got):
        assert np.array_equal(o.cpu().numpy(), orc.gather(idx.numpy())[k].astype(np.float32)), k
    sel = ring.sample_windows(T, B, starts=starts, select={0: (36, 72), 3: (1, 3), 1: None})
    assert sel[1] is None
    assert np.array_equal(sel[0].cpu().numpy(), want["frames"][..., 36:108].astype(np.float32))
    assert np.array_equal(sel[3].cpu().numpy(), want["odd_u8"][..., 1:4].astype(np.float32))
    snap, top, ln = ring.snapshot()
    ring2 = NativeRing(maxlen, dims, dev, dtypes=dtypes)
    ring2.restore(snap, top, ln)
    outs2 = ring2.sample_windows(T, B, starts=starts)
    for a, b in zip(outs, outs2):
        assert torch.equal(a, b)


def test_ring_full_size_properties(dev):
    """BASELINE size (1M-slot ring, T=50, B=256): size-independent properties instead of an oracle copy.
    Slot i stores its own index, so every gathered element is checkable in closed form:
    out[t, b] == (start[b] + t) % len, windows are consecutive, sample() rows equal their index,
    re-sampling with the same (seed, counter) is idempotent, and the wrap leaves len = maxlen - 1."""
    from fastdeepqlearning_amd.core import NativeRing
    maxlen, T, B = 1_000_000, 50, 256
    dims = [17, 6, 1, 1]
    ring = NativeRing(maxlen, dims, dev)
    n = maxlen + 12_345                      # wraps once
    for c0 in range(0, n, 250_000):
        m = min(250_000, n - c0)
        idx = torch.arange(c0, c0 + m, device=dev, dtype=torch.float32)
        rows = torch.cat([(idx % 4099).view(-1, 1).expand(m, 17), idx.view(-1, 1).expand(m, 6) % 7,
                          (idx % maxlen).view(-1, 1), torch.ones(m, 1, device=dev)], dim=1).contiguous()
        ring.add_rows(rows)
    assert len(ring) == maxlen - 1 and ring.top == n % maxlen
    outs, starts = ring.sample_windows(T, B, seed=3, counter=7, return_starts=True)
    slot = outs[2][..., 0]                   # the stored slot index
    want = (starts.view(1, B) + torch.arange(T, device=dev).view(T, 1)) % len(ring)
    assert torch.equal(slot.long(), want)
    step = (slot[1:] - slot[:-1])
    assert bool(((step == 1) | (step == 1 - len(ring))).all())
    # rows written after the wrap hold index (slot + maxlen) in the other keys
    logical = torch.where(slot < ring.top, slot + maxlen, slot)
    assert torch.equal(outs[0][..., 5], logical % 4099) and torch.equal(outs[1][..., 2], logical % 7)
    assert float(outs[3].sum()) == T * B
    outs2, starts2 = ring.sample_windows(T, B, seed=3, counter=7, return_starts=True)
    assert torch.equal(starts, starts2) and all(torch.equal(a, b) for a, b in zip(outs, outs2))
    flat = ring.sample_rows(4096, seed=5, counter=1)
    assert bool((flat[2][:, 0] < len(ring)).all()) and bool((flat[2][:, 0] >= 0).all())
    hist = torch.histc(starts.float(), bins=8, min=0, max=len(ring))
    assert float(hist.min()) > 0             # starts spread over the whole ring


# ---------------------------------------------------------------------------------------
# act(): SURVEY 8f rank 1 - device-resident inference on the trainer's weight arena
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", ACT_CASES)
def test_act_matches_reference_golden(dev, case):
    """fdql_agent_act == franQ DeepQLearning.act (deepQlearning.py:155-187) on the reference's own
    noise draw: integer actions exact, float outputs within TOL (log_prob beyond its conditioning)."""
    g = load("act_" + case)
    spec = spec_from_case(g["case"])
    ag = _agent_for(spec, dev)
    ag.load_tensors(g["init"])
    xp = g["xp"]
    res = ag.act(xp["obs_1d"], xp.get("achieved_goal"), xp.get("desired_goal"), xp["exploit_mask"], noise=g["noise"],
                 agent_state=xp.get("agent_state"))
    action, logp, explore, exploit = res[:4]
    rep = Report(f"act golden {case}")
    if spec.gru:
        rep.check("hidden_state", res[4], g["hidden_state"])
    if spec.discrete:
        for got, key in ((action, "action"), (explore, "explore_action"), (exploit, "exploit_action")):
            assert np.array_equal(got.cpu().numpy().astype(np.int64), g[key]), key
        rep.check("log_prob", logp, g["log_prob"])
    else:
        rep.check("action", action, g["action"])
        rep.check("explore_action", explore, g["explore_action"])
        rep.check("exploit_action", exploit, g["exploit_action"])
        rep.check("log_prob", logp, g["log_prob"], extra=logp_cond(g["explore_action"]))
        m = g["xp"]["exploit_mask"].reshape(-1)
        assert torch.equal(action[torch.as_tensor(m)], exploit[torch.as_tensor(m)])
        assert torch.equal(action[torch.as_tensor(~m)], explore[torch.as_tensor(~m)])
    rep.finish()


@pytest.mark.parametrize("rows,kw", [
    (1, {}), (8, {}), (9, {}), (33, dict(goal=3)), (300, {}),
    (3, dict(img=(2, 12, 12), conv=((8, 4, 2), (16, 3, 1)), obs=4)),                                   # pixels + obs_1d
    (2, dict(img=(4, 84, 84), conv=((32, 8, 4), (64, 4, 2), (64, 3, 1)), obs=0, discrete=True, act=6)),  # config-5 frames
    (5, dict(gru="zero")),
    (5, dict(enc_hidden=(), joint_hidden=(), pi_hidden=())),            # head-only MLPs
    (12, dict(enc_hidden=(40, 24), joint_hidden=(24, 24, 16), pi_hidden=(48, 40))),
    (17, dict(discrete=True, act=5)),
])
def test_act_matches_oracle(dev, rows, kw):
    """Ragged row counts (1, one past a row chunk, many chunks), goal segments, head-only and deep
    MLPs, discrete head - against the CPU oracle on the same weights and noise."""
    from oracle import update as oup
    base = dict(obs=17, act=6, C=3, Q=2, latent=64, enc_features=48, enc_hidden=(64,), joint_hidden=(48,),
                pi_hidden=(64,), critic_hidden=(32,), T=2, B=4)
    base.update(kw)
    spec = oup.Spec(**base)
    params = oup.init_params(spec, seed=rows)
    gen = torch.Generator().manual_seed(rows)
    for k in params:   # non-zero biases
        params[k] = params[k] + 0.05 * torch.randn(params[k].shape, generator=gen)
    ag = _agent_for(spec, dev)
    ag.load_tensors(params)
    xp = {"obs_1d": torch.randn(rows, spec.obs, generator=gen)} if spec.obs else {}
    if spec.img:
        xp["obs_2d"] = torch.randint(0, 256, (rows,) + tuple(spec.img), generator=gen).float()
    if spec.gru:
        xp["agent_state"] = torch.rand(rows, spec.latent, generator=gen)
    if spec.goal:
        xp["achieved_goal"] = torch.randn(rows, spec.goal, generator=gen)
        xp["desired_goal"] = torch.randn(rows, spec.goal, generator=gen)
    xp["exploit_mask"] = (torch.rand(rows, 1, generator=gen) < 0.4)
    noise = torch.rand(rows, spec.act, generator=gen) if spec.discrete else torch.randn(rows, spec.act, generator=gen)
    want = oup.act(params, spec, xp, noise)
    got = ag.act(xp.get("obs_1d"), xp.get("achieved_goal"), xp.get("desired_goal"), xp["exploit_mask"], noise=noise,
                 obs_2d=xp.get("obs_2d"), agent_state=xp.get("agent_state"))
    rep = Report(f"act oracle rows={rows} {kw}")
    if spec.gru:
        rep.check("hidden_state", got[4], want[4])
        got, want = got[:4], want[:4]
    if spec.discrete:
        for gt, wt, key in zip(got[:1] + got[2:], want[:1] + want[2:], ("action", "explore", "exploit")):
            assert np.array_equal(gt.cpu().numpy().astype(np.int64), wt.numpy()), key
        rep.check("log_prob", got[1], want[1])
    else:
        rep.check("action", got[0], want[0])
        rep.check("explore", got[2], want[2])
        rep.check("exploit", got[3], want[3])
        rep.check("log_prob", got[1], want[1], extra=logp_cond(want[2].numpy()))
    rep.finish()


@pytest.mark.parametrize("rows,kw", [(1, {}), (9, dict(goal=3)), (6, dict(discrete=True, act=5)),
                                     (2, dict(latent=256, enc_features=256, enc_hidden=(256,), joint_hidden=(256,), pi_hidden=(256,)))])
def test_act_fused_launches_equal_layerwise(dev, rows, kw, monkeypatch):
    """act()'s default launch list folds the encoder's first layer into its skip head's launch (few input columns: every
    workgroup recomputes it) and the policy head into the actor's last layer (k_act_head_policy); FDQL_ACT_NO_FUSE=1 is one
    launch per layer + the policy kernel.  Same results up to summation order."""
    from oracle import update as oup
    base = dict(obs=17, act=6, C=3, Q=2, latent=64, enc_features=48, enc_hidden=(64,), joint_hidden=(48,),
                pi_hidden=(64,), critic_hidden=(32,), T=2, B=4)
    base.update(kw)
    spec = oup.Spec(**base)
    params = oup.init_params(spec, seed=200 + rows)
    gen = torch.Generator().manual_seed(rows)
    for k in params:
        params[k] = params[k] + 0.05 * torch.randn(params[k].shape, generator=gen)
    ag = _agent_for(spec, dev)
    ag.load_tensors(params)
    xp = {"obs_1d": torch.randn(rows, spec.obs, generator=gen)}
    if spec.goal:
        xp["achieved_goal"] = torch.randn(rows, spec.goal, generator=gen)
        xp["desired_goal"] = torch.randn(rows, spec.goal, generator=gen)
    mask = torch.rand(rows, 1, generator=gen) < 0.4
    noise = torch.rand(rows, spec.act, generator=gen) if spec.discrete else torch.randn(rows, spec.act, generator=gen)
    outs = {}
    for mode in ("fused", "layerwise"):
        if mode == "layerwise":
            monkeypatch.setenv("FDQL_ACT_NO_FUSE", "1")
        outs[mode] = [g.cpu() for g in ag.act(xp["obs_1d"], xp.get("achieved_goal"), xp.get("desired_goal"), mask, noise=noise)[:4]]
    for x, y in zip(outs["fused"], outs["layerwise"]):
        assert float((x - y).abs().max()) <= 1e-5 * max(1.0, float(y.abs().max()))


def test_act_device_noise_and_live_weights(dev):
    """Without caller noise the device draws Philox noise keyed by (seed, counter): reproducible per
    key, different across counters, N(0,1) through the tanh-Gaussian; exploit rows ignore it; and
    act() reads the arena the trainer writes (no parameter copy to refresh)."""
    from oracle import update as oup
    spec = oup.Spec(obs=17, act=6, C=3, Q=2, latent=64, enc_features=48, enc_hidden=(64,), joint_hidden=(48,),
                    pi_hidden=(64,), critic_hidden=(32,), T=2, B=4)
    ag = _agent_for(spec, dev)
    params = oup.init_params(spec, seed=3)
    ag.load_tensors(params)
    rows = 4096
    obs = torch.randn(rows, spec.obs)
    mask = torch.zeros(rows, 1, dtype=torch.bool)
    mask[::2] = True
    a1, lp1, ex1, gr1 = ag.act(obs, exploit_mask=mask, seed=5, counter=1)
    a2, *_ = ag.act(obs, exploit_mask=mask, seed=5, counter=1)
    a3, _, ex3, gr3 = ag.act(obs, exploit_mask=mask, seed=5, counter=2)
    assert torch.equal(a1, a2)
    assert torch.equal(a1[::2], a3[::2]) and torch.equal(gr1, gr3)          # greedy rows: no noise
    assert not torch.equal(ex1, ex3)
    # recover eps from the explore action: atanh(a) = mean + std * eps
    s = oup.encoder(params, spec, {"obs_1d": obs})
    logits = oup.skip_head_mlp(params, "actor_critic.actor", s, 1)
    mean, log_std = torch.chunk(logits, 2, -1)
    eps = (torch.atanh(ex1.cpu().double().clamp(-1 + 1e-9, 1 - 1e-9)) - mean.double()) / log_std.clamp(-20, 2).exp().double()
    inside = ((ex1.cpu().abs() < 0.9999) & (eps.abs() < 1.0)).double().mean()   # P(|eps| < 1) = 0.6827
    assert abs(float(inside) - 0.6827) < 0.012, float(inside)
    assert abs(float(torch.sign(eps).mean())) < 0.03
    # live weights: an in-place change of the arena is seen by the next act()
    ag.tensors["actor_critic.actor.head.bias"].add_(0.25)
    _, _, _, gr4 = ag.act(obs, exploit_mask=mask, seed=5, counter=1)
    assert not torch.equal(gr1, gr4)


# --------------------------------------------------------------------------- gradients: GPU vs oracle-f32 vs an fp64 evaluation
def _gpu_branch_pattern(ag, spec, xp_cpu):
    """Which side of every kink the GPU took: sign of each stored hidden activation (LeakyReLU keeps the sign of its
    input) and of mc - q for the lower-bound term.  Keys are oracle.update's "<prefix>.<layer>" names."""
    T, B = spec.T, spec.B
    pat = {}

    def grab(key, name, lead, width):
        pat[key] = (ag.debug(name, lead + (width,)) > 0).cpu()

    for i, w in enumerate(spec.enc_hidden):
        grab(f"encoder.visible_layer_encoders.obs_1d.{i}", f"enc_obs.h{i}", (T, B), w)
    if not spec.gru:
        for i, w in enumerate(spec.joint_hidden):
            grab(f"encoder.joiner.{i}", f"joiner.h{i}", (T, B), w)
    for i, w in enumerate(spec.pi_hidden):
        grab(f"actor_critic.actor.{i}", f"actor.h{i}", (T - 1, B), w)
    for k in range(spec.C):
        for i, w in enumerate(spec.critic_hidden):
            grab(f"actor_critic.critic.nets.{k}.{i}", f"crit{k}.h{i}", (T - 1, B), w)
            grab(f"actor_critic.critic_frozen.nets.{k}.{i}", f"crit_f{k}.h{i}", (T - 1, B), w)
    if spec.lowerbound and spec.distributional:
        pat["lowerbound"] = (xp_cpu["mc_return"][1:] - ag.debug("q_pred", (T - 1, B, spec.Nq)).cpu()) > 0
    return pat


def _f64_eval(dev, spec, params, xp, nt, na, alpha, force=None):
    """The fp64 arbiter of the three-way test: oracle.update.grads_in in float64.  dev = a GPU: the SAME Python code evaluated
    by torch on the device (torch's own float64 kernels and rocBLAS dgemm - nothing of this library); the evaluation is 2-10 s
    of CPU time otherwise, two per case, which was most of the suite's wall clock.  The float32 evaluations - the reference
    CPU path - always run on the CPU, and so does the arbiter of the default full-size case (dev = None);
    test_fp64_arbiter_on_the_device_agrees_with_the_cpu pins the two float64 evaluations to each other.
    FDQL_TEST_FP64_CPU=1: the CPU everywhere."""
    from oracle import update as oup
    if dev is None or os.environ.get("FDQL_TEST_FP64_CPU"):
        return oup.grads_in(torch.float64, spec, params, xp, nt, na, alpha, force=force)
    to = lambda d: {k: v.to(dev) for k, v in d.items()}
    with torch.device(dev):   # the oracle's factory calls (arange, eye, zeros) follow its inputs to the device
        loss, g, dq, rec, _ = oup.grads_in(torch.float64, spec, to(params), to(xp), nt.to(dev), na.to(dev), alpha,
                                           force=None if force is None else to(force))
    return loss.cpu(), {k: v.cpu() for k, v in g.items()}, dq.cpu(), {k: v.cpu() for k, v in rec.items()}, None


@pytest.mark.gpu
@pytest.mark.parametrize("T,B", [(8, 96), pytest.param(50, 256, marks=pytest.mark.skipif(
    not os.environ.get("FDQL_TEST_FP64_PIN_LARGE"), reason="the 12 544-row pin of the device-side arbiter (about 40 s of CPU float64): FDQL_TEST_FP64_PIN_LARGE=1"))])
def test_fp64_arbiter_on_the_device_agrees_with_the_cpu(dev, T, B):
    """Both float64 evaluations of the oracle - torch on the CPU and torch on the GPU - on one batch, unforced and with a
    branch pattern imposed: gradients equal to 1e-11, pre-activations to 1e-12 (max-norm relative).  The full-size shape
    (the one the B = 384 and 25-quantile three-way cases lean on) runs on request; its result of round 6 is in
    profiles/r06_fp64_arbiter_pin.txt."""
    from oracle import update as oup
    spec = oup.Spec(obs=17, act=6, C=5, Q=2, T=T, B=B)
    params = oup.init_params(spec, seed=3)
    st = oup.new_state(spec, params)
    g = torch.Generator().manual_seed(1)
    T, B = spec.T, spec.B
    xp = {"obs_1d": torch.randn(T, B, spec.obs, generator=g), "action": torch.rand(T, B, spec.act, generator=g) * 2 - 1,
          "reward": torch.randn(T, B, 1, generator=g), "mc_return": torch.randn(T, B, 1, generator=g) * 2,
          "task_done": (torch.rand(T, B, 1, generator=g) < 0.05).float(),
          "episode_step": (torch.arange(T).view(T, 1, 1) + torch.randint(0, 900, (1, B, 1), generator=g)).float()}
    nt, na = torch.randn(T - 1, B, spec.act, generator=g), torch.randn(T - 1, B, spec.act, generator=g)
    _, gc, dqc, recc, _ = oup.grads_in(torch.float64, spec, params, xp, nt, na, st.alpha)
    _, gd, dqd, recd, _ = _f64_eval(dev, spec, params, xp, nt, na, st.alpha)
    assert set(gc) == set(gd) and set(recc) == set(recd)
    rel = lambda a, b: float((a - b).abs().max() / (b.abs().max() + 1e-300))
    worst_g, worst_r = max(rel(gd[n], gc[n]) for n in gc), max(rel(recd[k].double(), recc[k].double()) for k in recc)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/fp64_arbiter_pin.txt", "a") as f:
        f.write(f"T={T} B={B} ({(T - 1) * B} gradient rows): device fp64 vs CPU fp64: gradients {worst_g:.3e}, d loss/d q {rel(dqd, dqc):.3e}, "
                f"pre-activations {worst_r:.3e} (max-norm relative)\n")
    assert max(rel(gd[n], gc[n]) for n in gc) < 1e-11 and rel(dqd, dqc) < 1e-11
    assert max(rel(recd[k].double(), recc[k].double()) for k in recc) < 1e-12
    pat = {k: (v > 0) ^ (torch.rand(v.shape, generator=g) < 1e-3) for k, v in recc.items()}   # a pattern with flips imposed
    _, gcf, dqcf, _, _ = oup.grads_in(torch.float64, spec, params, xp, nt, na, st.alpha, force=pat)
    _, gdf, dqdf, _, _ = _f64_eval(dev, spec, params, xp, nt, na, st.alpha, force=pat)
    assert max(rel(gdf[n], gcf[n]) for n in gcf) < 1e-11 and rel(dqdf, dqcf) < 1e-11


@pytest.mark.parametrize("name,kw", [
    ("config 2 full size (T=50, B=256)", dict(obs=17, act=6, C=5, Q=2, T=50, B=256)),
    ("config 4 dims (obs 376, act 17, 5x25 quantiles, T=3, B=96)", dict(obs=376, act=17, C=5, Q=25, T=3, B=96)),
    ("config 2 full size, every forward pass through the row-block chain kernel (FDQL_CHAIN=all)",
     dict(obs=17, act=6, C=5, Q=2, T=50, B=256, env={"FDQL_CHAIN": "all"})),
    ("config 2 dims at T=4, B=64 with the weight-stationary launches forced (few tiles per workgroup: prologue + flush paths)",
     dict(obs=17, act=6, C=5, Q=2, T=4, B=64, env={"FDQL_ROWGEMM": "all"})),
    ("config 2 full size without the row-block kernel (FDQL_ROWGEMM=0: k_head_dgrad + tile kernels)",
     dict(obs=17, act=6, C=5, Q=2, T=50, B=256, env={"FDQL_ROWGEMM": "0"})),
    ("config 2 full size, weight gradients without riders and without the streaming launch (FDQL_WGRAD_RIDERS=0 "
     "FDQL_STREAM_WGRAD=0: the head / action-column gradients on the tile kernels' narrow launches)",
     dict(obs=17, act=6, C=5, Q=2, T=50, B=256, env={"FDQL_WGRAD_RIDERS": "0", "FDQL_STREAM_WGRAD": "0"})),
    ("config 2 full size, every narrow weight gradient through the streaming launch (FDQL_WGRAD_RIDERS=0: head rows over "
     "state / h0 / h1, action columns with the roles swapped)",
     dict(obs=17, act=6, C=5, Q=2, T=50, B=256, env={"FDQL_WGRAD_RIDERS": "0"})),
    ("config 2 full size, the single-network dgrads (joiner / d enc / encoder) on the tile kernel instead of the row-block "
     "dgrad kernel (FDQL_ROWDGRAD=0)", dict(obs=17, act=6, C=5, Q=2, T=50, B=256, env={"FDQL_ROWDGRAD": "0"})),
    ("config 2 full size, dense weight gradients riding in the dgrad launches (FDQL_WGRAD_STAT=0)",
     dict(obs=17, act=6, C=5, Q=2, T=50, B=256, env={"FDQL_WGRAD_STAT": "0"})),
    ("temporal_len 2 at config 2 dims, B=256: the small-batch kernel on every GEMM stage of the plan",
     dict(obs=17, act=6, C=5, Q=2, T=2, B=256)),
    ("config 2 dims at T=50, B=384 (18 816 gradient rows = 294 blocks of 64: k_rowdgrad_chain beyond one round of workgroups)",
     dict(obs=17, act=6, C=5, Q=2, T=50, B=384)),
    ("25-quantile heads at config 2's full row count (obs 17, act 6, 5x25, T=50, B=256: 12 544 rows = 49 per CU): k_head_dgrad_masked "
     "with several 4 x 64-row groups per instance, k_loss_wave<2> on 12 544 waves, the masked weight-stationary dgrad at 5 narrow steps",
     dict(obs=17, act=6, C=5, Q=25, T=50, B=256)),
    ("3-layer 256-wide critics at T=20, B=64 with the weight-stationary launches forced (the shape whose fraction gate a kink flip "
     "defeats at T=5 in test_update_matches_oracle_other_configs; 1 216 gradient rows: a 12-element bias gradient's max error is not "
     "one draw of 256 terms): k_fwd3<1> forward, non-fused + fused dgrad forms",
     dict(obs=17, act=6, C=3, Q=2, T=20, B=64, critic_hidden=(256, 256, 256), env={"FDQL_ROWGEMM": "all"})),
    ("config 2 dims at T=50, B=64: one rank's share at N = 4 on the small-block kernels (k_fwd3<1>: 200 forward blocks of 16 rows, "
     "k_rowdgrad_chain<1>: 196 blocks of 16)", dict(obs=17, act=6, C=5, Q=2, T=50, B=64)),
    ("config 4 dims at T=50, B=32 (1 600 rows: k_fwd3<1> with the 376 observation columns through the 16-byte request ring, "
     "34 actor outputs = three head tiles; k_rowdgrad_chain<1>; 1 568 x 125 atoms: two critics' first layers hold a unit on its kink, "
     "which is why this size is compared here and not by the fraction gate)", dict(obs=376, act=17, C=5, Q=25, T=50, B=32)),
    ("config 3 dims at T=50, B=64: three observation segments (obs 28 + two goals of 10 = 48 ragged columns: three 16-k groups) "
     "through k_fwd3<1>, k_rowdgrad_chain<1> behind d state", dict(obs=28, goal=10, act=6, C=5, Q=2, T=50, B=64)),
    ("config 4 dims (5x25 quantiles, 17 action columns) at T=6, B=64 with the stationary and streaming launches forced "
     "(FDQL_ROWGEMM=all FDQL_STREAM_WGRAD=2: 25 head rows and 17 input columns per streaming problem, no riders fit)",
     dict(obs=376, act=17, C=5, Q=25, T=6, B=64, env={"FDQL_ROWGEMM": "all", "FDQL_STREAM_WGRAD": "2"})),
])
def test_gradient_parity_three_way_fp64(dev, name, kw, monkeypatch):
    """north_star: gradients within 1e-5 rel fp32 of the reference CPU path.  Two fp32 evaluations of this loss cannot
    agree to 1e-5 element by element: LeakyReLU and relu(mc - q) make the gradient discontinuous in the activations, and
    a unit whose pre-activation is ~1e-7 from its kink picks its branch by rounding.  An fp64 evaluation of the oracle
    is the arbiter:
      (a) every unit where the GPU's branch differs from the fp64 one has an fp64 pre-activation within KINK of zero
          (relative to the largest pre-activation of its layer): the GPU picked the other side of a kink it sits on;
      (b) with exactly those branch choices imposed on the CPU evaluations (fp64 and the fp32 oracle), EVERY element of
          every gradient tensor and of d loss / d q_pred satisfies
              |gpu - f64| <= 2 |oracle_f32 - f64| + 1e-5 max|f64|
          (max-norm-relative, like every bound of this file; at most one element in 10^4 may miss it), and per tensor
          max|gpu - f64| <= max(1e-5 max|f64|, 1.25 max|oracle_f32 - f64|) (factor 2 for tensors of fewer than 64 elements,
          where the maximum is the element-wise quantity): the GPU is within 1e-5, or - where the
          reference's own fp32 formula is ill-conditioned (log(1 - tanh^2 + 1e-4) and its derivative at saturated
          actions) - no farther from fp64 than the CPU path's own fp32 evaluation is.
    The table also lists the unforced comparison.  It goes to gpurun_out/parity_three_way.txt (committed as
    profiles/r02_parity_report.txt)."""
    import os
    from oracle import update as oup
    kw = dict(kw)
    for k, v in kw.pop("env", {}).items():
        monkeypatch.setenv(k, v)
    spec = oup.Spec(**kw)
    T, B = spec.T, spec.B
    params = oup.init_params(spec, seed=3)
    st = oup.new_state(spec, params)
    g = torch.Generator().manual_seed(1)
    xp = {"obs_1d": torch.randn(T, B, spec.obs, generator=g), "action": torch.rand(T, B, spec.act, generator=g) * 2 - 1,
          "reward": torch.randn(T, B, 1, generator=g), "mc_return": torch.randn(T, B, 1, generator=g) * 2,
          "task_done": (torch.rand(T, B, 1, generator=g) < 0.05).float(),
          "episode_step": (torch.arange(T).view(T, 1, 1) + torch.randint(0, 900, (1, B, 1), generator=g)).float()}
    xp["episode_step"][T // 2:, ::7] = torch.arange(T - T // 2).view(-1, 1, 1).float()
    nt, na = torch.randn(T - 1, B, spec.act, generator=g), torch.randn(T - 1, B, spec.act, generator=g)
    if spec.goal:   # (drawn last: the cases without goals keep the draws they always had)
        xp["achieved_goal"] = torch.randn(T, B, spec.goal, generator=g)
        xp["desired_goal"] = torch.randn(T, B, spec.goal, generator=g)
    ag = _agent_for(spec, dev)
    ag.load_tensors(params)
    ag.update({k: v.to(dev) for k, v in xp.items()}, nt.to(dev), na.to(dev), phase=1)     # FDQL_PHASE_GRAD: weights untouched
    pat = _gpu_branch_pattern(ag, spec, xp)
    g_gpu = {n: ag.grad_views[n].cpu().double() for n in ag.trainable}
    g_gpu["d loss / d q_pred"] = ag.debug("dz", (T - 1, B, spec.Nq)).cpu().double()
    # the unforced CPU evaluations depend on the spec alone (fixed seeds): shared by the cases that differ only in plan switches
    ckey = repr(sorted(kw.items()))
    # the arbiter of the default full-size case is evaluated on the CPU, the others' by torch on the GPU (_f64_eval)
    dev64 = None if name == "config 2 full size (T=50, B=256)" else dev
    if ckey not in _THREE_WAY_CACHE:
        _, g32, dq32, rec32, _ = oup.grads_in(torch.float32, spec, params, xp, nt, na, st.alpha)
        _, g64, dq64, rec64, _ = _f64_eval(dev64, spec, params, xp, nt, na, st.alpha)
        while len(_THREE_WAY_CACHE) >= 3:   # (three specs at most: a full-size case holds ~1 GB of pre-activations)
            _THREE_WAY_CACHE.pop(next(iter(_THREE_WAY_CACHE)))
        _THREE_WAY_CACHE[ckey] = (g32, dq32, rec32, g64, dq64, rec64)
    g32, dq32, rec32, g64, dq64, rec64 = (dict(v) if isinstance(v, dict) else v for v in _THREE_WAY_CACHE[ckey])
    _, g64f, dq64f, _, _ = _f64_eval(dev64, spec, params, xp, nt, na, st.alpha, force=pat)
    _, g32f, dq32f, _, _ = oup.grads_in(torch.float32, spec, params, xp, nt, na, st.alpha, force=pat)
    for d, dq in ((g32, dq32), (g64, dq64), (g64f, dq64f), (g32f, dq32f)):
        d["d loss / d q_pred"] = dq
    KINK, TOL_G = 2e-6, 1e-5
    lines = [f"== {name}: gradients, GPU vs CPU oracle (fp32) vs fp64 evaluation of the oracle",
             "branch flips (units whose LeakyReLU / relu(mc-q) branch differs from the fp64 evaluation):",
             f"{'kink set':58s} {'units':>10s} {'gpu flips':>9s} {'max |pre64|/max|pre|':>21s} {'oracle-f32 flips':>16s}"]
    bad = []
    for key, p64 in rec64.items():
        if key not in pat:
            continue
        b64 = p64 > 0
        flips = pat[key].reshape(b64.shape) != b64
        f32 = (rec32[key] > 0) != b64
        scale = float(p64.abs().max())
        dist = float(p64[flips].abs().max() / scale) if bool(flips.any()) else 0.0
        lines.append(f"{key:58s} {b64.numel():10d} {int(flips.sum()):9d} {dist:21.3e} {int(f32.sum()):16d}")
        if dist > KINK:
            bad.append(("kink distance", key, dist))
    lines.append("max-norm-relative errors; 'forced' = the GPU's branch choices imposed on the CPU evaluation")
    lines.append(f"{'tensor':66s} {'oracle32-f64':>12s} {'gpu-f64':>10s} {'o32-f64 forced':>14s} {'gpu-f64 forced':>14s} "
                 f"{'>1e-5':>6s} {'viol.(b)':>8s}")
    by_1e5, by_cpu_clause = [], []   # which clause each tensor passes by (so "within 1e-5" is never read more broadly than it holds)
    for n in g_gpu:
        ref, reff = g64[n].double(), g64f[n].double()
        sc, scf = float(ref.abs().max()) + 1e-300, float(reff.abs().max()) + 1e-300
        e_or = (g32[n].double() - ref).abs()
        e_gpu = (g_gpu[n].reshape(ref.shape) - ref).abs()
        e_orf = (g32f[n].double() - reff).abs()
        e_f = (g_gpu[n].reshape(ref.shape) - reff).abs()
        n_out = int((e_f > TOL_G * scf).sum())
        viol = int((e_f > 2 * e_orf + TOL_G * scf).sum())
        lines.append(f"{n:66s} {float(e_or.max()) / sc:12.3e} {float(e_gpu.max()) / sc:10.3e} {float(e_orf.max()) / scf:14.3e} "
                     f"{float(e_f.max()) / scf:14.3e} {n_out:6d} {viol:8d}")
        # (b) element by element, allowing one element in 10^4 (an element where the CPU path's own error happens to
        # vanish leaves the bound at 1e-5 while the tensor's fp32 error level is far above it), and in the max norm
        if viol > 1e-4 * e_f.numel():
            bad.append(("(b) violated element-wise", n, viol, e_f.numel()))
        # (tensors of a few elements - log_alpha, the bias of a 12-output head: the maximum over so few draws IS the element-wise
        # quantity, so the factor of the element-wise clause applies; two fp32 evaluations of one ill-conditioned scalar differ
        # by a factor of two either way - round 6: d loss / d log_alpha 7.4e-5 against the CPU path's own 4.8e-5 once the
        # forward kernel's summation order changed)
        cpu_factor = 2.0 if e_f.numel() < 64 else 1.25
        if float(e_f.max()) / scf > max(TOL_G, cpu_factor * float(e_orf.max()) / scf):
            bad.append(("(b) violated in the max norm", n, float(e_f.max()) / scf, float(e_orf.max()) / scf))
        (by_1e5 if float(e_f.max()) / scf <= TOL_G else by_cpu_clause).append(f"{n} ({float(e_f.max()) / scf:.1e} vs CPU fp32 {float(e_orf.max()) / scf:.1e})")
    lines.append(f"SUMMARY {name}: {len(by_1e5)} of {len(by_1e5) + len(by_cpu_clause)} tensors within 1e-5 of fp64 in the max norm; "
                 + ("the others pass by the 'no farther from fp64 than 1.25 x the CPU path's own fp32 error' clause: " + "; ".join(by_cpu_clause)
                    if by_cpu_clause else "none needs the CPU-error clause"))
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/parity_three_way.txt", "a") as f:
        f.write("\n".join(lines) + "\n")
    assert not bad, bad[:8]


@pytest.mark.gpu
@pytest.mark.parametrize("T,B", [(2, 256), (6, 64)])
def test_graph_replay_matches_eager_launches(dev, T, B, monkeypatch):
    """With FDQL_GRAPH=1 (read at create; the default is the eager launch list) fdql_agent_update replays one hipGraph
    per launch plan from its second use on (agent.hip capture_update).  Same weights, same persistent batch tensors,
    device-drawn noise: the two must stay bit-identical over several steps, the replayed one through different batches refilled in place and
    through a second plan (another set of batch pointers) and back."""
    from oracle import update as oup
    spec = oup.Spec(obs=17, act=6, C=5, Q=2, latent=64, enc_features=64, enc_hidden=(64,), joint_hidden=(64,),
                    pi_hidden=(64,), critic_hidden=(64, 64), T=T, B=B)
    params = oup.init_params(spec, seed=5)
    agents = []
    for mode in ("0", "1"):
        monkeypatch.setenv("FDQL_GRAPH", mode)
        ag = _agent_for(spec, dev)
        ag.load_tensors(params)
        agents.append(ag)
    g = torch.Generator().manual_seed(2)

    def batch():
        return {"obs_1d": torch.randn(T, B, 17, generator=g), "action": torch.rand(T, B, 6, generator=g) * 2 - 1,
                "reward": torch.randn(T, B, 1, generator=g), "mc_return": torch.randn(T, B, 1, generator=g) * 2,
                "task_done": (torch.rand(T, B, 1, generator=g) < 0.05).float(),
                "episode_step": (torch.arange(T).view(T, 1, 1) + torch.randint(0, 900, (1, B, 1), generator=g)).float()}
    pools = [[{k: v.to(dev) for k, v in batch().items()} for _ in range(2)] for _ in agents]   # two pointer sets each
    order = [0, 0, 0, 1, 1, 0, 1, 0]
    for step, which in enumerate(order):
        fresh = batch()
        for ag, pool in zip(agents, pools):
            for k, v in fresh.items():
                pool[which][k].copy_(v)
            ag.update(pool[which], seed=11)
        torch.cuda.synchronize()
        for k in agents[0].tensors:
            assert torch.equal(agents[0].tensors[k], agents[1].tensors[k]), (step, k)
    s0, s1 = agents[0].stats(), agents[1].stats()
    assert s0["graph_launches"] == 0 and s0["plans_built"] == 2
    assert s1["plans_built"] == 2 and s1["graph_launches"] == len(order) - 2      # first use of each plan is eager
    sc0, sc1 = agents[0].scalars(), agents[1].scalars()
    assert sc0 == sc1 and sc0["step"] == len(order)


@pytest.mark.gpu
@pytest.mark.parametrize("wstat", ["1", "presum"])
@pytest.mark.parametrize("form", ["fwd", "fwd_hf", "fwd_minor_hf", "dgrad", "dgrad_fused", "dual_hf", "fwd_wide", "dual_wide", "dgrad_wide"])
@pytest.mark.parametrize("M,ninst", [(64, 1), (192, 3), (1280, 15), (12544, 10)])
def test_rowgemm_forms(dev, form, M, ninst, wstat, monkeypatch):
    """The weight-stationary row-block kernel (csrc/wstat.hip) through fdql_test_rowgemm against fp64 torch: every form (forward with /
    without head fusion and a narrow extra input block, two-output forward, dgrad with the LeakyReLU' gate and column
    sums, dgrad with the head dgrad of the layer above fused into its loader), for one tile per workgroup, a few, more
    tiles than workgroups (the software-pipelined path) and the update's own size."""
    from fastdeepqlearning_amd import _native as nat
    presum = wstat == "presum"   # the weight-stationary kernel summing a tile's head planes itself (WsArgs::hf_presum, the update's default)
    if presum:
        if form not in ("fwd_hf", "fwd_minor_hf", "dual_hf"):
            pytest.skip("presum concerns the head-fusion forms")
        wstat = "1"
    wide = form.endswith("_wide")       # narrow blocks of 17 / 25 columns (config 4: act 17, 25 quantiles): 3 / 4 steps of 8
    lib = nat.load(); st = nat.current_stream(dev)
    g = torch.Generator().manual_seed(7 + M + ninst)
    rnd = lambda *s: torch.randn(*s, generator=g)
    R, Q, k1 = M * ninst, (25 if form == "dgrad_wide" else 2), (17 if wide else 6)
    ks = form.startswith("dgrad")
    dual, fused = form in ("dual_hf", "dual_wide"), form == "dgrad_fused"
    bmm = lambda x, w, k: torch.bmm(x.double().view(ninst, M, k), w.double())
    A0, A1, A2, W1, W2, bias, ref, hfw, fzh, fzw = (None,) * 10
    if ks:
        W0 = rnd(ninst, 256, 256) / 16                       # [k][n]
        A1, W1, ref = rnd(R, Q), rnd(ninst, Q, 256), rnd(R, 256)     # dY and the head's rows over this layer's columns
        if fused:
            fzh, fzw = rnd(R, 256), rnd(ninst, Q, 300)
            pre1 = bmm(A1, fzw[:, :, :256], Q)
            a0 = torch.where(fzh.double().view(ninst, M, 256) > 0, pre1, 0.01 * pre1)
            A0 = torch.full((R, 256), float("nan"))
        else:
            A0 = rnd(R, 256)
            a0 = A0.double().view(ninst, M, 256)
        pre = torch.bmm(a0, W0.double()) + bmm(A1, W1, Q)
        want = torch.where(ref.double().view(ninst, M, 256) > 0, pre, 0.01 * pre)
    else:
        A0, W0, bias = rnd(R, 256), rnd(ninst, 256, 256) / 16, rnd(ninst, 256)      # W0 [n][k]
        pre = bmm(A0, W0.transpose(1, 2), 256) + bias.double()[:, None, :]
        if form in ("fwd_minor_hf", "dual_hf", "fwd_wide", "dual_wide"):
            A1, W1 = rnd(R, k1), rnd(ninst, 256, k1)
            pre = pre + bmm(A1, W1.transpose(1, 2), k1)
        want = torch.nn.functional.leaky_relu(pre, 0.01)
        if dual:
            A2, W2 = rnd(R, k1), rnd(ninst, 256, k1)
            want2 = torch.nn.functional.leaky_relu(pre + bmm(A2, W2.transpose(1, 2), k1), 0.01)
        if form != "fwd" and not wide:
            hfw = rnd(ninst, Q, 300)
    d = lambda t: None if t is None else t.to(dev).contiguous()
    A0_d, A1_d, A2_d, W0_d, W1_d, W2_d, bias_d, ref_d, hfw_d, fzh_d, fzw_d = map(d, (A0, A1, A2, W0, W1, W2, bias, ref, hfw, fzh, fzw))
    C = torch.full((R, 256), float("nan"), device=dev)
    C2 = torch.full((R, 256), float("nan"), device=dev) if dual else None
    cs = torch.full((ninst, M // 64, 256), float("nan"), device=dev) if ks else None
    fcs = torch.full((ninst, M // 64, 256), float("nan"), device=dev) if fused else None
    hfo = torch.full((ninst, 8, M, Q), float("nan"), device=dev) if hfw is not None else None
    hfo2 = torch.full((ninst, 8, M, Q), float("nan"), device=dev) if dual and hfw is not None else None
    rc = lib.fdql_test_rowgemm(nat.ptr(A0_d), nat.ptr(A1_d), 0 if A1 is None else A1.shape[1], nat.ptr(A2_d), 0 if A2 is None else k1,
                               nat.ptr(W0_d), 256, nat.ptr(W1_d), nat.ptr(W2_d), nat.ptr(bias_d), nat.ptr(C), nat.ptr(C2), nat.ptr(ref_d),
                               nat.ptr(cs), nat.ptr(hfw_d), 300, Q if hfw is not None else 0, nat.ptr(hfo), nat.ptr(hfo2), M, ninst,
                               int(ks), int(ks), int(dual), -8 if presum else 8, nat.ptr(fzh_d), nat.ptr(fzw_d), 300, nat.ptr(fcs), st)   # (planes < 0: summed in the kernel)
    assert rc == 0, lib.fdql_last_error().decode()
    torch.cuda.synchronize()
    close = lambda got, ref_, tol: float((got.double().cpu().reshape(ref_.shape) - ref_).abs().max()) <= tol * float(ref_.abs().max())
    assert close(C, want, 2e-5)
    if dual:
        assert close(C2, want2, 2e-5)
    # column sums: one partial row per workgroup of the instance (the rest of the buffer cleared by the hook): the partial rows
    # must add up to the column sums
    if ks:
        assert close(cs.sum(1), want.view(ninst, M, 256).sum(1), 1e-4)
    if fused:
        assert close(A0_d, a0, 2e-5)
        assert close(fcs.sum(1), a0.view(ninst, M, 256).sum(1), 1e-4)
    if hfw is not None:
        w = hfw[:, :, :256].double().view(ninst, Q, 8, 32)
        if presum:   # plane 0 holds the sum over the eight column planes, the others were cleared by the hook
            assert close(hfo[:, 0], torch.einsum("impc,iqpc->imq", want.view(ninst, M, 8, 32), w), 1e-4)
            assert float(hfo[:, 1:].abs().max()) == 0.0
            if dual:
                assert close(hfo2[:, 0], torch.einsum("impc,iqpc->imq", want2.view(ninst, M, 8, 32), w), 1e-4)
        else:
            assert close(hfo, torch.einsum("impc,iqpc->ipmq", want.view(ninst, M, 8, 32), w), 1e-4)
            if dual:
                assert close(hfo2, torch.einsum("impc,iqpc->ipmq", want2.view(ninst, M, 8, 32), w), 1e-4)
    if wide and not ks:
        assert float(want.abs().max()) > 1.0   # (the wide narrow block contributes: 17 columns of unit-variance weights)


@pytest.mark.gpu
@pytest.mark.parametrize("M,nprob,ldw,nslab", [(64, 1, 256, 4), (12544, 15, 256, 32), (2048, 5, 262, 32), (320, 3, 273, 8)])
def test_wgrad_stat_blocks(dev, M, nprob, ldw, nslab):
    """The output-stationary weight-gradient kernel (csrc/wgrad.hip) against fp64 torch: dW[i] = G[i]^T X[i] over M rows for
    nprob 256 x 256 blocks, written as K-split slabs (one partial per workgroup of a block, the other slabs cleared) into
    weight tensors of row pitch 256 / 262 (critic layer 0) / 273 (encoder head, 4-byte aligned rows): the slab sum is the
    gradient, everything outside the blocks' columns stays untouched."""
    from fastdeepqlearning_amd import _native as nat
    lib = nat.load(); st = nat.current_stream(dev)
    g = torch.Generator().manual_seed(M + nprob)
    G = torch.randn(nprob * M, 256, generator=g)
    X = torch.randn(nprob * M, 256, generator=g)
    stride = nprob * 256 * ldw + 8
    slabs = torch.full((nslab, stride), 7.0, device=dev)
    G_d, X_d = G.to(dev), X.to(dev)
    rc = lib.fdql_test_wgrad_stat(nat.ptr(G_d), nat.ptr(X_d), nat.ptr(slabs), M, nprob, ldw, nslab, stride, st)
    assert rc == 0, lib.fdql_last_error().decode()
    torch.cuda.synchronize()
    got = slabs[:, :nprob * 256 * ldw].double().cpu().view(nslab, nprob, 256, ldw)
    want = torch.bmm(G.double().view(nprob, M, 256).transpose(1, 2), X.double().view(nprob, M, 256))
    err = float((got[..., :256].sum(0) - want).abs().max() / want.abs().max())
    assert err < 2e-5, err
    assert bool((got[..., 256:] == 7.0).all()) and bool((slabs[:, nprob * 256 * ldw:] == 7.0).all())   # nothing outside the blocks


@pytest.mark.gpu
@pytest.mark.parametrize("M,nprob,nx2,x2_every,ng2,nslab", [(64, 1, 6, 1, 2, 4), (12544, 15, 6, 3, 2, 32), (2048, 4, 8, 2, 4, 32),
                                                         (320, 3, 1, 1, 1, 8), (1024, 5, 3, 2, 0, 16), (1024, 5, 0, 0, 3, 16)])
def test_wgrad_stat_riders(dev, M, nprob, nx2, x2_every, ng2, nslab):
    """Riders of the output-stationary weight-gradient kernel (csrc/wgrad.h): next to dW[i] = G[i]^T X[i], block i (every
    x2_every-th) also yields the gradient of a few extra INPUT columns, dW2[i] = G[i]^T X2[i] (a critic's action columns:
    columns 256.. of its layer-0 weight, row pitch 262), and every block the gradient of a few extra OUTPUT rows over the same
    input, dW3[i] = G2[i]^T X[i] (the skip head's rows over this layer's input: row pitch 774) - as v_mfma_f32_4x4x1 on the
    operand registers of the main block.  fp64 torch is the reference; slabs hold partials or zeros, nothing else is touched."""
    from fastdeepqlearning_amd import _native as nat
    lib = nat.load(); st = nat.current_stream(dev)
    g = torch.Generator().manual_seed(M + 7 * nprob + nx2)
    G = torch.randn(nprob * M, 256, generator=g)
    X = torch.randn(nprob * M, 256, generator=g)
    ldw, ldx2, ldg2, ldw3 = (262 if nx2 <= 6 else 265), max(nx2, 1), max(ng2, 1) + 1, 774
    X2 = torch.randn(nprob * M, ldx2, generator=g)
    G2 = torch.randn(nprob * M, ldg2, generator=g)
    # one arena per slab: [dense blocks incl. the action columns | head rows]
    n_dense, n_head = nprob * 256 * ldw, nprob * max(ng2, 1) * ldw3
    stride = n_dense + n_head + 8
    slabs = torch.full((nslab, stride), 7.0, device=dev)
    G_d, X_d, X2_d, G2_d = G.to(dev), X.to(dev), X2.to(dev), G2.to(dev)
    base = slabs.data_ptr()
    rc = lib.fdql_test_wgrad_stat_riders(nat.ptr(G_d), nat.ptr(X_d), base, M, nprob, ldw, nslab, stride,
                                         nat.ptr(X2_d) if nx2 else None, nx2, ldx2, base + 4 * 256, ldw, x2_every,
                                         nat.ptr(G2_d) if ng2 else None, ng2, ldg2, base + 4 * n_dense, ldw3, st)
    assert rc == 0, lib.fdql_last_error().decode()
    torch.cuda.synchronize()
    tot = slabs.double().sum(0).cpu()
    dense = tot[:n_dense].view(nprob, 256, ldw)
    head = tot[n_dense:n_dense + n_head].view(nprob, max(ng2, 1), ldw3)
    Gd, Xd = G.double().view(nprob, M, 256), X.double().view(nprob, M, 256)
    want = torch.bmm(Gd.transpose(1, 2), Xd)
    assert float((dense[..., :256] - want).abs().max() / want.abs().max()) < 2e-5
    untouched = 7.0 * nslab
    for i in range(nprob):
        if nx2 and i % x2_every == 0:
            w2 = Gd[i].T @ X2.double().view(nprob, M, ldx2)[i][:, :nx2]
            assert float((dense[i, :, 256:256 + nx2] - w2).abs().max() / w2.abs().max()) < 2e-5, i
            assert bool((dense[i, :, 256 + nx2:] == untouched).all())
        else:
            assert bool((dense[i, :, 256:] == untouched).all())
    if ng2:
        w3 = torch.bmm(G2.double().view(nprob, M, ldg2)[:, :, :ng2].transpose(1, 2), Xd)
        assert float((head[:, :ng2, :256] - w3).abs().max() / w3.abs().max()) < 2e-5
        assert bool((head[:, :, 256:] == untouched).all())
    else:
        assert bool((head == untouched).all())
    assert bool((tot[n_dense + n_head:] == untouched).all())
    # (a slab the kernel left alone inside a block or a rider would put 7.0 into the sums checked above)


@pytest.mark.gpu
@pytest.mark.parametrize("kw,want,absent", [
    (dict(obs=17, act=6, T=50, B=32), ["fwd3<16>:", "rowdchain:", "wstat<1,1,2>:critics.fwd0", "wstatg<1,0,1>:critics.dpre1+0"], ["chain:", "k:dstate.sum"]),
    (dict(obs=17, act=6, T=50, B=64), ["fwd3<16>:", "rowdchain:"], ["chain:", "k:dstate.sum"]),
    (dict(obs=17, act=6, T=50, B=128), ["fwd3<32>:", "rowdchain:", "wgstat:"], ["chain:", "k:dstate.sum"]),
    (dict(obs=17, act=6, T=50, B=256), ["fwd3<32>:", "rowdchain:", "wgstat:"], ["chain:"]),
    (dict(obs=17, act=6, T=2, B=256), ["fwd3<16>:", "k:dstate.sum"], ["rowdchain:", "gemmsmall:enc_obs.fwd0"]),
    (dict(obs=17, act=6, T=50, B=384), ["chain:enc_joiner_actors", "rowdchain:"], ["fwd3"]),
    (dict(obs=376, act=17, Q=25, T=50, B=32), ["fwd3<16>:", "rowdchain:"], ["chain:"]),
    (dict(obs=28, goal=10, act=6, T=50, B=128), ["fwd3<32>:", "rowdchain:"], ["chain:"]),
])
def test_default_plans_run_the_small_block_kernels(dev, kw, want, absent):
    """Which kernels a default plan launches is a measured decision; a rule edited by accident would not fail
    any parity test - the launches it falls back to are parity-green too - it would only be slower.  The launch names of
    fdql_agent_profile_update pin the round-6 choices: k_fwd3 on 16- / 32-row blocks up to one dispatch round, the dgrad chain on
    small blocks from 1 024 rows, the critics' layer 0 weight-stationary from 224 tiles."""
    from fastdeepqlearning_amd.core import NativeAgent, make_config
    kw = dict(kw)
    T, B, obs, act, goal, Q = kw["T"], kw["B"], kw["obs"], kw["act"], kw.get("goal", 0), kw.get("Q", 2)
    ag = NativeAgent(make_config(obs, act, T, B, goal_dim=goal, n_critics=5, n_quantiles=Q), dev)
    ag.init_weights(0)
    g = torch.Generator(device=dev).manual_seed(0)
    xp = {"obs_1d": torch.randn(T, B, obs, device=dev, generator=g), "action": torch.rand(T, B, act, device=dev, generator=g) * 2 - 1,
          "reward": torch.randn(T, B, 1, device=dev, generator=g), "mc_return": torch.randn(T, B, 1, device=dev, generator=g),
          "task_done": torch.zeros(T, B, 1, device=dev), "episode_step": torch.arange(T, device=dev, dtype=torch.float32).view(T, 1, 1).expand(T, B, 1).contiguous()}
    if goal:
        xp["achieved_goal"] = torch.randn(T, B, goal, device=dev, generator=g)
        xp["desired_goal"] = torch.randn(T, B, goal, device=dev, generator=g)
    names = [r[0] for r in ag.profile_update(xp, seed=1)]
    for w in want:
        assert any(n.startswith(w) for n in names), (w, names)
    for w in absent:
        assert not any(n.startswith(w) for n in names), (w, names)
    assert np.isfinite(ag.scalars()["loss"])
