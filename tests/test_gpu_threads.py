"""The Runner's threading model against the native handles (reference: franQ/Replay/async_replay_memory.py:55-70 -
an add thread beside a sample thread on one shard; franQ/Runner/runner.py:177-191 - one `_replay_handler` thread per
shard; franQ/Agent/deepQlearning.py:83-94 - the trainer loop).  ctypes drops the GIL around every C call, so only the
mutex inside the handles keeps these threads apart."""
import threading
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _record(code, step, done):
    """A record whose every field is a function of `code`: a torn row (fields from two records) cannot go unnoticed."""
    c = np.float32(code)
    return {"obs_1d": np.full(5, c, np.float32), "action": np.full(3, c, np.float32) * np.float32(1e-7),
            "reward": float(c), "mc_return": float(c) * 0.5, "task_done": False, "episode_done": bool(done), "episode_step": int(step),
            "idx": float(c)}


def _check_rows(rows, expect_codes):
    """rows: dict key -> [n, dim] numpy of ring slots.  Every slot holds exactly one record, untorn."""
    code = rows["idx"][:, 0]
    assert np.array_equal(rows["obs_1d"], np.repeat(code[:, None], 5, 1)), "torn row: obs_1d"
    assert np.array_equal(rows["action"], np.repeat(code[:, None], 3, 1) * np.float32(1e-7)), "torn row: action"
    assert np.array_equal(rows["reward"][:, 0], code), "torn row: reward"
    assert np.array_equal(rows["mc_return"][:, 0], code * np.float32(0.5)), "torn row: mc_return"
    if expect_codes is not None:
        got = np.sort(code.astype(np.int64))
        assert np.array_equal(got, np.sort(np.asarray(expect_codes, np.int64))), "lost or duplicated records"


def test_two_writers_and_async_trainer_share_one_shard(dev):
    """use_async_train=True (the reference default, conf.py:73): the trainer thread samples and updates for >= 2000
    steps while one thread add()s records one by one and another appends whole episodes to the SAME shard.  No row is
    torn, no record is lost, and len / top are what the same number of serial adds gives."""
    from fastdeepqlearning_amd import Agent
    from fastdeepqlearning_amd.Replay import ReplayMemory
    from test_gpu_facade import _conf
    conf = _conf(dev, T=4, B=16)
    conf.use_async_train = True
    conf.param_update_interval = 50
    maxlen = 1 << 17
    shard = ReplayMemory(maxlen, conf.batch_size, conf.temporal_len, device=dev, seed=3)
    agent = Agent.make(conf)
    n_a, ep_len, n_eps = 24_000, 40, 500
    errors = []

    def writer_add():          # per-record path: pinned staging, flushed by whoever samples or fills it
        try:
            for i in range(n_a):
                shard.add(_record(i, i % 50, i % 50 == 49))
        except Exception as e:   # noqa: BLE001
            errors.append(e)

    def writer_episodes():     # whole-episode path: its own staging + device kernels + scatter
        try:
            for e in range(n_eps):
                recs = [_record(1_000_000 + e * ep_len + j, j, j == ep_len - 1) for j in range(ep_len)]
                shard.append_episode(recs)
        except Exception as e:   # noqa: BLE001
            errors.append(e)

    # enough rows for the first sample, then everything concurrently
    for i in range(200):
        shard.add(_record(2_000_000 + i, i % 50, i % 50 == 49))
    agent.enable_training([shard])
    threads = [threading.Thread(target=writer_add), threading.Thread(target=writer_episodes)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    deadline = time.time() + 120
    while agent.iteration < 2000 and time.time() < deadline and agent.trainer_error is None:
        time.sleep(0.01)
    agent.disable_training()
    assert not errors, errors
    assert agent.trainer_error is None, agent.trainer_error
    assert agent.iteration >= 2000, agent.iteration
    total = 200 + n_a + n_eps * ep_len
    assert total < maxlen
    assert len(shard) == total and shard._ring.top == total           # the serial result (no wrap)
    got = shard[np.arange(total)]
    rows = {k: v.cpu().numpy().reshape(total, -1) for k, v in got.items()}
    expect = np.concatenate([2_000_000 + np.arange(200), np.arange(n_a), 1_000_000 + np.arange(n_eps * ep_len)])
    _check_rows(rows, expect)
    # an episode appended in one call occupies consecutive slots in order
    code = rows["idx"][:, 0].astype(np.int64)
    ep = code >= 1_000_000
    ep &= code < 2_000_000
    pos = np.nonzero(ep)[0]
    starts = pos[(code[pos] - 1_000_000) % ep_len == 0]
    for s in starts[:: max(1, len(starts) // 50)]:
        assert np.array_equal(code[s:s + ep_len], code[s] + np.arange(ep_len))
    sc = agent.native.scalars()
    assert np.isfinite(sc["loss"])
    assert agent.native.stats()["plans_built"] <= 4                    # the facade's sample buffers recur


def test_writers_wrap_the_ring_while_sampling(dev):
    """Same with a small ring that wraps many times under a sampler thread: rows stay whole, len / top follow
    replay_memory.py:45-46 (len caps at maxlen - 1, quirk q1)."""
    from fastdeepqlearning_amd.Replay import ReplayMemory
    maxlen, n_a, n_b = 3001, 30_000, 20_000
    shard = ReplayMemory(maxlen, 32, 8, device=dev, seed=1)
    for i in range(100):
        shard.add(_record(5_000_000 + i, i % 50, False))
    stop = threading.Event()
    errors, samples = [], [0]

    def sampler():
        try:
            while not stop.is_set():
                xp = shard.temporal_sample()
                if samples[0] % 64 == 0:     # sampled rows are whole records too (the kernel runs after the scatter)
                    flat = {k: v.reshape(-1, v.shape[-1]).cpu().numpy() for k, v in xp.items()}
                    _check_rows(flat, None)
                samples[0] += 1
        except Exception as e:   # noqa: BLE001
            errors.append(e)

    def writer(base, n, bulk):
        try:
            if bulk:   # packed rows, 37 at a time (crosses staging and wrap boundaries at odd places)
                keys = shard._keys
                for i0 in range(0, n, 37):
                    m = min(37, n - i0)
                    rows = np.zeros((m, int(shard._offsets[-1])), np.float32)
                    for r in range(m):
                        rec = _record(base + i0 + r, 0, False)
                        for j, k in enumerate(keys):
                            rows[r, shard._offsets[j]:shard._offsets[j + 1]] = np.asarray(rec[k], np.float32).reshape(-1)
                    shard.add_rows(rows)
            else:
                for i in range(n):
                    shard.add(_record(base + i, i % 50, False))
        except Exception as e:   # noqa: BLE001
            errors.append(e)

    ts = [threading.Thread(target=sampler), threading.Thread(target=writer, args=(0, n_a, False)),
          threading.Thread(target=writer, args=(1_000_000, n_b, True))]
    for t in ts:
        t.start()
    for t in ts[1:]:
        t.join()
    stop.set()
    ts[0].join()
    assert not errors, errors
    assert samples[0] > 0
    total = 100 + n_a + n_b
    assert len(shard) == maxlen - 1 and shard._ring.top == total % maxlen
    got = shard[np.arange(maxlen)]
    rows = {k: v.cpu().numpy().reshape(maxlen, -1) for k, v in got.items()}
    _check_rows(rows, None)
    # the ring holds the newest maxlen records of the interleaved stream: per writer, its newest ones, contiguous
    code = rows["idx"][:, 0].astype(np.int64)
    a = np.sort(code[code < 1_000_000])
    b = np.sort(code[(code >= 1_000_000) & (code < 5_000_000)])
    assert len(a) + len(b) == maxlen
    assert np.array_equal(a, np.arange(n_a - len(a), n_a)) and np.array_equal(b, 1_000_000 + np.arange(n_b - len(b), n_b))


def test_getitem_follows_numpy_indexing(dev):
    """replay_memory.py:67-70 indexes [maxlen, ...] arrays: slots beyond len are addressable, negatives count from
    maxlen, anything else raises IndexError, and an index array larger than len is fine (no OversampleError)."""
    from fastdeepqlearning_amd.Replay import ReplayMemory
    maxlen = 64
    r = ReplayMemory(maxlen, 4, 2, device=dev)
    for i in range(10):
        r.add({"obs": np.full(3, i, np.float32), "reward": float(i)})
    host = np.zeros((maxlen, 3), np.float32)
    host[:10] = np.arange(10, dtype=np.float32)[:, None]
    idx = np.array([[0, 9, 10, 63], [-1, -54, -64, 5]])
    got = r[idx]
    assert tuple(got["obs"].shape) == (2, 4, 3)
    np.testing.assert_array_equal(got["obs"].cpu().numpy(), host[idx])
    big = np.arange(-maxlen, maxlen).repeat(3)          # 384 indices from a ring of len 10
    np.testing.assert_array_equal(r[big]["obs"].cpu().numpy(), host[big])
    for bad in (64, -65, np.array([0, 1000])):
        with pytest.raises(IndexError):
            r[bad]
    with pytest.raises(IndexError):
        r[np.array([0.5])]
    # caller-supplied window starts of any sign are reduced mod len like the reference's `% _len`
    for i in range(10, 30):
        r.add({"obs": np.full(3, i, np.float32), "reward": float(i)})
    starts = torch.tensor([-1, 29, 30, 1 << 40], dtype=torch.int64)
    xp = r.temporal_sample(starts=starts)
    want = (np.arange(2)[:, None] + starts.numpy()[None, :]) % 30
    np.testing.assert_array_equal(xp["reward"].cpu().numpy()[..., 0], want.astype(np.float32))


def test_ring_orders_writes_and_reads_across_streams(dev):
    """Writes on one HIP stream, samples on another (two torch streams in two threads would do this): the handle
    makes the gather wait for the scatter and the next scatter for the gather."""
    from fastdeepqlearning_amd.core import NativeRing
    ring = NativeRing(4096, [4, 1], dev)
    s_w, s_r = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    n = 1024
    for rnd in range(20):
        rows = torch.full((n, 5), float(rnd), device=dev)
        with torch.cuda.stream(s_w):
            spin = torch.randn(2048, 2048, device=dev)
            for _ in range(4):
                spin = spin @ spin * 1e-3          # keeps the write stream busy ahead of the scatter
            ring.add_rows(rows + 0 * spin[0, 0])
        with torch.cuda.stream(s_r):
            idx = torch.arange(n, device=dev) + (rnd * n) % 4096
            outs = ring.gather_rows(idx % 4096)
            got = outs[0].clone()
        s_r.synchronize()
        assert bool((got == float(rnd)).all()), rnd
    torch.cuda.synchronize(dev)


def test_ring_orders_a_writer_that_moves_to_a_new_stream(dev):
    """Several rounds on ONE stream (no ordering events recorded yet), then the writer moves to a second stream while
    a slow gather is still queued on the first: the scatter must wait for that read (write-after-read across the
    single-stream -> multi-stream transition), and a reader on a third stream must still see it."""
    from fastdeepqlearning_amd.core import NativeRing
    n = 2048
    ring = NativeRing(n, [4, 1], dev)
    s_a, s_b, s_c = torch.cuda.Stream(dev), torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    idx = torch.arange(n, device=dev)
    with torch.cuda.stream(s_a):
        for rnd in range(5):
            ring.add_rows(torch.full((n, 5), float(rnd), device=dev))
            assert bool((ring.gather_rows(idx)[0] == float(rnd)).all())
        # a long queue on stream A, then the gather of the rows of round 4
        spin = torch.randn(4096, 4096, device=dev)
        for _ in range(6):
            spin = spin @ spin * 1e-4
        got_a = ring.gather_rows(idx)[0]
    with torch.cuda.stream(s_b):          # first time this handle sees a second stream: overwrite every slot
        ring.add_rows(torch.full((n, 5), 99.0, device=dev))
    with torch.cuda.stream(s_c):          # a second reader stream
        got_c = ring.gather_rows(idx)[0]
    with torch.cuda.stream(s_a):          # the writer comes back: must wait for BOTH readers
        ring.add_rows(torch.full((n, 5), 7.0, device=dev))
    torch.cuda.synchronize(dev)
    assert bool((got_a == 4.0).all()), "scatter on the new stream overtook the gather queued on the old one"
    assert bool((got_c == 99.0).all()), "reader on a third stream missed / was overwritten"
    assert bool((ring.gather_rows(idx)[0] == 7.0).all())
