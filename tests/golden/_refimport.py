"""Import helper used ONLY by tests/golden/make_golden.py (build container only).

It makes the read-only reference at /root/reference importable in this image by
putting permissive stand-in modules into ``sys.modules`` for third-party packages the
image lacks (gym, numba, jax, autoslot, tensorboard, torchvision, cv2, py_ics).  None of
the stand-ins touches hot-path arithmetic; ``numba.njit`` becomes the identity decorator,
so the reference's own 3-line Python loops run as plain Python.

Nothing in tests/, bench.py or the package imports this file at run time on the GPU box:
the reference cannot travel, only the vectors generated from it (tests/golden/*.npz).
"""
import sys
import types

REFERENCE_ROOT = "/root/reference"


class _AnyMeta(type):
    def __getattr__(cls, item):
        if item.startswith("__"):
            raise AttributeError(item)
        return _Any


class _Any(metaclass=_AnyMeta):
    """A class that can be subclassed, instantiated with anything, and poked at."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Any()

    def __getattr__(self, item):
        if item.startswith("__"):
            raise AttributeError(item)
        return _Any()


class _StubModule(types.ModuleType):
    def __getattr__(self, item):
        if item.startswith("__"):
            raise AttributeError(item)
        return _Any


def _stub(name):
    m = _StubModule(name)
    m.__path__ = []  # behave like a package so sub-imports resolve
    sys.modules[name] = m
    return m


def install():
    import numpy as np
    if not hasattr(np, "product"):
        np.product = np.prod  # removed in numpy 2; used at soft_actor_critic.py:42

    for name in ["gym", "gym.spaces", "gym.wrappers", "gym.envs", "gym.envs.classic_control",
                 "gym.utils", "gym.error",
                 "jax", "jax.numpy", "autoslot", "torchvision", "torchvision.transforms",
                 "cv2", "py_ics", "py_ics.gym_env", "py_ics.gym_env.envs",
                 "zarr", "caterva", "highway_env"]:
        if name not in sys.modules:
            _stub(name)
    # numba: njit must be a real identity decorator (with and without arguments)
    nb = _stub("numba")

    def njit(*args, **kwargs):
        if len(args) == 1 and callable(args[0]) and not kwargs:
            return args[0]
        return lambda f: f

    nb.njit = njit
    nb.jit = njit

    # tensorboard writer: swallow everything
    tb = _stub("torch.utils.tensorboard")

    class SummaryWriter:
        def __init__(self, *a, **k): pass
        def add_scalar(self, *a, **k): pass
        def add_scalars(self, *a, **k): pass
        def add_histogram(self, *a, **k): pass
        def flush(self): pass
        def close(self): pass

    tb.SummaryWriter = SummaryWriter
    import torch.utils
    torch.utils.tensorboard = tb

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


class Space:
    """Tiny object exposing what the reference reads from gym spaces
    (encoder.py:16-33, soft_actor_critic.py:14-22,42)."""

    def __init__(self, shape=None, n=None, spaces=None):
        if shape is not None:
            self.shape = tuple(shape)
        if n is not None:
            self.n = n
        if spaces is not None:
            self.spaces = spaces
