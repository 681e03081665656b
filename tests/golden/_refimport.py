"""Import helper used ONLY by tests/golden/make_golden.py (build container only).

It makes the read-only reference at /root/reference importable in this image by
putting permissive stand-in modules into ``sys.modules`` for third-party packages the
image lacks (gym, numba, jax, autoslot, tensorboard, torchvision, cv2, py_ics).  Two of the
stand-ins sit under hot-path arithmetic and are therefore written out rather than stubbed:
``numba.njit`` becomes the identity decorator, so the reference's own 3-line Python loops
run as plain Python (pure float32 under numpy 2; real numba forms the product in double),
and ``jax`` / ``jax.numpy`` become a numpy-backed stand-in (`_install_jax`) with exactly
what franQ/Replay/wrappers/her_vmap.py uses: ``vmap`` (loop over the mapped axis + stack,
honouring ``in_axes``), ``logical_and/or/not``, ``devices`` and ``device_put`` (which, like
jax with x64 disabled, demotes float64 -> float32 and int64 -> int32).  Fixtures produced
through it (tests/golden/her_vmap.npz) are "shim-pinned": the reference's own file ran,
on a stand-in for a library the image lacks.

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
                 "autoslot", "torchvision", "torchvision.transforms",
                 "cv2", "py_ics", "py_ics.gym_env", "py_ics.gym_env.envs",
                 "zarr", "caterva", "highway_env"]:
        if name not in sys.modules:
            _stub(name)
    _install_jax()
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


def _install_jax():
    """numpy-backed jax / jax.numpy with the handful of entry points her_vmap.py:26-45,66-78 calls."""
    import numpy as np

    def demote(x):     # jax with x64 disabled (the default): 64-bit inputs become 32-bit on device_put / asarray
        a = np.asarray(x)
        if a.dtype == np.float64:
            return a.astype(np.float32)
        if a.dtype == np.int64:
            return a.astype(np.int32)
        return a

    def vmap(fn, in_axes=0, out_axes=0):
        assert out_axes == 0

        def mapped(*args):
            axes = tuple(in_axes) if isinstance(in_axes, (tuple, list)) else (in_axes,) * len(args)
            assert len(axes) == len(args)
            sizes = {np.shape(a)[ax] for a, ax in zip(args, axes) if ax is not None}
            assert len(sizes) == 1, f"vmap: mapped axes disagree: {sizes}"
            outs = []
            for i in range(sizes.pop()):
                call = [a if ax is None else np.take(demote(a), i, axis=ax) for a, ax in zip(args, axes)]
                outs.append(fn(*call))
            if isinstance(outs[0], (tuple, list)):
                return tuple(np.stack([np.asarray(o[j]) for o in outs]) for j in range(len(outs[0])))
            return np.stack([np.asarray(o) for o in outs])

        return mapped

    jax = types.ModuleType("jax")
    jnp = types.ModuleType("jax.numpy")
    jax.__path__ = []
    jax.vmap = vmap
    jax.devices = lambda kind=None: ["cpu:0"]
    jax.device_put = lambda x, device=None: demote(x)
    for name in ("logical_and", "logical_or", "logical_not", "sqrt", "sum", "abs", "square", "where", "linalg",
                 "float32", "asarray", "array", "all", "any", "equal", "less", "greater", "mean"):
        setattr(jnp, name, getattr(np, name))
    jax.numpy = jnp
    sys.modules["jax"] = jax
    sys.modules["jax.numpy"] = jnp


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
