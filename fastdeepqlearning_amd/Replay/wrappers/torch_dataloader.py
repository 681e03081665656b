"""TorchDataLoader: the trainer-side read head (reference: franQ/Replay/wrappers/torch_dataloader.py:11-50).

The reference starts a thread that keeps one sampled batch converted to float32 on the GPU in a queue.  Over the HBM
ring there is nothing to prefetch or copy: a sample IS a kernel launch on the current stream that returns float32
device tensors, so this class only keeps the reference's interface (``ready``, ``temporal_sample``, ``sample`` and the
``use_temporal`` switch that selects which of the two is legal) and converts per key when it wraps a foreign replay
object that still yields numpy arrays."""
import torch

from .wrapper_base_class import ReplayMemoryWrapper


class ConfigurationError(Exception):
    """Asked for the kind of sample the loader was not configured for (same rule as the reference)."""


class TorchDataLoader(ReplayMemoryWrapper):
    def __init__(self, replay_buffer, device="cuda:0", precision=torch.float32, use_temporal=True, vectorized=True):
        ReplayMemoryWrapper.__init__(self, replay_buffer)
        self.device = torch.device(device)
        self.precision = precision
        self.vectorized = vectorized
        self._use_temporal = bool(use_temporal)    # False: plain [B, *] minibatches (the reference never built this mode)

    def ready(self):
        probe = getattr(self.replay_buffer, "ready", None)
        return True if probe is None else bool(probe())

    def _on_device(self, batch):
        out = {}
        for key, value in batch.items():
            if isinstance(value, torch.Tensor) and value.device == self.device and value.dtype == self.precision:
                out[key] = value
            else:
                out[key] = torch.as_tensor(value).to(device=self.device, dtype=self.precision)
        return out

    def _require(self, temporal):
        if self._use_temporal != temporal:
            want = "use_temporal=True" if temporal else "use_temporal=False"
            raise ConfigurationError(f"this loader was built for the other sample kind; construct it with {want}")

    def sample(self):
        self._require(temporal=False)
        return self._on_device(self.replay_buffer.sample())

    def temporal_sample(self, *args, **kwargs):
        self._require(temporal=True)
        return self._on_device(self.replay_buffer.temporal_sample())
