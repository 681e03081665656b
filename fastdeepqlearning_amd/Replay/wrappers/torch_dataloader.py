"""TorchDataLoader: the trainer-side read head (reference:
franQ/Replay/wrappers/torch_dataloader.py:11-50).  Over the HBM ring there is nothing to
prefetch or copy — ``temporal_sample`` launches the gather kernel on the current stream and
returns float32 device tensors — so no thread or queue is started.  A foreign replay object
that yields numpy arrays is still converted per key like the reference does."""
import torch

from .wrapper_base_class import ReplayMemoryWrapper


class ConfigurationError(Exception):
    ...


class TorchDataLoader(ReplayMemoryWrapper):
    def __init__(self, replay_buffer, device="cuda:0", precision=torch.float32, use_temporal=True, vectorized=True):
        ReplayMemoryWrapper.__init__(self, replay_buffer)
        self.device, self.precision, self._use_temporal, self.vectorized = device, precision, use_temporal, vectorized
        if not use_temporal:
            raise NotImplementedError("TODO: Add support for pre-fetching and batching non-temporal samples")

    def ready(self):
        r = getattr(self.replay_buffer, "ready", None)
        return bool(r()) if r is not None else True

    def _convert(self, experience):
        return {k: (v if isinstance(v, torch.Tensor) and v.device == torch.device(self.device) and v.dtype == self.precision
                    else torch.as_tensor(v).to(device=self.device, dtype=self.precision))
                for k, v in experience.items()}

    def sample(self):
        if self._use_temporal:
            raise ConfigurationError("Incorrect Config! Unset `use_temporal` in init to support this feature")
        return self._convert(self.replay_buffer.sample())

    def temporal_sample(self, *args, **kwargs):
        if not self._use_temporal:
            raise ConfigurationError("Incorrect config! Set `use_temporal` in init to support this feature")
        return self._convert(self.replay_buffer.temporal_sample())
