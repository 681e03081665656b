"""TorchDataLoader: the trainer-side read head (reference: franQ/Replay/wrappers/torch_dataloader.py:11-50).

The reference starts a thread that keeps one sampled batch converted to float32 on the GPU in a queue
(torch_dataloader.py:22, 40-50).  Over the HBM ring a sample IS a kernel launch that returns float32 device tensors, so
there is nothing to copy or convert.  The reference's one-batch-ahead overlap is available (``prefetch=True``: over an
HBM ring shard with a rotating pool of persistent sample buffers - what ``Replay.make`` builds - the NEXT batch's gather is
issued on a side stream while the trainer's update of the current one runs, and ``temporal_sample()`` hands out the batch
drawn one call earlier, the semantics of the reference's ``Queue(maxsize=1)``) but OFF by default: measured on MI355X
(round 4, config 2) the step is 0.8-1.5 % slower with it - the update's dense kernels are persistent one-workgroup-per-CU
launches and the gather's workgroups delay them by more than the 12 us of gather they hide.  By default the class only keeps the reference's interface (``ready``, ``temporal_sample``, ``sample`` and the ``use_temporal`` switch that
selects which of the two is legal) and converts per key when it wraps a foreign replay object that still yields numpy
arrays."""
import os

import torch

from ..._native import OversampleError

from .wrapper_base_class import ReplayMemoryWrapper


class ConfigurationError(Exception):
    """Asked for the kind of sample the loader was not configured for (same rule as the reference)."""


class TorchDataLoader(ReplayMemoryWrapper):
    def __init__(self, replay_buffer, device="cuda:0", precision=torch.float32, use_temporal=True, vectorized=True, prefetch=False):
        ReplayMemoryWrapper.__init__(self, replay_buffer)
        self.device = torch.device(device)
        self.precision = precision
        self.vectorized = vectorized
        self._use_temporal = bool(use_temporal)    # False: plain [B, *] minibatches (the reference never built this mode)
        # one batch ahead on a side stream: only over a ring shard whose samples land in >= 3 persistent buffers (the batch
        # in use, the one being prefetched and the previous one, which the update before may still be reading)
        self._prefetch = bool(prefetch) and os.environ.get("FDQL_NO_PREFETCH") is None and self.device.type == "cuda"
        self._pending, self._side, self._events, self._ev_i = None, None, None, 0
        self.keep_uint8 = set()   # keys the consumer wants as uint8 device tensors instead of `precision` (set by the agent)

    def _can_prefetch(self):
        from ..replay_memory import ReplayMemory
        rb = self.replay_buffer
        # a shard whose frames are read IN PLACE is never prefetched: such a batch is not a snapshot (the update reads the
        # ring's own slots when it runs), so a batch drawn one step ahead could see frames a writer replaced in between
        return self._prefetch and isinstance(rb, ReplayMemory) and rb._pool_n >= 3 and rb.device == self.device and not rb._in_place

    def _issue_prefetch(self):
        """The next batch's gather on the side stream, behind everything the current stream holds so far (the update that
        last read the buffer being refilled) and beside whatever is enqueued after this call (the update of the batch
        returned now).  Writers that arrive on another stream afterwards (``add`` / ``add_rows`` scatters on the main
        stream) are ordered behind this gather by the ring itself: every launch that writes slots waits for the events of
        the reads issued on other streams (csrc/ring.hip ``begin_write`` / ``end_read``), so the gather never sees
        half-written rows."""
        if self._side is None:
            self._side = torch.cuda.Stream(self.device)
            self._events = [torch.cuda.Event() for _ in range(6)]
        main = torch.cuda.current_stream(self.device)
        e_main, e_done = self._events[self._ev_i], self._events[self._ev_i + 1]
        self._ev_i = (self._ev_i + 2) % 6
        e_main.record(main)
        self._side.wait_event(e_main)
        try:
            with torch.cuda.stream(self._side):
                batch = self.replay_buffer.temporal_sample()
                e_done.record(self._side)
            self._pending = (batch, e_done)
        except OversampleError:        # (the shard shrank below a window: the next call samples in place and reports it)
            self._pending = None

    def ready(self):
        probe = getattr(self.replay_buffer, "ready", None)
        return True if probe is None else bool(probe())

    def _on_device(self, batch):
        out = {}
        for key, value in batch.items():
            if isinstance(value, torch.Tensor) and value.device == self.device and (
                    (key.endswith("_slots") and value.dtype == torch.int32) or (value.dtype == torch.uint8 and key + "_slots" in batch)):
                out[key] = value          # a key read in place (ReplayMemory.enable_in_place): the ring's uint8 block + row slots
            elif key in self.keep_uint8:  # the agent takes these frames as bytes (fdql_batch_t.obs_2d_u8), whatever the shard stored
                out[key] = torch.as_tensor(value).to(device=self.device, dtype=torch.uint8).contiguous()
            elif isinstance(value, torch.Tensor) and value.device == self.device and value.dtype == self.precision:
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
        if not self._can_prefetch():
            return self._on_device(self.replay_buffer.temporal_sample())
        if self._pending is None:
            batch = self.replay_buffer.temporal_sample()           # first call: drawn in place on the current stream
        else:
            batch, done = self._pending
            self._pending = None
            torch.cuda.current_stream(self.device).wait_event(done)
        self._issue_prefetch()
        return self._on_device(batch)      # (a pass-through for the ring's float32 device tensors; applies keep_uint8)
