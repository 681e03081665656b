"""franQ.Replay API over the HBM ring (reference: franQ/Replay/__init__.py:8-38)."""
from .replay_memory import ReplayMemory, AsyncReplayMemory, OversampleError
from . import wrappers


def make(conf, **kwargs):
    """Build one ring shard per env instance plus the write/read wrapper stacks selected by the
    conf flags; returns ``(read_heads, write_heads)`` exactly like the reference."""
    device = getattr(conf, "training_device", "cuda:0")
    shards = [AsyncReplayMemory(int(conf.replay_size), conf.batch_size, conf.temporal_len, device=device, seed=i)
              for i in range(conf.num_instances)]
    write_heads = read_heads = shards
    if conf.use_nStep_lowerbounds:
        if conf.her_mode == "vmap":
            write_heads = [wrappers.NStepReturnVmap(r, conf.nStep_return_steps, conf.gamma) for r in write_heads]
        else:
            write_heads = [wrappers.NStepReturn(r, conf.nStep_return_steps, conf.gamma) for r in write_heads]
    if conf.use_squashed_rewards and not conf.use_HER:
        write_heads = [wrappers.SquashRewards(r) for r in write_heads]
    if conf.use_HER:
        if conf.her_mode == "vmap":   # parity unpinned (reference needs jax); see wrappers/her_vmap.py
            write_heads = [wrappers.HindsightVmapWrite(r, kwargs["compute_reward"]) for r in write_heads]
            read_heads = [wrappers.HindsightVmapRead(r) for r in read_heads]
        else:
            write_heads = [wrappers.HindsightNStepReplay(r, kwargs["compute_reward"], mode=conf.her_mode)
                           for r in write_heads]
    return read_heads, write_heads
