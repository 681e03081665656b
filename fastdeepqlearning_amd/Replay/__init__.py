"""franQ.Replay API over the HBM ring (reference: franQ/Replay/__init__.py:8-38)."""
from .replay_memory import ReplayMemory, AsyncReplayMemory, OversampleError
from . import wrappers


def _write_stack(conf, shard, compute_reward):
    """Write-side wrappers of one shard, innermost first, chosen by the same conf flags as the reference:
    n-step return (vmap variant under her_mode == "vmap"), reward squash (never together with HER),
    hindsight relabel outermost."""
    vmap = conf.use_HER and conf.her_mode == "vmap"
    head = shard
    if conf.use_nStep_lowerbounds:
        cls = wrappers.NStepReturnVmap if conf.her_mode == "vmap" else wrappers.NStepReturn
        head = cls(head, conf.nStep_return_steps, conf.gamma)
    if conf.use_squashed_rewards and not conf.use_HER:
        head = wrappers.SquashRewards(head)
    if conf.use_HER:
        if vmap:        # shim-pinned (the reference needs jax); see wrappers/her_vmap.py
            head = wrappers.HindsightVmapWrite(head, compute_reward)
        else:
            head = wrappers.HindsightNStepReplay(head, compute_reward, mode=conf.her_mode)
    return head


def make(conf, **kwargs):
    """One ring shard per env instance; returns ``(read_heads, write_heads)`` like the reference.  The read
    heads are the shards themselves (the gather kernel already returns float32 device tensors), wrapped only
    for the read-time goal selection of HER-vmap."""
    device = getattr(conf, "training_device", "cuda:0")
    # (sample_buffers=3: the trainer's read heads recycle three batch buffers, so the agent's cached launch plans recur)
    shards = [AsyncReplayMemory(int(conf.replay_size), conf.batch_size, conf.temporal_len, device=device, seed=i, sample_buffers=3)
              for i in range(conf.num_instances)]
    reward_fn = kwargs.get("compute_reward") if conf.use_HER else None
    if conf.use_HER and reward_fn is None:
        raise KeyError("compute_reward")        # the reference indexes kwargs["compute_reward"]
    write_heads = [_write_stack(conf, s, reward_fn) for s in shards]
    read_heads = [wrappers.HindsightVmapRead(s) for s in shards] if conf.use_HER and conf.her_mode == "vmap" else shards
    return read_heads, write_heads
