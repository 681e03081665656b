"""Agent hyper-parameters under the reference's field names and defaults (franQ/Agent/conf.py:8-98), so that a conf
object built for franQ drives this package unchanged.  Attribute and item access are interchangeable
(franQ/common_utils.py:59-67)."""
from dataclasses import dataclass
from enum import Enum
from pathlib import Path

from ..common_utils import AttrDict


@dataclass
class EncoderConf:
    hidden_features = 256
    joint_hidden_dims: tuple = 256,
    obs_1d_hidden_dims: tuple = 256,

    class JoinerModeEnum(Enum):
        feedforward = 1
        gru = 2

    joiner_mode = JoinerModeEnum.feedforward

    class RnnLatentStateTrainMode(Enum):
        zero = 0
        store = 1
        learned = 2

    rnn_latent_state_training_mode = RnnLatentStateTrainMode.zero
    use_burn_in = False
    burn_in_portion = 0.2
    # addition of this implementation: the pixel encoder the reference never had (obs_2d in the obs space)
    conv_layers = ((32, 8, 4), (64, 4, 2), (64, 3, 1))     # (out_channels, kernel, stride) per layer


# field -> default, grouped the way the update path consumes them
_REPLAY = dict(batch_size=256, replay_size=int(5e4), temporal_len=50, use_squashed_rewards=False,
               use_nStep_lowerbounds=True, nStep_return_steps=1000, use_HER=False, her_mode="final")
_ALGORITHM = dict(use_distributional_sac=True, use_max_entropy_q=True, use_hard_updates=False, hard_update_interval=200,
                  use_bootstrap_minibatch_nstep=False, init_log_alpha=-2, gamma=0.99, learning_rate=3e-4, tau=5e-2,
                  clip_grad_norm=5e-3, top_quantiles_to_drop=0.2)
_NETWORKS = dict(num_critics=2, num_q_predictions=10, latent_state_dim=256, use_decoder=False,
                 use_hsv_data_augmentation=False)
_RUNTIME = dict(algorithm="deep_q_learning", log_extra_debug_info=False, enable_timers=False, log_interval=50,
                param_update_interval=50, use_async_train=True,
                inference_input_keys=("obs_1d", "obs_2d", "idx", "achieved_goal", "desired_goal", "agent_state"))
# additions of this implementation: ring shards per process (the reference's launcher sets it) and data-parallel ranks
_ADDED = dict(num_instances=1, world_size=1, force_distributed_step=False)


class AgentConf(AttrDict):
    def __init__(self):
        AttrDict.__init__(self)
        import torch
        from torch import multiprocessing as mp
        for group in (_RUNTIME, _REPLAY, _ALGORITHM, _NETWORKS, _ADDED):
            for name, default in group.items():
                self[name] = default
        # live objects and mutable defaults are made per instance
        self.obs_space = self.action_space = self.discrete = None     # filled in by the env / launcher
        self.train_step = mp.Value("i", 0)
        self.training_device = self.inference_device = torch.device("cuda:0" if torch.cuda.is_available() else "cpu:0")
        self.dtype = torch.float32
        self.eval_envs = [0]
        self.log_dir = Path("logs")
        self.encoder_conf = EncoderConf()
        self.pi_hidden_dims = [256]
        self.critic_hidden_dims = [256, 256]
