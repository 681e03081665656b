"""Agent hyper-parameters with the reference's field names and defaults
(reference: franQ/Agent/conf.py:8-98)."""
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


class AgentConf(AttrDict):
    def __init__(self):
        AttrDict.__init__(self)
        import torch
        from torch import multiprocessing as mp
        self.algorithm = "deep_q_learning"
        self.obs_space = None
        self.action_space = None
        self.discrete = None
        self.train_step = mp.Value("i", 0)
        self.inference_input_keys = "obs_1d", "obs_2d", "idx", "achieved_goal", "desired_goal", "agent_state"
        dev = torch.device("cuda:0" if torch.cuda.is_available() else "cpu:0")
        self.training_device = dev
        self.inference_device = dev
        self.dtype = torch.float32
        self.eval_envs = [0]
        self.log_dir = Path("logs")
        self.log_extra_debug_info = False
        self.enable_timers = False
        self.log_interval = 50
        self.param_update_interval = 50
        self.batch_size = 256
        self.replay_size = int(5e4)
        self.temporal_len = 50
        self.clip_grad_norm = 5e-3
        self.use_squashed_rewards = False
        self.use_hard_updates = False
        self.use_nStep_lowerbounds = True
        self.nStep_return_steps = 1000
        self.use_max_entropy_q = True
        self.use_HER = False
        self.her_mode = "final"
        self.use_distributional_sac = True
        self.init_log_alpha = -2
        self.gamma = 0.99
        self.learning_rate = 3e-4
        self.tau = 5e-2
        self.hard_update_interval = 200
        self.encoder_conf = EncoderConf()
        self.pi_hidden_dims = [256]
        self.critic_hidden_dims = [256, 256]
        self.num_critics = 2
        self.num_q_predictions = 10
        self.latent_state_dim = 256
        self.top_quantiles_to_drop = 0.2
        self.use_bootstrap_minibatch_nstep = False
        self.use_async_train = True
        self.use_decoder = False
        self.use_hsv_data_augmentation = False
        # additions of this implementation
        self.num_instances = 1
        self.world_size = 1
