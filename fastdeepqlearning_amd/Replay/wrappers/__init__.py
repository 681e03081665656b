from . import squash_rewards, wrapper_base_class, torch_dataloader, nstep_return, her, episode_ops, her_vmap, nstep_return_vmap
from .nstep_return import NStepReturn
from .her import HindsightNStepReplay
from .torch_dataloader import TorchDataLoader
from .squash_rewards import SquashRewards
from .episode_ops import SparseL2Reward
from .her_vmap import HindsightVmapWrite, HindsightVmapRead
from .nstep_return_vmap import NStepReturnVmap
