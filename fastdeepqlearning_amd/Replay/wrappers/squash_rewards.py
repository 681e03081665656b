"""SquashRewards: variance-reducing reward transform applied once, at write time.

Reference: franQ/Replay/wrappers/squash_rewards.py:5-18 (selected by ``use_squashed_rewards and not use_HER``,
franQ/Replay/__init__.py:28).  h(x) = sign(x) (sqrt(|x| + 1) - 1) + 0.01 x  (Pohlen et al., arXiv:1805.11593)."""
import numpy as np

from .wrapper_base_class import ReplayMemoryWrapper


class SquashRewards(ReplayMemoryWrapper):
    SLOPE = 1e-2      # the linear term keeps h invertible

    @classmethod
    def squash(cls, reward):
        r = np.asarray(reward, dtype=np.float64)
        magnitude = np.sqrt(np.abs(r) + 1.0) - 1.0
        out = np.copysign(magnitude, r) + cls.SLOPE * r
        return out if out.ndim else out.item()

    def add(self, experience_dict):
        record = dict(experience_dict)
        record["reward"] = self.squash(record["reward"])
        self.replay_buffer.add(record)
