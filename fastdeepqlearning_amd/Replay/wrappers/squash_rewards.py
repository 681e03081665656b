"""Pohlen reward squash at write time (reference: franQ/Replay/wrappers/squash_rewards.py:5-18)."""
import numpy as np

from .wrapper_base_class import ReplayMemoryWrapper


def _pohlen_transform(x, epsilon=1e-2, pow=0.5):
    return np.sign(x) * (np.power(np.abs(x) + 1, pow) - 1) + epsilon * x


class SquashRewards(ReplayMemoryWrapper):
    def add(self, experience_dict):
        experience_dict["reward"] = _pohlen_transform(experience_dict["reward"])
        ReplayMemoryWrapper.add(self, experience_dict)
