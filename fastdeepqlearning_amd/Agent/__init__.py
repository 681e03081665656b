"""franQ.Agent API (reference: franQ/Agent/__init__.py:4-14)."""
from .conf import AgentConf, EncoderConf
from .deepQlearning import DeepQLearning


def make(agent_conf):
    if "obs_1d" in agent_conf.obs_space.spaces:
        assert len(agent_conf.obs_space.spaces["obs_1d"].shape) == 1
    algo = str(agent_conf.algorithm).lower()
    if algo == "deep_q_learning":
        return DeepQLearning(agent_conf)
    raise NotImplementedError(f"algorithm '{agent_conf.algorithm}' (the reference's RandomAgent cannot be built by "
                              f"Agent.make either: randomagent.py:9 takes a different signature)")
