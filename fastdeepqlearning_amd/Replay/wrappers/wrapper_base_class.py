"""Delegating base of the replay wrappers (reference: franQ/Replay/wrappers/wrapper_base_class.py:17-39)."""


class ReplayMemoryWrapper:
    def __init__(self, replay_buffer):
        self.replay_buffer = replay_buffer

    def add(self, experience_dict):
        self.replay_buffer.add(experience_dict)

    def sample(self):
        return self.replay_buffer.sample()

    def temporal_sample(self):
        return self.replay_buffer.temporal_sample()

    def __getattr__(self, item):
        if "replay_buffer" in self.__dict__:
            return getattr(self.replay_buffer, item)
        raise AttributeError(item)

    def __len__(self):
        return len(self.replay_buffer)

    def __getitem__(self, item):
        return self.replay_buffer[item]
