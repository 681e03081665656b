"""Config container with attribute == item access (reference: franQ/common_utils.py:59-67)."""


class AttrDict(dict):
    __setattr__ = dict.__setitem__

    def __getattr__(self, item):
        try:
            return dict.__getitem__(self, item)
        except KeyError as e:
            raise AttributeError(e)
