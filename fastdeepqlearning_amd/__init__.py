"""fastdeepqlearning_amd — MI355X-native SAC/TQC update path with the franQ Replay/Agent API.

    from fastdeepqlearning_amd import Replay, Agent      # franQ.Replay / franQ.Agent shapes
    from fastdeepqlearning_amd.core import NativeRing, NativeAgent   # thin C-ABI wrappers

All compute on the hot path runs in libfdql_hip.so (hand-written HIP for gfx950, C ABI in
include/fdql.h).  There is no CPU fallback: importing works anywhere, but constructing a
ring or an agent without the library (or without a GPU) raises.
"""
__version__ = "0.1.0"
