"""ctypes binding of libfdql_hip.so (C ABI: include/fdql.h).

The product path has NO CPU fallback: if the HIP library is missing or a call fails this
module raises.  torch is imported first so that the library's libamdhip64.so.7 dependency
resolves to the HIP runtime torch already loaded (one runtime per process: device pointers
and streams are shared with torch tensors).
"""
import ctypes as C
import os

import torch  # noqa: F401  (must precede loading the HIP library; see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
# FDQL_LIB_PATH: load another build of the same library (A/B runs of kernel variants on one GPU box)
LIB_PATH = os.environ.get("FDQL_LIB_PATH") or os.path.join(_HERE, "libfdql_hip.so")

FDQL_OK, FDQL_EINVAL, FDQL_EHIP, FDQL_EOVERSAMPLE, FDQL_ESTATE, FDQL_ENOMEM = 0, -1, -2, -3, -4, -5
PHASE_ALL, PHASE_GRAD, PHASE_APPLY, PHASE_GRAD_CRITICS, PHASE_GRAD_REST = 0, 1, 2, 3, 4
MAX_HIDDEN = 4
MAX_CONV = 3


class NativeLibraryMissing(ImportError):
    pass


class FdqlError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"fdql error {code}: {msg}")
        self.code = code


class OversampleError(Exception):
    """franQ/Replay/replay_memory.py:6"""


class AgentConfig(C.Structure):
    _fields_ = [
        ("obs_dim", C.c_int32), ("goal_dim", C.c_int32), ("act_dim", C.c_int32), ("discrete", C.c_int32),
        ("n_critics", C.c_int32), ("n_quantiles", C.c_int32), ("latent", C.c_int32), ("enc_features", C.c_int32),
        ("n_enc_hidden", C.c_int32), ("enc_hidden", C.c_int32 * MAX_HIDDEN),
        ("n_joint_hidden", C.c_int32), ("joint_hidden", C.c_int32 * MAX_HIDDEN),
        ("n_pi_hidden", C.c_int32), ("pi_hidden", C.c_int32 * MAX_HIDDEN),
        ("n_critic_hidden", C.c_int32), ("critic_hidden", C.c_int32 * MAX_HIDDEN),
        ("img_c", C.c_int32), ("img_h", C.c_int32), ("img_w", C.c_int32), ("n_conv", C.c_int32),
        ("conv_out", C.c_int32 * MAX_CONV), ("conv_k", C.c_int32 * MAX_CONV), ("conv_s", C.c_int32 * MAX_CONV),
        ("joiner_gru", C.c_int32), ("gru_state_mode", C.c_int32),
        ("distributional", C.c_int32), ("use_lowerbound", C.c_int32), ("use_max_entropy", C.c_int32),
        ("hard_updates", C.c_int32), ("keep_frozen_copy", C.c_int32), ("bootstrap_nstep", C.c_int32), ("obs_2d_u8", C.c_int32),
        ("burn_in_steps", C.c_int32),
        ("T", C.c_int32), ("B", C.c_int32), ("world_size", C.c_int32),
        ("gamma", C.c_double), ("tau", C.c_double), ("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double),
        ("adam_eps", C.c_double), ("init_log_alpha", C.c_double), ("drop_frac", C.c_double),
    ]


class Batch(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("obs_1d", "achieved_goal", "desired_goal", "action", "reward", "mc_return",
                                          "task_done", "episode_step", "obs_2d", "obs_2d_u8", "obs_2d_slots", "agent_state")]


class AgentStats(C.Structure):
    _fields_ = [("gemm_flops", C.c_double), ("skinny_flops", C.c_double), ("n_launches", C.c_int32),
                ("n_gemm_launches", C.c_int32), ("params", C.c_int64), ("plans_built", C.c_int64),
                ("graph_launches", C.c_int64)]


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("ms", C.c_float), ("flops", C.c_double), ("bytes", C.c_double)]


class RewardFn(C.Structure):
    _fields_ = [("kind", C.c_int32), ("threshold", C.c_float), ("miss_reward", C.c_float)]


class EpisodeSpec(C.Structure):
    _fields_ = [("reward_key", C.c_int32), ("return_key", C.c_int32), ("emit_pop", C.c_int32), ("n_step", C.c_int32),
                ("gamma", C.c_float), ("her", C.c_int32), ("achieved_key", C.c_int32), ("desired_key", C.c_int32),
                ("task_done_key", C.c_int32), ("step_key", C.c_int32), ("goal_row", C.c_int32),
                ("reward_fn", RewardFn)]


class EpisodeVmapSpec(C.Structure):
    _fields_ = [("reward_key", C.c_int32), ("task_done_key", C.c_int32), ("achieved_key", C.c_int32), ("desired_key", C.c_int32),
                ("vgoals_key", C.c_int32), ("vrewards_key", C.c_int32), ("vdones_key", C.c_int32), ("vreturn_key", C.c_int32),
                ("K", C.c_int32), ("n_step", C.c_int32), ("gamma", C.c_float), ("reward_fn", RewardFn)]


# Every symbol include/fdql.h declares, with its ctypes signature (restype, argtypes).
_vp, _i32, _i64, _u64, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_float
SIGNATURES = {
    "fdql_last_error": (C.c_char_p, []),
    "fdql_version": (C.c_int, []),
    "fdql_abi_sizes": (None, [C.POINTER(C.c_int32)]),
    "fdql_ring_create": (C.c_int, [C.POINTER(_vp), _i64, _i32, C.POINTER(_i32)]),
    "fdql_ring_create_typed": (C.c_int, [C.POINTER(_vp), _i64, _i32, C.POINTER(_i32), C.POINTER(_i32)]),
    "fdql_ring_destroy": (C.c_int, [_vp]),
    "fdql_ring_add": (C.c_int, [_vp, _vp, _i64, _vp]),
    "fdql_ring_add_device": (C.c_int, [_vp, _vp, _i64, _vp]),
    "fdql_ring_flush": (C.c_int, [_vp, _vp]),
    "fdql_ring_append_episode": (C.c_int, [_vp, _vp, _i64, C.POINTER(EpisodeSpec), C.POINTER(_i64), _vp]),
    "fdql_ring_append_episode_vmap": (C.c_int, [_vp, _vp, _i64, _vp, C.POINTER(EpisodeVmapSpec), C.POINTER(_i64), _vp]),
    "fdql_ring_snapshot": (C.c_int, [_vp, _vp, _i64, _vp]),
    "fdql_ring_restore": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _vp]),
    "fdql_ring_len": (_i64, [_vp]),
    "fdql_ring_top": (_i64, [_vp]),
    "fdql_ring_row_floats": (_i64, [_vp]),
    "fdql_ring_key_ptr": (C.c_int, [_vp, _i32, C.POINTER(_vp)]),
    "fdql_ring_key_ptr_u8": (C.c_int, [_vp, _i32, C.POINTER(_vp)]),
    "fdql_ring_window_slots": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _vp]),
    "fdql_ring_external_read": (C.c_int, [_vp, _i32, _vp]),
    "fdql_ring_sample_windows": (C.c_int, [_vp, _i32, _i32, _vp, _u64, _u64, C.POINTER(_vp), _vp, _vp]),
    "fdql_ring_sample_windows_sel": (C.c_int, [_vp, _i32, _i32, _vp, _u64, _u64, C.POINTER(_vp), C.POINTER(_i32),
                                               C.POINTER(_i32), _vp, _vp]),
    "fdql_ring_sample_rows": (C.c_int, [_vp, _i32, _vp, _u64, _u64, C.POINTER(_vp), _vp, _vp]),
    "fdql_debug_chain_stamps": (C.c_int, [_vp, _i32]),
    "fdql_test_chain_mlp": (C.c_int, [_vp, _i32, _i32, C.POINTER(_i32), _i32, _i32, _vp, C.POINTER(_vp), _vp, _vp]),
    "fdql_ring_gather_rows": (C.c_int, [_vp, _i64, _vp, C.POINTER(_vp), _vp]),
    "fdql_episode_mc_return": (C.c_int, [_vp, _vp, _i32, _f32, _vp]),
    "fdql_episode_her_vmap": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, C.POINTER(RewardFn), _vp, _vp, _vp, _vp]),
    "fdql_episode_mc_return_vmap": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _f32, _vp]),
    "fdql_episode_her_relabel": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, C.POINTER(RewardFn), _vp, _vp, _vp, _vp]),
    "fdql_agent_create": (C.c_int, [C.POINTER(_vp), C.POINTER(AgentConfig)]),
    "fdql_agent_destroy": (C.c_int, [_vp]),
    "fdql_agent_arena_floats": (_i64, [_vp, _i32]),
    "fdql_agent_tensor_info": (_i32, [_vp, _i32, C.c_char_p, _i32, C.POINTER(_i32), C.POINTER(_i64), C.POINTER(_i32)]),
    "fdql_agent_workspace_bytes": (_i64, [_vp]),
    "fdql_agent_conv_reads_ring": (_i32, [_vp]),
    "fdql_agent_bind": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64]),
    "fdql_agent_update": (C.c_int, [_vp, C.POINTER(Batch), _vp, _vp, _u64, _i32, _vp]),
    "fdql_agent_grad_bucket": (C.c_int, [_vp, C.POINTER(_i64)]),
    "fdql_agent_set_launch_mode": (C.c_int, [_vp, _i32]),
    "fdql_agent_scalars": (C.c_int, [_vp, C.POINTER(_f32), _vp]),
    "fdql_agent_summaries": (C.c_int, [_vp, C.POINTER(_f32), _i32, _i32, _vp]),
    "fdql_agent_set_alpha": (C.c_int, [_vp, _f32, _vp]),
    "fdql_agent_set_step": (C.c_int, [_vp, _i32, _vp]),
    "fdql_agent_debug_ptr": (C.c_int, [_vp, C.c_char_p, C.POINTER(_vp), C.POINTER(_i64)]),
    "fdql_agent_act_workspace_bytes": (_i64, [_vp, _i32]),
    "fdql_agent_act": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u64, _u64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i64,
                                 _vp]),
    "fdql_agent_stats": (C.c_int, [_vp, C.POINTER(AgentStats)]),
    "fdql_agent_profile_update": (_i32, [_vp, C.POINTER(Batch), _vp, _vp, _u64, C.POINTER(KernelTime), _i32, _vp]),
    "fdql_debug_set_gemm_dense_shape": (C.c_int, [_i32]),
    "fdql_test_gemm": (C.c_int, [_vp, _i32, _i32, _vp, _i32, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _i32,
                                 _i32, _vp]),
    "fdql_debug_side_copy": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp]),
    "fdql_test_conv": (C.c_int, [_i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "fdql_test_wgrad_stat": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i64, _vp]),
    "fdql_test_wgrad_stat_riders": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i64, _vp, _i32, _i32, _vp, _i32, _i32,
                                              _vp, _i32, _i32, _vp, _i32, _vp]),
    "fdql_test_rowgemm": (C.c_int, [_vp, _vp, _i32, _vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32,
                                    _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _vp]),
}

_lib = None


def load():
    """Load libfdql_hip.so; raises NativeLibraryMissing (never falls back to CPU code)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryMissing(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C fastdeepqlearning_amd/csrc`. There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    sizes = (C.c_int32 * 6)()
    lib.fdql_abi_sizes(sizes)
    mine = [C.sizeof(x) for x in (AgentConfig, Batch, AgentStats, KernelTime, RewardFn, EpisodeSpec)]
    if list(sizes) != mine:
        raise ImportError(f"struct layout mismatch between _native.py {mine} and libfdql_hip.so {list(sizes)}")
    _lib = lib
    return lib


def check(rc):
    if rc == FDQL_OK:
        return
    msg = load().fdql_last_error().decode("utf-8", "replace")
    if rc == FDQL_EOVERSAMPLE:
        raise OversampleError(msg)
    raise FdqlError(rc, msg)


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    if t is None:
        return None
    assert t.is_contiguous() and t.dtype == torch.float32, "fdql expects contiguous float32 tensors"
    return C.c_void_p(t.data_ptr())


def current_stream(device=None):
    if torch.cuda.is_available():
        return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    return C.c_void_p(0)
