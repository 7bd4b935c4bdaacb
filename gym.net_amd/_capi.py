"""ctypes binding of include/gymnet_amd.h — the same entry points a C# host would P/Invoke.

The shared library is gym.net_amd/lib/libgymnet_amd.so (HIP kernels + C ABI, built by
gym.net_amd/build.py).  There is no fallback: if the library is missing this module raises, and
without a GPU every compute call returns GYMNET_ERR_NO_DEVICE which is raised as NoDeviceError.
"""
import ctypes as C
import importlib.util
import os
import sys

from .errors import (AlreadySteppingError, GymNetError, InvalidActionError, NoDeviceError,
                     NotSteppingError)

HERE = os.path.dirname(os.path.abspath(__file__))
# GYMNET_LIB_PATH points the binding at another build of the same library (A/B timing of two builds, packaged installs)
LIB_PATH = os.environ.get("GYMNET_LIB_PATH") or os.path.join(HERE, "lib", "libgymnet_amd.so")

OK = 0
ERR_INVALID_ARG = -1
ERR_INVALID_ACTION = -2
ERR_HIP = -3
ERR_OOM = -4
ERR_NO_DEVICE = -5
ERR_ALREADY_STEPPING = -6
ERR_NOT_STEPPING = -7
ERR_UNSUPPORTED = -8
ERR_RCCL = -9

ENV_CARTPOLE, ENV_PENDULUM, ENV_MOUNTAINCAR, ENV_ACROBOT = 0, 1, 2, 3
ENV_IDS = {"CartPole-v1": 0, "Pendulum-v1": 1, "MountainCar-v0": 2, "Acrobot-v1": 3}

FLAG_AUTORESET = 0x01
FLAG_VALIDATE_ACTIONS = 0x02
FLAG_DONE_LIST = 0x04
FLAG_EPISODE_STATS = 0x08
FLAG_FINAL_OBS = 0x10
FLAG_DOUBLE_BUFFER = 0x20
FLAG_F64 = 0x40
FLAG_COMPACT_RECORDS_ONLY = 0x80
FLAG_RESIDENT = 0x100

DTYPE_F32, DTYPE_F64 = 0, 1
(ARRAY_REWARD, ARRAY_DONE, ARRAY_STEPS_BEYOND_DONE, ARRAY_EPISODE_RETURN, ARRAY_EPISODE_LENGTH, ARRAY_FINISHED_RETURN,
 ARRAY_FINISHED_LENGTH, ARRAY_FINAL_OBS, ARRAY_LANE_SEEDS) = range(9)

GATHER_NONE, GATHER_DIRECT, GATHER_RCCL = 0, 1, 2
ABI_VERSION = 6
RECORDS_NO_OVERFLOW = 1      # gymnet_rollout_spec.record_flags


class Config(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("env_id", C.c_int32), ("num_envs", C.c_int64),
                ("lane_offset", C.c_int64), ("device", C.c_int32), ("flags", C.c_uint32),
                ("seed", C.c_uint64), ("stream", C.c_void_p), ("d_ext_obs", C.c_void_p),
                ("ext_obs_stride", C.c_int64), ("max_episode_steps", C.c_int32), ("reserved", C.c_int32),
                ("d_ext_obs_alt", C.c_void_p)]


class EnvInfo(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("env_id", C.c_int32), ("name", C.c_char * 32),
                ("state_dim", C.c_int32), ("obs_dim", C.c_int32), ("obs_aliases_state", C.c_int32),
                ("action_is_box", C.c_int32), ("action_n", C.c_int32),
                ("action_low", C.c_float), ("action_high", C.c_float),
                ("obs_low", C.c_float * 8), ("obs_high", C.c_float * 8),
                ("reward_low", C.c_float), ("reward_high", C.c_float),
                ("algorithmic_bytes_per_step", C.c_int32), ("traffic_bytes_per_step", C.c_int32),
                ("state_row_in_obs", C.c_int32 * 8)]


class DeviceView(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("state_dim", C.c_int32), ("obs_dim", C.c_int32),
                ("obs_aliases_state", C.c_int32), ("num_envs", C.c_int64), ("state_stride", C.c_int64),
                ("obs_stride", C.c_int64), ("d_state", C.c_void_p), ("d_obs", C.c_void_p),
                ("d_reward", C.c_void_p), ("d_done", C.c_void_p), ("d_steps_beyond_done", C.c_void_p),
                ("d_final_obs", C.c_void_p), ("d_done_list", C.c_void_p),
                ("d_episode_return", C.c_void_p), ("d_episode_length", C.c_void_p),
                ("d_finished_return", C.c_void_p), ("d_finished_length", C.c_void_p),
                ("stream", C.c_void_p), ("obs_buffer", C.c_int32), ("state_dtype", C.c_int32),
                ("d_obs_alt", C.c_void_p)]


class LaunchPolicy(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("vec", C.c_int32), ("block", C.c_int32), ("nt", C.c_int32),
                ("sequential_lanes", C.c_int32), ("reset_form", C.c_int32), ("lds_pipe", C.c_int32),
                ("occupancy_lds_bytes", C.c_int32), ("graph", C.c_int32)]


class GroupConfig(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("env_id", C.c_int32), ("global_num_envs", C.c_int64),
                ("num_members", C.c_int32), ("flags", C.c_uint32), ("seed", C.c_uint64),
                ("devices", C.POINTER(C.c_int32)), ("gather", C.c_int32), ("max_episode_steps", C.c_int32)]


class IpcHandle(C.Structure):
    _fields_ = [("bytes", C.c_char * 64)]


class RolloutBuffers(C.Structure):
    _fields_ = [("d_obs", C.c_void_p), ("d_reward", C.c_void_p), ("d_done", C.c_void_p)]


ACTIONS_RING, ACTIONS_SAMPLE, ACTIONS_EPSILON_GREEDY = 0, 1, 2


class RolloutSpec(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("action_source", C.c_int32), ("d_actions", C.c_void_p), ("steps", C.c_int64),
                ("action_stride", C.c_int64), ("ring", C.c_int64), ("action_seed", C.c_uint64), ("action_tick0", C.c_uint64),
                ("epsilon", C.c_float), ("record_flags", C.c_int32), ("d_rec_obs", C.c_void_p), ("d_rec_reward", C.c_void_p),
                ("d_rec_done", C.c_void_p), ("d_rec_actions", C.c_void_p),
                ("d_ep_step", C.c_void_p), ("d_ep_lane", C.c_void_p), ("d_ep_return", C.c_void_p), ("d_ep_length", C.c_void_p),
                ("ep_capacity", C.c_int64), ("d_ep_count", C.c_void_p)]


class Counters(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("reserved", C.c_uint32), ("tick", C.c_uint64),
                ("lane_steps", C.c_uint64), ("stepped_after_done", C.c_uint64),
                ("last_done_count", C.c_int64)]


_H = C.c_void_p
_P = C.c_void_p

# name -> (restype, argtypes); exactly the functions include/gymnet_amd.h declares
PROTOTYPES = {
    "gymnet_abi_version": (C.c_int, []),
    "gymnet_status_string": (C.c_char_p, [C.c_int]),
    "gymnet_last_error": (C.c_char_p, []),
    "gymnet_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "gymnet_env_describe": (C.c_int, [C.c_int, C.POINTER(EnvInfo)]),
    "gymnet_vecenv_create": (C.c_int, [C.POINTER(Config), C.POINTER(_H)]),
    "gymnet_vecenv_destroy": (C.c_int, [_H]),
    "gymnet_vecenv_seed": (C.c_int, [_H, C.c_uint64]),
    "gymnet_vecenv_seed_lanes": (C.c_int, [_H, _P, C.c_int64]),
    "gymnet_vecenv_reset": (C.c_int, [_H, _P]),
    "gymnet_vecenv_reset_where": (C.c_int, [_H, _P, _P]),
    "gymnet_vecenv_step": (C.c_int, [_H, _P, _P, _P, _P]),
    "gymnet_vecenv_step_broadcast": (C.c_int, [_H, C.c_int32, _P, _P, _P]),
    "gymnet_vecenv_step_async": (C.c_int, [_H, _P]),
    "gymnet_vecenv_step_wait": (C.c_int, [_H, _P, _P, _P]),
    "gymnet_vecenv_read": (C.c_int, [_H, _P, _P, _P]),
    "gymnet_vecenv_reset_device": (C.c_int, [_H]),
    "gymnet_vecenv_reset_where_device": (C.c_int, [_H, _P]),
    "gymnet_vecenv_step_device": (C.c_int, [_H, _P]),
    "gymnet_vecenv_rollout_device": (C.c_int, [_H, _P, C.c_int64, C.c_int64, C.c_int64]),
    "gymnet_vecenv_rollout_fused_device": (C.c_int, [_H, _P, C.c_int64, C.c_int64, C.c_int64, C.POINTER(RolloutBuffers)]),
    "gymnet_vecenv_rollout_fused_ex_device": (C.c_int, [_H, C.POINTER(RolloutSpec)]),
    "gymnet_vecenv_pack_obs_device": (C.c_int, [_H, _P]),
    "gymnet_vecenv_sync": (C.c_int, [_H]),
    "gymnet_vecenv_device_view": (C.c_int, [_H, C.POINTER(DeviceView)]),
    "gymnet_vecenv_launch_policy": (C.c_int, [_H, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                             C.POINTER(C.c_int32)]),
    "gymnet_vecenv_set_launch_policy": (C.c_int, [_H, C.POINTER(LaunchPolicy)]),
    "gymnet_vecenv_get_launch_policy": (C.c_int, [_H, C.POINTER(LaunchPolicy)]),
    "gymnet_vecenv_kernel_name": (C.c_int, [_H, C.c_char_p, C.c_int32]),
    "gymnet_vecenv_host_buffers": (C.c_int, [_H, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "gymnet_vecenv_get_state": (C.c_int, [_H, _P]),
    "gymnet_vecenv_set_state": (C.c_int, [_H, _P]),
    "gymnet_vecenv_get_steps_beyond_done": (C.c_int, [_H, _P]),
    "gymnet_vecenv_set_steps_beyond_done": (C.c_int, [_H, _P]),
    "gymnet_vecenv_get_tick": (C.c_int, [_H, C.POINTER(C.c_uint64)]),
    "gymnet_vecenv_set_tick": (C.c_int, [_H, C.c_uint64]),
    "gymnet_vecenv_counters": (C.c_int, [_H, C.POINTER(Counters)]),
    "gymnet_vecenv_get_array": (C.c_int, [_H, C.c_int32, _P, C.c_int64]),
    "gymnet_vecenv_set_array": (C.c_int, [_H, C.c_int32, _P, C.c_int64]),
    "gymnet_vecenv_get_seed": (C.c_int, [_H, C.POINTER(C.c_uint64), C.POINTER(C.c_int32)]),
    "gymnet_vecenv_done_lanes": (C.c_int, [_H, _P, C.c_int64, C.POINTER(C.c_int64)]),
    "gymnet_vecenv_done_lanes_device": (C.c_int, [_H, _P, _P]),
    "gymnet_vecenv_episode_stats": (C.c_int, [_H, _P, _P]),
    "gymnet_vecenv_final_obs": (C.c_int, [_H, _P]),
    "gymnet_vecenv_done_records": (C.c_int, [_H, _P, _P, _P, _P, C.c_int64, C.POINTER(C.c_int64)]),
    "gymnet_vecenv_done_records_device": (C.c_int, [_H, _P, _P, _P, _P, C.c_int64, _P]),
    "gymnet_sample_discrete_device": (C.c_int, [C.c_int, _P, _P, C.c_int64, C.c_int32, C.c_int32,
                                                C.c_uint64, C.c_uint64, C.c_uint64]),
    "gymnet_sample_discrete_masked_device": (C.c_int, [C.c_int, _P, _P, C.c_int64, C.c_int32, C.c_int32, _P, C.c_int64,
                                                       C.c_uint64, C.c_uint64, C.c_uint64]),
    "gymnet_vecenv_sample_actions_masked_device": (C.c_int, [_H, _P, _P, C.c_int64, C.c_uint64, C.c_uint64]),
    "gymnet_sample_box_device": (C.c_int, [C.c_int, _P, _P, C.c_int64, C.c_float, C.c_float,
                                           C.c_uint64, C.c_uint64, C.c_uint64]),
    "gymnet_sample_box_elementwise_device": (C.c_int, [C.c_int, _P, _P, C.c_int64, C.c_int32, _P, _P, C.c_uint64, C.c_uint64, C.c_uint64]),
    "gymnet_vecenv_sample_actions_device": (C.c_int, [_H, _P, C.c_uint64, C.c_uint64]),
    "gymnet_vecenv_sample_actions": (C.c_int, [_H, _P, C.c_uint64, C.c_uint64]),
    "gymnet_vecenv_compose_actions_device": (C.c_int, [_H, _P, C.c_float, _P, C.c_uint64, C.c_uint64]),
    "gymnet_peer_buffer_create": (C.c_int, [C.c_int, C.c_int64, C.POINTER(C.c_void_p), C.POINTER(IpcHandle)]),
    "gymnet_peer_buffer_open": (C.c_int, [C.c_int, C.POINTER(IpcHandle), C.POINTER(C.c_void_p)]),
    "gymnet_peer_buffer_close": (C.c_int, [C.c_int, _P]),
    "gymnet_peer_buffer_destroy": (C.c_int, [C.c_int, _P]),
    "gymnet_push_obs_device": (C.c_int, [C.c_int, _P, _P, C.POINTER(C.c_void_p), C.c_int32, C.c_int64]),
    "gymnet_group_create": (C.c_int, [C.POINTER(GroupConfig), C.POINTER(_H)]),
    "gymnet_group_destroy": (C.c_int, [_H]),
    "gymnet_group_size": (C.c_int, [_H, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    "gymnet_group_member": (C.c_int, [_H, C.c_int32, C.POINTER(_H)]),
    "gymnet_group_seed": (C.c_int, [_H, C.c_uint64]),
    "gymnet_group_reset_device": (C.c_int, [_H]),
    "gymnet_group_step_device": (C.c_int, [_H, C.POINTER(C.c_void_p)]),
    "gymnet_group_rollout_device": (C.c_int, [_H, C.POINTER(C.c_void_p), C.c_int64, C.c_int64, C.c_int64]),
    "gymnet_group_allgather_obs": (C.c_int, [_H]),
    "gymnet_group_wait_gather": (C.c_int, [_H]),
    "gymnet_group_global_obs": (C.c_int, [_H, C.c_int32, C.POINTER(C.c_void_p)]),
    "gymnet_group_read_replica": (C.c_int, [_H, C.c_int32, _P]),
    "gymnet_group_sync": (C.c_int, [_H]),
    "gymnet_group_reset": (C.c_int, [_H, _P]),
    "gymnet_group_step": (C.c_int, [_H, _P, _P, _P, _P]),
}

_lib = None


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  PyTorch wheels bundle their own libamdhip64.so (soname
    libamdhip64.so.7, the same soname this library needs); if this library were loaded first it would
    bring in /opt/rocm's copy and a later `import torch` would start a SECOND runtime that finds no GPU.
    So when torch is installed (it is only located, not imported) its copy is loaded first and this
    library binds to it by soname.  GYMNET_HIP_RUNTIME=system opts out (hosts that never load torch)."""
    if os.environ.get("GYMNET_HIP_RUNTIME", "auto") == "system" or "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec and spec.submodule_search_locations:
        p = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(p):
            C.CDLL(p, mode=C.RTLD_GLOBAL)


def load_library():
    """Loads libgymnet_amd.so.  Raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GymNetError(
            f"{LIB_PATH} is missing: the HIP extension has not been built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
    _share_hip_runtime_with_torch()
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    if lib.gymnet_abi_version() != ABI_VERSION:
        raise GymNetError("libgymnet_amd.so ABI version mismatch")
    _lib = lib
    return lib


def last_error():
    msg = load_library().gymnet_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


_ERRORS = {
    ERR_INVALID_ARG: ValueError,                 # ArgumentException
    ERR_INVALID_ACTION: InvalidActionError,
    ERR_NO_DEVICE: NoDeviceError,
    ERR_ALREADY_STEPPING: AlreadySteppingError,
    ERR_NOT_STEPPING: NotSteppingError,
    ERR_OOM: MemoryError,
    ERR_UNSUPPORTED: NotImplementedError,        # NotSupportedException
    ERR_RCCL: RuntimeError,
}


def check(status):
    """Maps a gymnet_status to the exception the reference would throw for the same condition."""
    if status == OK:
        return
    exc = _ERRORS.get(status, GymNetError)
    msg = last_error() or load_library().gymnet_status_string(status).decode()
    if exc is InvalidActionError:
        raise InvalidActionError(msg)
    raise exc(f"{msg} (gymnet status {status})")


def device_count():
    """Number of HIP devices; 0 when there is none (no exception: callers probe with this)."""
    n = C.c_int(0)
    load_library().gymnet_device_count(C.byref(n))
    return n.value


def env_describe(env_id):
    info = EnvInfo()
    check(load_library().gymnet_env_describe(int(env_id), C.byref(info)))
    return info
