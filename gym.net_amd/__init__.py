"""gym.net_amd — MI355X-native batched classic-control environment engine.

One hot path of SciSharp/Gym.NET — the per-instance Env.Step()/Reset() of the classic-control
environments — rebuilt as hand-written HIP kernels for gfx950 behind a C ABI
(include/gymnet_amd.h), plus this thin host-side mirror of the reference's Env / VecEnv / Space /
Step interface.  The directory name contains a dot, so it is loaded by path
(`__graft_entry__.load_package()` registers it as module `gymnet_amd`).
"""
from . import _capi
from ._capi import (ENV_IDS, FLAG_AUTORESET, FLAG_COMPACT_RECORDS_ONLY, FLAG_DONE_LIST, FLAG_DOUBLE_BUFFER, FLAG_EPISODE_STATS, FLAG_F64,
                    FLAG_FINAL_OBS, FLAG_VALIDATE_ACTIONS, LIB_PATH, device_count, env_describe, load_library)
from .errors import (AlreadySteppingError, GymNetError, InvalidActionError, NoDeviceError,
                     NotSteppingError)
from .sharding import ShardPlan, ShardedVectorEnv
from .spaces import Box, Discrete, Space
from .step import Step
from .vector_env import (AcrobotEnv, BatchStep, CartPoleEnv, DummyVecEnv, GpuEnv, GroupVectorEnv, MountainCarEnv, PendingStep,
                         PendulumEnv, VectorEnv)

__all__ = ["VectorEnv", "GroupVectorEnv", "DummyVecEnv", "BatchStep", "PendingStep", "GpuEnv", "CartPoleEnv", "PendulumEnv", "MountainCarEnv",
           "AcrobotEnv", "Space", "Box", "Discrete", "Step", "InvalidActionError", "AlreadySteppingError",
           "NotSteppingError", "GymNetError", "NoDeviceError", "ShardPlan", "ShardedVectorEnv", "device_count",
           "env_describe", "load_library", "LIB_PATH", "ENV_IDS"]
