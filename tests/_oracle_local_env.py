"""TEST-ONLY stand-in for the HIP VectorEnv, backed by the oracle's "kernel semantics" CartPole (float32, or the float64 twin of
GYMNET_FLAG_F64 with dtype="float64").

It exists so that the host-side sharding logic (gym.net_amd/sharding.py: lane offsets, the rank-major gather
buffer the rank's state lives in, the all-gather) can be exercised with world_size 2 on the gloo backend in a
container that has no GPU.  The product never imports this file (nothing under tests/ ships)."""
import ctypes

import numpy as np

from oracle import capi as oracle


class OracleLocalEnv:
    def __init__(self, env, num_envs, lane_offset, seed, auto_reset, ext_obs, ext_obs_stride, device, stream, ext_obs_alt=None, dtype="float32"):
        assert env in ("CartPole-v1", 0) and auto_reset
        self.n, self.lane_offset, self.seed, self.tick = num_envs, lane_offset, seed, 0
        self.f64 = np.dtype(dtype) == np.float64

        def view(ptr):
            buf = ((ctypes.c_double if self.f64 else ctypes.c_float) * (4 * ext_obs_stride)).from_address(int(ptr))
            return np.ctypeslib.as_array(buf).reshape(4, ext_obs_stride)[:, :num_envs]   # zero-copy view
        self.state = view(ext_obs)
        self.alt = view(ext_obs_alt) if ext_obs_alt is not None else None    # GYMNET_FLAG_DOUBLE_BUFFER twin
        self.cur = 0
        self.reward = np.zeros(num_envs, np.float32)
        self.done = np.zeros(num_envs, np.uint8)

    def ResetDevice(self):
        if self.f64:
            self.state[:] = oracle.cartpole_reset_f64(self.seed, self.lane_offset, self.tick, self.n)
        else:
            self.state[:] = oracle.cartpole_reset(self.seed, self.lane_offset, self.tick, self.n)
        self.tick += 1

    def StepDevice(self, actions):
        a = actions.numpy() if hasattr(actions, "numpy") else np.asarray(actions)
        if self.f64:
            s, r, d = oracle.cartpole_autoreset_step_f64(self.seed, self.lane_offset, self.tick, self.state, a.astype(np.int32))
        else:
            s, r, d, _ = oracle.cartpole_step(self.state, a.astype(np.int32), dtype=np.float32)
            fresh = oracle.cartpole_reset(self.seed, self.lane_offset, self.tick, self.n)
            fin = d.astype(bool)
            s[:, fin] = fresh[:, fin]
        if self.alt is not None:                 # ping-pong: write the other buffer, which becomes current
            self.alt[:] = s
            self.state, self.alt = self.alt, self.state
            self.cur ^= 1
        else:
            self.state[:] = s
        self.reward[:], self.done[:] = r, d
        self.tick += 1

    def ObsBufferIndex(self):
        return self.cur

    def Sync(self):
        pass

    def Close(self):
        pass
