"""VectorEnv — the batched replacement for Gym.NET's VecEnv / Env hot path, over the C ABI.

Mirrors (paths relative to the Gym.NET tree):
  IVecEnv / VecEnv     src/Gym/Envs/IVecEnv.cs:8-19, src/Gym/Envs/VecEnv.cs:12-93
  VecEnvWrapper        src/Gym/Envs/VecEnvWrapper.cs:9-30      (the sequential map this replaces)
  IEnv / Env           src/Gym/Envs/IEnv.cs:11-22, src/Gym/Envs/Env.cs:13-41
  CartPoleEnv          src/Gym.Environments/Envs/Classic/CartPoleEnv.cs:43-67,137-198

Same member names (Reset, Step, StepAsync, Seed, Close, ActionSpace, ObservationSpace, Metadata,
RewardRange, NumberOfEnvironments), same argument meaning, same error behaviour.  Deviations, all
forced by scale: results are ARRAYS — Reset() returns one ndarray [N, D] instead of
NDArray[N], Step() returns a BatchStep whose items are the reference's Step records — and Step()
also accepts one action PER LANE.  `Environments` is empty: 2^20 IEnv objects are never built.

Everything here is a thin shim over libgymnet_amd.so: all compute happens in HIP kernels.  NDArray
is numpy.ndarray on this side of the boundary.
"""
import ctypes as C
import enum

import numpy as np

from . import _capi as capi
from .errors import AlreadySteppingError, NotSteppingError
from .spaces import Box, Discrete
from .step import Step


def _ptr(x):
    """Device pointer from an int, a ctypes pointer or anything with .data_ptr() (torch tensor)."""
    if x is None:
        return None
    if hasattr(x, "data_ptr"):
        return C.c_void_p(x.data_ptr())
    if isinstance(x, C.c_void_p):
        return x
    return C.c_void_p(int(x))


def _host(a):
    return a.ctypes.data_as(C.c_void_p)


class BatchStep:
    """What `Step[]` (IVecEnv.cs:15) becomes at 2^20 lanes: three arrays.  Indexing / iterating yields the
    reference's per-env Step records (Step.cs:7-20), materialised lazily."""
    __slots__ = ("Observation", "Reward", "Done", "Information", "Truncated")

    def __init__(self, observation, reward, done, information=None, truncated=None):
        self.Observation, self.Reward, self.Done, self.Information = observation, reward, done, information
        self.Truncated = truncated          # bool [N]: the episode ended by the max_episode_steps extension (done byte, bit 1)

    def __len__(self):
        return self.Reward.shape[0]

    def __getitem__(self, i):
        info = self.Information
        if self.Truncated is not None and self.Truncated[i]:
            info = {"TimeLimit.truncated": True}                  # upstream gym's TimeLimit convention (the reference has none)
        return Step(self.Observation[i], float(self.Reward[i]), bool(self.Done[i]), info)

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]

    def ToSteps(self):
        return list(self)


class PendingStep:
    """Task<Step[]> of VecEnv.StepAsync (VecEnv.cs:63-65): Result() blocks like Task.Result."""

    def __init__(self, env):
        self._env, self._result = env, None

    def Result(self):
        if self._result is None:
            self._result = self._env.StepWait()
        return self._result


class VectorEnv:
    def __init__(self, env="CartPole-v1", num_envs=1, device=0, seed=0, auto_reset=False,
                 validate_actions=False, done_list=False, episode_stats=False, final_obs=False,
                 lane_offset=0, stream=None, ext_obs=None, ext_obs_stride=0, max_episode_steps=0,
                 double_buffer=False, ext_obs_alt=None, dtype=np.float32, compact_records_only=False, launch_policy=None, resident=False):
        """dtype=np.float64 (CartPole only) selects GYMNET_FLAG_F64: the reference's own float64 arithmetic, float64 state and
        float64 observations (what CartPoleEnv.Step actually returns, CartPoleEnv.cs:166,185); the default float32 is the
        engine's structure-of-arrays hot path.  resident=True (num_envs <= 64): GYMNET_FLAG_RESIDENT — Step / Reset / ResetWhere(None)
        are served by a resident single-wave kernel through a mailbox in pinned host memory (no launch, no synchronize per step).  launch_policy: dict for SetLaunchPolicy (probes / tests that pin a kernel form)."""
        env_id = capi.ENV_IDS[env] if isinstance(env, str) else int(env)
        self._lib = capi.load_library()
        self._info = capi.env_describe(env_id)
        self._dtype = np.dtype(dtype)
        if self._dtype not in (np.dtype(np.float32), np.dtype(np.float64)):
            raise ValueError("dtype must be float32 or float64")
        flags = ((capi.FLAG_AUTORESET if auto_reset else 0) | (capi.FLAG_VALIDATE_ACTIONS if validate_actions else 0)
                 | (capi.FLAG_DONE_LIST if done_list else 0) | (capi.FLAG_EPISODE_STATS if episode_stats else 0)
                 | (capi.FLAG_FINAL_OBS if final_obs else 0) | (capi.FLAG_DOUBLE_BUFFER if double_buffer else 0)
                 | (capi.FLAG_F64 if self._dtype == np.float64 else 0)
                 | (capi.FLAG_COMPACT_RECORDS_ONLY if compact_records_only else 0)
                 | (capi.FLAG_RESIDENT if resident else 0))
        cfg = capi.Config(struct_size=C.sizeof(capi.Config), env_id=env_id, num_envs=int(num_envs),
                          lane_offset=int(lane_offset), device=int(device), flags=flags,
                          seed=int(seed) & 0xFFFFFFFFFFFFFFFF, stream=_ptr(stream), d_ext_obs=_ptr(ext_obs),
                          ext_obs_stride=int(ext_obs_stride), max_episode_steps=int(max_episode_steps), reserved=0,
                          d_ext_obs_alt=_ptr(ext_obs_alt))
        self._h = C.c_void_p()
        self._owns_handle = True
        self._bookkeeping = bool(episode_stats or max_episode_steps)
        self._final_obs = bool(final_obs)
        self.Resident = bool(resident)
        capi.check(self._lib.gymnet_vecenv_create(C.byref(cfg), C.byref(self._h)))
        self._describe(env_id, num_envs, auto_reset)
        if launch_policy:
            self.SetLaunchPolicy(**launch_policy)

    @classmethod
    def _borrow(cls, handle, env_id, num_envs, auto_reset, dtype=np.float32):
        """A VectorEnv view over a handle somebody else owns (a member of a GroupVectorEnv): Close() does not destroy it."""
        self = cls.__new__(cls)
        self._lib = capi.load_library()
        self._info = capi.env_describe(env_id)
        self._h = handle
        self._owns_handle = False
        self._bookkeeping = True            # a group member: flags unknown here (DoneRecords() asks for everything it may have)
        self._final_obs = False
        self.Resident = False
        self._dtype = np.dtype(dtype)
        self._describe(env_id, num_envs, auto_reset)
        return self

    def _describe(self, env_id, num_envs, auto_reset):
        i = self._info
        self.EnvId = env_id
        self.Name = i.name.decode()
        self.NumberOfEnvironments = int(num_envs)                                   # VecEnv.cs:14,24
        self.StateDim, self.ObsDim = int(i.state_dim), int(i.obs_dim)
        self.AutoReset = bool(auto_reset)
        # spaces exactly as the env ctor builds them (CartPoleEnv.cs:46-48)
        if i.action_is_box:
            self.ActionSpace = Box(np.array([i.action_low], np.float32), np.array([i.action_high], np.float32), dtype=np.float32)
            self._adtype = np.float32
        else:
            self.ActionSpace = Discrete(int(i.action_n))
            self._adtype = np.int32
        # the DECLARED dtype stays float32 (CartPoleEnv.cs:48) even when the arrays returned are float64, exactly as in the reference
        self.ObservationSpace = Box(np.array(i.obs_low[:self.ObsDim], np.float32), np.array(i.obs_high[:self.ObsDim], np.float32), dtype=np.float32)
        self.Dtype = self._dtype            # dtype of the observation / state arrays this handle hands out
        self.Metadata = {"render.modes": ["human", "rgb_array"], "video.frames_per_second": 50}   # CartPoleEnv.cs:51
        self.RewardRange = (float(i.reward_low), float(i.reward_high))
        self.Environments = []          # VecEnv.cs:17 — deliberately empty, see module docstring
        self.AlgorithmicBytesPerStep = int(i.algorithmic_bytes_per_step)
        self.TrafficBytesPerStep = int(i.traffic_bytes_per_step)      # < algorithmic where a state row is stored once, in the observation

    # ---- lifecycle ------------------------------------------------------------------------------
    def Close(self):                                                                 # VecEnvWrapper.cs:26-30
        if getattr(self, "_h", None) is not None and self._h:
            self._pinned = None                     # views over memory the handle owns
            if self._owns_handle:
                self._lib.gymnet_vecenv_destroy(self._h)
            self._h = C.c_void_p()

    Dispose = Close                                                                  # Env.cs:38-40

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.Close()

    def __del__(self):
        try:
            self.Close()
        except Exception:
            pass

    def Seed(self, seed):                                                            # VecEnv.cs:44-53
        if isinstance(seed, (int, np.integer)):
            capi.check(self._lib.gymnet_vecenv_seed(self._h, int(seed) & 0xFFFFFFFFFFFFFFFF))
        else:
            s = np.ascontiguousarray(np.asarray(seed, dtype=np.int64).astype(np.uint64))
            capi.check(self._lib.gymnet_vecenv_seed_lanes(self._h, _host(s), s.shape[0]))

    # ---- host-boundary path ---------------------------------------------------------------------
    def _outs(self):
        n = self.NumberOfEnvironments
        return (np.empty((n, self.ObsDim), self._dtype), np.empty(n, np.float32), np.empty(n, np.uint8))

    def Reset(self):                                                                 # VecEnvWrapper.cs:18-20
        obs = np.empty((self.NumberOfEnvironments, self.ObsDim), self._dtype)
        capi.check(self._lib.gymnet_vecenv_reset(self._h, _host(obs)))
        return obs

    def ResetWhere(self, mask=None):
        """Batched `if (done) Reset()` (README.md:36-40). mask None = lanes whose last Done flag is set."""
        obs = np.empty((self.NumberOfEnvironments, self.ObsDim), self._dtype)
        m = None if mask is None else np.ascontiguousarray(np.asarray(mask).astype(np.uint8))
        if m is not None and m.shape[0] != self.NumberOfEnvironments:
            raise ValueError("mask length must equal NumberOfEnvironments")
        capi.check(self._lib.gymnet_vecenv_reset_where(self._h, None if m is None else _host(m), _host(obs)))
        return obs

    def _actions(self, action):
        a = np.ascontiguousarray(np.asarray(action).reshape(-1).astype(self._adtype, copy=False))
        if a.shape[0] != self.NumberOfEnvironments:
            raise ValueError("Number of actions passed should be equals to number of environments")
        return a

    def Step(self, action):
        """IVecEnv.Step(int action) (IVecEnv.cs:15, VecEnvWrapper.cs:22-24) broadcasts ONE scalar action;
        an array-like gives one action per lane (extension)."""
        obs, rew, done = self._outs()
        if isinstance(action, enum.Enum):                                            # Env<TAction>.Step(TAction), Env.cs:43-53
            action = int(action.value)
        if isinstance(action, (int, np.integer)) and not isinstance(action, (bool, np.bool_)):
            capi.check(self._lib.gymnet_vecenv_step_broadcast(self._h, int(action), _host(obs), _host(rew), _host(done)))
        else:
            a = self._actions(action)
            capi.check(self._lib.gymnet_vecenv_step(self._h, _host(a), _host(obs), _host(rew), _host(done)))
        return BatchStep(obs, rew, done.astype(bool), None, truncated=(done & 2) != 0)

    def HostBuffers(self):
        """(actions, obs, reward, done): numpy views over the library's page-locked, device-mapped host buffers
        (gymnet_vecenv_host_buffers).  Passing them to StepInto / ResetInto runs the host boundary without staging copies."""
        if getattr(self, "_pinned", None) is None:
            a, o, r, d = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
            capi.check(self._lib.gymnet_vecenv_host_buffers(self._h, C.byref(a), C.byref(o), C.byref(r), C.byref(d)))
            n, D = self.NumberOfEnvironments, self.ObsDim

            def view(ptr, ctype, count, dtype, shape):
                return np.frombuffer((ctype * count).from_address(ptr.value), dtype=dtype).reshape(shape)
            self._pinned = (view(a, C.c_int32, n, self._adtype, (n,)),
                            view(o, C.c_double if self._dtype == np.float64 else C.c_float, n * D, self._dtype, (n, D)),
                            view(r, C.c_float, n, np.float32, (n,)), view(d, C.c_uint8, n, np.uint8, (n,)))
        return self._pinned

    def StepInto(self, actions, obs_out, reward_out, done_out):
        """gymnet_vecenv_step with CALLER-OWNED buffers (what a C# host with long-lived NDArrays does): actions [N] of the
        action dtype, obs_out float32 [N, D], reward_out float32 [N], done_out uint8 [N]; nothing is allocated."""
        n = self.NumberOfEnvironments
        if (actions.dtype != self._adtype or obs_out.dtype != self._dtype or reward_out.dtype != np.float32 or done_out.dtype != np.uint8
                or actions.shape != (n,) or obs_out.shape != (n, self.ObsDim) or reward_out.shape != (n,) or done_out.shape != (n,)
                or not (actions.flags.c_contiguous and obs_out.flags.c_contiguous and reward_out.flags.c_contiguous and done_out.flags.c_contiguous)):
            raise ValueError("StepInto needs C-contiguous buffers of the exact dtypes and shapes")
        capi.check(self._lib.gymnet_vecenv_step(self._h, _host(actions), _host(obs_out), _host(reward_out), _host(done_out)))

    def ResetInto(self, obs_out):
        if obs_out.dtype != self._dtype or obs_out.shape != (self.NumberOfEnvironments, self.ObsDim) or not obs_out.flags.c_contiguous:
            raise ValueError("ResetInto needs a C-contiguous [N, D] buffer of the handle's dtype")
        capi.check(self._lib.gymnet_vecenv_reset(self._h, _host(obs_out)))

    def StepAsync(self, action):                                                     # VecEnv.cs:63-65
        if isinstance(action, enum.Enum):
            action = int(action.value)
        if isinstance(action, (int, np.integer)):
            action = np.full(self.NumberOfEnvironments, action, dtype=self._adtype)
        a = self._actions(action)
        capi.check(self._lib.gymnet_vecenv_step_async(self._h, _host(a)))
        return PendingStep(self)

    def StepWait(self):
        obs, rew, done = self._outs()
        capi.check(self._lib.gymnet_vecenv_step_wait(self._h, _host(obs), _host(rew), _host(done)))
        return BatchStep(obs, rew, done.astype(bool), None, truncated=(done & 2) != 0)

    def Read(self):
        obs, rew, done = self._outs()
        capi.check(self._lib.gymnet_vecenv_read(self._h, _host(obs), _host(rew), _host(done)))
        return BatchStep(obs, rew, done.astype(bool), None, truncated=(done & 2) != 0)

    def SampleActions(self, seed=0, tick=0):
        """ActionSpace.Sample() for every lane, on the device (TrainingPlaySession.cs:46-49 batched)."""
        a = np.empty(self.NumberOfEnvironments, self._adtype)
        capi.check(self._lib.gymnet_vecenv_sample_actions(self._h, _host(a), int(seed), int(tick)))
        return a

    # ---- device-resident path (device pointers: ints, ctypes pointers or torch tensors) -------------
    def ResetDevice(self):
        capi.check(self._lib.gymnet_vecenv_reset_device(self._h))

    def ResetWhereDevice(self, d_mask=None):
        capi.check(self._lib.gymnet_vecenv_reset_where_device(self._h, _ptr(d_mask)))

    def StepDevice(self, d_actions):
        capi.check(self._lib.gymnet_vecenv_step_device(self._h, _ptr(d_actions)))

    def RolloutDevice(self, d_actions, steps, action_stride, ring):
        capi.check(self._lib.gymnet_vecenv_rollout_device(self._h, _ptr(d_actions), int(steps), int(action_stride), int(ring)))

    def RolloutFusedDevice(self, d_actions, steps, action_stride=0, ring=1, rec_obs=None, rec_reward=None, rec_done=None, actions="ring",
                           action_seed=0, action_tick0=0, epsilon=0.0, rec_actions=None, episodes=None):
        """`steps` vector steps in ONE kernel launch (state stays in registers); optional device-side rollout
        buffers rec_obs [T][D][N] (of the handle's dtype: float64 for a float64 handle), rec_reward [T][N], rec_done [T][N]
        (ReplayMemory.cs:25-67, batched).  gymnet_vecenv_rollout_fused_ex_device:
          actions   "ring" (d_actions[t % ring]), "sample" (ActionSpace.Sample() drawn in the kernel: the values SampleActionsDevice
                    (seed=action_seed, tick=action_tick0 + t) would write; d_actions may be None) or "epsilon_greedy" (ComposeActionsDevice
                    over d_actions as the policy's actions, TrainingPlaySession.cs:46-52)
          rec_actions  [T][N] device buffer for the actions taken
          episodes  dict(step=, lane=, ret=, length=, capacity=, count=): device arrays for the compact records of the episodes
                    that end during the rollout (any array may be omitted; count: uint32[2] = records written, episodes ended);
                    no_overflow=True selects the 8 % faster kernel variant that may drop records of very unevenly finishing lanes
                    below `capacity` (GYMNET_RECORDS_NO_OVERFLOW; count[1] > count[0] says so);
                    needs a bookkeeping handle (BasePlaySession.cs:58-69)."""
        src = {"ring": capi.ACTIONS_RING, "sample": capi.ACTIONS_SAMPLE, "epsilon_greedy": capi.ACTIONS_EPSILON_GREEDY}[actions]
        ep = episodes or {}
        unknown = set(ep) - {"step", "lane", "ret", "length", "capacity", "count", "no_overflow"}
        if unknown:
            raise TypeError(f"unknown episode record field(s): {sorted(unknown)}")
        spec = capi.RolloutSpec(struct_size=C.sizeof(capi.RolloutSpec), action_source=src, d_actions=_ptr(d_actions), steps=int(steps),
                                action_stride=int(action_stride), ring=int(ring), action_seed=int(action_seed) & 0xFFFFFFFFFFFFFFFF,
                                action_tick0=int(action_tick0), epsilon=float(epsilon), record_flags=capi.RECORDS_NO_OVERFLOW if ep.get("no_overflow") else 0,
                                d_rec_obs=_ptr(rec_obs), d_rec_reward=_ptr(rec_reward), d_rec_done=_ptr(rec_done), d_rec_actions=_ptr(rec_actions),
                                d_ep_step=_ptr(ep.get("step")), d_ep_lane=_ptr(ep.get("lane")), d_ep_return=_ptr(ep.get("ret")),
                                d_ep_length=_ptr(ep.get("length")), ep_capacity=int(ep.get("capacity", 0)), d_ep_count=_ptr(ep.get("count")))
        capi.check(self._lib.gymnet_vecenv_rollout_fused_ex_device(self._h, C.byref(spec)))

    def SampleActionsDevice(self, d_actions, seed=0, tick=0):
        capi.check(self._lib.gymnet_vecenv_sample_actions_device(self._h, _ptr(d_actions), int(seed), int(tick)))

    def SampleActionsMaskedDevice(self, d_actions, d_mask, per_lane=True, seed=0, tick=0):
        """ActionSpace.Sample(mask) for every lane (Discrete.cs:18-26): d_mask uint8 [N][n] (per_lane) or one shared row [n]."""
        stride = self.ActionSpace.N if per_lane else 0
        capi.check(self._lib.gymnet_vecenv_sample_actions_masked_device(self._h, _ptr(d_actions), _ptr(d_mask), int(stride),
                                                                        int(seed), int(tick)))

    def ComposeActionsDevice(self, d_policy_actions, epsilon, d_actions_out, seed=0, tick=0):
        """Batched epsilon-greedy ComposeAction (TrainingPlaySession.cs:46-52): with probability epsilon the lane's
        action is ActionSpace.Sample(), otherwise the policy's."""
        capi.check(self._lib.gymnet_vecenv_compose_actions_device(self._h, _ptr(d_policy_actions), float(epsilon),
                                                                  _ptr(d_actions_out), int(seed), int(tick)))

    def PackObsDevice(self, d_obs_rowmajor):
        capi.check(self._lib.gymnet_vecenv_pack_obs_device(self._h, _ptr(d_obs_rowmajor)))

    def Sync(self):
        capi.check(self._lib.gymnet_vecenv_sync(self._h))

    def DeviceView(self):
        v = capi.DeviceView()
        capi.check(self._lib.gymnet_vecenv_device_view(self._h, C.byref(v)))
        return v

    def ObsBufferIndex(self):
        """GYMNET_FLAG_DOUBLE_BUFFER: which of the two observation buffers holds the latest observation (0 / 1)."""
        return int(self.DeviceView().obs_buffer)

    def LaunchPolicy(self):
        """The step kernel's launch configuration the handle chose."""
        v, b, nt, sq = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        capi.check(self._lib.gymnet_vecenv_launch_policy(self._h, C.byref(v), C.byref(b), C.byref(nt), C.byref(sq)))
        return {"envs_per_thread": v.value, "block": b.value, "nontemporal_mask": nt.value, "sequential_lanes_per_thread": sq.value}

    _POLICY_FIELDS = ("vec", "block", "nt", "sequential_lanes", "reset_form", "lds_pipe", "occupancy_lds_bytes", "graph")

    def SetLaunchPolicy(self, **fields):
        """gymnet_vecenv_set_launch_policy: override fields of the step kernel's launch configuration (vec, block, nt,
        sequential_lanes, reset_form, lds_pipe, occupancy_lds_bytes, graph; unnamed fields stay).  Every configuration computes
        bit-identical results; a value the handle cannot run raises ValueError.  (This replaces the GYMNET_* environment
        variables of rounds 1-3: the library no longer reads the process environment.)"""
        unknown = set(fields) - set(self._POLICY_FIELDS)
        if unknown:
            raise TypeError(f"unknown launch policy field(s): {sorted(unknown)}")
        p = capi.LaunchPolicy(struct_size=C.sizeof(capi.LaunchPolicy), **{k: int(fields.get(k, -1)) for k in self._POLICY_FIELDS})
        capi.check(self._lib.gymnet_vecenv_set_launch_policy(self._h, C.byref(p)))

    def GetLaunchPolicy(self):
        p = capi.LaunchPolicy()
        capi.check(self._lib.gymnet_vecenv_get_launch_policy(self._h, C.byref(p)))
        return {k: getattr(p, k) for k in self._POLICY_FIELDS}

    def KernelName(self):
        """The template instantiation the next step launch runs, as the launcher itself resolves it
        (e.g. "step_kernel<CartPole,4,true,false,15,1>")."""
        buf = C.create_string_buffer(128)
        capi.check(self._lib.gymnet_vecenv_kernel_name(self._h, buf, 128))
        return buf.value.decode()

    # ---- state access / bookkeeping -----------------------------------------------------------------
    def GetState(self):
        s = np.empty((self.StateDim, self.NumberOfEnvironments), self._dtype)
        capi.check(self._lib.gymnet_vecenv_get_state(self._h, _host(s)))
        return s

    def SetState(self, state_soa):
        s = np.ascontiguousarray(np.asarray(state_soa, dtype=self._dtype))
        if s.shape != (self.StateDim, self.NumberOfEnvironments):
            raise ValueError(f"state must have shape ({self.StateDim}, {self.NumberOfEnvironments})")
        capi.check(self._lib.gymnet_vecenv_set_state(self._h, _host(s)))

    def GetStepsBeyondDone(self):
        b = np.empty(self.NumberOfEnvironments, np.int32)
        capi.check(self._lib.gymnet_vecenv_get_steps_beyond_done(self._h, _host(b)))
        return b

    def SetStepsBeyondDone(self, sbd):
        b = np.ascontiguousarray(np.asarray(sbd, dtype=np.int32))
        if b.shape[0] != self.NumberOfEnvironments:
            raise ValueError("length must equal NumberOfEnvironments")
        capi.check(self._lib.gymnet_vecenv_set_steps_beyond_done(self._h, _host(b)))

    @property
    def Tick(self):
        t = C.c_uint64()
        capi.check(self._lib.gymnet_vecenv_get_tick(self._h, C.byref(t)))
        return t.value

    @Tick.setter
    def Tick(self, value):
        capi.check(self._lib.gymnet_vecenv_set_tick(self._h, int(value)))

    def Counters(self):
        c = capi.Counters()
        capi.check(self._lib.gymnet_vecenv_counters(self._h, C.byref(c)))
        return {"tick": c.tick, "lane_steps": c.lane_steps, "stepped_after_done": c.stepped_after_done,
                "last_done_count": c.last_done_count}

    def DoneLanes(self):
        n = self.NumberOfEnvironments
        lanes = np.empty(n, np.int32)
        cnt = C.c_int64()
        capi.check(self._lib.gymnet_vecenv_done_lanes(self._h, _host(lanes), n, C.byref(cnt)))
        return lanes[:cnt.value].copy()

    def DoneLanesDevice(self, d_lanes_out, d_count_out):
        """Compact list of the lanes that finished in the last step, left on the device (stream-ordered)."""
        capi.check(self._lib.gymnet_vecenv_done_lanes_device(self._h, _ptr(d_lanes_out), _ptr(d_count_out)))

    def DoneRecords(self, episode=None, final_obs=None):
        """Compact records of the lanes that finished in the most recent step (gymnet_vecenv_done_records): dict with
        "lanes" int32 [c] and — by default whenever the handle keeps them — "return" float32 [c], "length" int32 [c],
        "final_obs" [c, D] of the handle's dtype, all in the same (unspecified) order."""
        n = self.NumberOfEnvironments
        episode = self._bookkeeping if episode is None else episode
        final_obs = self._final_obs if final_obs is None else final_obs
        lanes = np.empty(n, np.int32)
        ret = np.empty(n, np.float32) if episode else None
        ln = np.empty(n, np.int32) if episode else None
        fo = np.empty((n, self.ObsDim), self._dtype) if final_obs else None
        cnt = C.c_int64()
        capi.check(self._lib.gymnet_vecenv_done_records(self._h, _host(lanes), None if ret is None else _host(ret), None if ln is None else _host(ln),
                                                        None if fo is None else _host(fo), n, C.byref(cnt)))
        c = cnt.value
        out = {"lanes": lanes[:c].copy()}
        if episode:
            out["return"], out["length"] = ret[:c].copy(), ln[:c].copy()
        if final_obs:
            out["final_obs"] = fo[:c].copy()
        return out

    def DoneRecordsDevice(self, d_lanes, d_return, d_length, d_final_obs, capacity, d_count):
        capi.check(self._lib.gymnet_vecenv_done_records_device(self._h, _ptr(d_lanes), _ptr(d_return), _ptr(d_length), _ptr(d_final_obs),
                                                               int(capacity), _ptr(d_count)))

    def EpisodeStats(self):
        n = self.NumberOfEnvironments
        ret, ln = np.empty(n, np.float32), np.empty(n, np.int32)
        capi.check(self._lib.gymnet_vecenv_episode_stats(self._h, _host(ret), _host(ln)))
        return ret, ln

    def FinalObs(self):
        o = np.empty((self.NumberOfEnvironments, self.ObsDim), self._dtype)
        capi.check(self._lib.gymnet_vecenv_final_obs(self._h, _host(o)))
        return o

    # ---- per-lane arrays by id (gymnet_vecenv_get_array / _set_array) -------------------------------------------------
    _ARRAYS = {"reward": (capi.ARRAY_REWARD, np.float32), "done": (capi.ARRAY_DONE, np.uint8),
               "steps_beyond_done": (capi.ARRAY_STEPS_BEYOND_DONE, np.int32),
               "episode_return": (capi.ARRAY_EPISODE_RETURN, np.float32), "episode_length": (capi.ARRAY_EPISODE_LENGTH, np.int32),
               "finished_return": (capi.ARRAY_FINISHED_RETURN, np.float32), "finished_length": (capi.ARRAY_FINISHED_LENGTH, np.int32),
               "final_obs": (capi.ARRAY_FINAL_OBS, np.float32), "lane_seeds": (capi.ARRAY_LANE_SEEDS, np.uint64)}

    def GetArray(self, name):
        """One of the handle's per-lane arrays by name (see _ARRAYS); "final_obs" is structure-of-arrays [D, N].
        Raises NotImplementedError when the handle's configuration does not keep that array."""
        which, dt = self._ARRAYS[name]
        shape = (self.ObsDim, self.NumberOfEnvironments) if name == "final_obs" else (self.NumberOfEnvironments,)
        if name == "final_obs":
            dt = self._dtype                       # terminal observations carry the handle's state scalar
        a = np.empty(shape, dt)
        capi.check(self._lib.gymnet_vecenv_get_array(self._h, which, _host(a), a.nbytes))
        return a

    def SetArray(self, name, value):
        which, dt = self._ARRAYS[name]
        if name == "final_obs":
            dt = self._dtype
        a = np.ascontiguousarray(np.asarray(value, dtype=dt))
        capi.check(self._lib.gymnet_vecenv_set_array(self._h, which, _host(a), a.nbytes))

    def GetSeed(self):
        """(Philox key in use, whether per-lane keys from Seed(int[]) are active)."""
        s, per = C.c_uint64(), C.c_int32()
        capi.check(self._lib.gymnet_vecenv_get_seed(self._h, C.byref(s), C.byref(per)))
        return s.value, bool(per.value)

    # ---- checkpoint / resume (an extension: the reference has none; everything the engine needs to continue bit for bit) ----
    def Checkpoint(self):
        """EVERY piece of state a handle carries, for ANY configuration (get_state / set_state double as checkpoint /
        resume): the state SoA, the engine tick (the Philox counter word), the seed, and whichever per-lane arrays the handle
        keeps — steps_beyond_done (CartPole without auto-reset), the reward / done flags of the last step (ResetWhere(None)
        consumes them), running episode return / length (they drive the max_episode_steps truncation), the dense
        finished-episode views, terminal observations, the per-lane Philox keys of Seed(int[]).  Restore() on a handle
        created with the same arguments + the same actions reproduce the continuation bit for bit
        (tests/test_gpu_cartpole.py::test_checkpoint_*).  Not part of it: the compacted done LIST of the step before the
        checkpoint (a transient result; DoneLanes() on the restored handle reports none until the next step)."""
        seed, per_lane = self.GetSeed()
        ck = {"env_id": self.EnvId, "num_envs": self.NumberOfEnvironments, "dtype": self._dtype.name, "state": self.GetState(),
              "tick": self.Tick, "seed": seed, "arrays": {}}
        for name in self._ARRAYS:
            if name == "lane_seeds" and not per_lane:
                continue
            try:
                ck["arrays"][name] = self.GetArray(name)
            except NotImplementedError:
                pass                                   # this configuration does not keep that array
        return ck

    def Restore(self, ck):
        if ck["env_id"] != self.EnvId or ck["num_envs"] != self.NumberOfEnvironments or ck.get("dtype", "float32") != self._dtype.name:
            raise ValueError("checkpoint belongs to a different environment / batch size / dtype")
        self.Seed(int(ck["seed"]))                     # also rewinds the tick and drops per-lane keys; both are set again below
        for name, value in ck["arrays"].items():
            self.SetArray(name, value)                 # NotImplementedError: this handle lacks an array the checkpoint carries
        self.SetState(ck["state"])
        self.Tick = ck["tick"]

    # VecEnv.get_attr / set_attr (VecEnv.cs:74-92) select over IEnv objects; here the per-lane
    # attributes that exist are exposed by name.
    def get_attr(self, name):
        if name is None:
            raise ValueError("selector")                       # ArgumentNullException
        if name == "state":
            return self.GetState().T.copy()
        if name == "steps_beyond_done":
            return self.GetStepsBeyondDone()
        raise AttributeError(name)

    def set_attr(self, name, value):
        if value is None:
            raise ValueError("object")                         # ArgumentNullException, VecEnv.cs:88
        if name == "state":
            return self.SetState(np.asarray(value, self._dtype).T)    # the handle's own dtype: float64 state round-trips bit for bit
        if name == "steps_beyond_done":
            return self.SetStepsBeyondDone(value)
        raise AttributeError(name)

    # python-style aliases
    reset, step, seed, close = Reset, Step, Seed, Close


class GroupVectorEnv:
    """One process driving G GPUs (or G logical members on one GPU): the ctypes mirror of gymnet_group_* — what a
    P/Invoking C# host uses instead of torch.distributed.  Member m owns global lanes [m*N/G, (m+1)*N/G); every member
    keeps a replica [G][D][N/G] of all observations on its GPU, completed by AllGatherObs() (hand-written direct push over
    peer-mapped memory, or RCCL).  Same VecEnv surface for the host-boundary path (Reset / Step over the whole batch)."""

    def __init__(self, env="CartPole-v1", global_num_envs=1, num_members=1, devices=None, seed=0, auto_reset=False,
                 gather="direct", overlap=False, validate_actions=False, max_episode_steps=0, episode_stats=False, dtype=np.float32):
        """dtype=np.float64 (CartPole): every member runs GYMNET_FLAG_F64 — the reference's own arithmetic — and the replicas,
        the gather and the host batches carry doubles."""
        env_id = capi.ENV_IDS[env] if isinstance(env, str) else int(env)
        self._lib = capi.load_library()
        self._dtype = np.dtype(dtype)
        if self._dtype not in (np.dtype(np.float32), np.dtype(np.float64)):
            raise ValueError("dtype must be float32 or float64")
        flags = ((capi.FLAG_AUTORESET if auto_reset else 0) | (capi.FLAG_VALIDATE_ACTIONS if validate_actions else 0)
                 | (capi.FLAG_DOUBLE_BUFFER if overlap else 0) | (capi.FLAG_EPISODE_STATS if episode_stats else 0)
                 | (capi.FLAG_F64 if self._dtype == np.float64 else 0))
        mode = {"none": capi.GATHER_NONE, "direct": capi.GATHER_DIRECT, "rccl": capi.GATHER_RCCL}[gather] if isinstance(gather, str) else int(gather)
        devs = None
        if devices is not None:
            if len(devices) != num_members:
                raise ValueError("len(devices) must equal num_members")
            devs = (C.c_int32 * num_members)(*[int(d) for d in devices])
        cfg = capi.GroupConfig(struct_size=C.sizeof(capi.GroupConfig), env_id=env_id, global_num_envs=int(global_num_envs),
                               num_members=int(num_members), flags=flags, seed=int(seed) & 0xFFFFFFFFFFFFFFFF,
                               devices=devs, gather=mode, max_episode_steps=int(max_episode_steps))
        self._g = C.c_void_p()
        capi.check(self._lib.gymnet_group_create(C.byref(cfg), C.byref(self._g)))
        info = capi.env_describe(env_id)
        self.NumberOfEnvironments = int(global_num_envs)
        self.NumMembers = int(num_members)
        self.LanesPerMember = int(global_num_envs) // int(num_members)
        self.ObsDim = int(info.obs_dim)
        self.Overlap = bool(overlap)
        self.Members = []
        for m in range(self.NumMembers):
            h = C.c_void_p()
            capi.check(self._lib.gymnet_group_member(self._g, m, C.byref(h)))
            self.Members.append(VectorEnv._borrow(h, env_id, self.LanesPerMember, auto_reset, self._dtype))
        self.ActionSpace, self.ObservationSpace = self.Members[0].ActionSpace, self.Members[0].ObservationSpace
        self.Metadata, self.RewardRange = self.Members[0].Metadata, self.Members[0].RewardRange
        self._adtype = self.Members[0]._adtype
        self.Environments = []

    def Close(self):
        if getattr(self, "_g", None) is not None and self._g:
            for m in self.Members:
                m.Close()
            self._lib.gymnet_group_destroy(self._g)
            self._g = C.c_void_p()

    Dispose = Close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.Close()

    def __del__(self):
        try:
            self.Close()
        except Exception:
            pass

    def Seed(self, seed):
        capi.check(self._lib.gymnet_group_seed(self._g, int(seed) & 0xFFFFFFFFFFFFFFFF))

    # host-boundary path over the whole batch
    def Reset(self):
        obs = np.empty((self.NumberOfEnvironments, self.ObsDim), self._dtype)
        capi.check(self._lib.gymnet_group_reset(self._g, _host(obs)))
        return obs

    def Step(self, action):
        n = self.NumberOfEnvironments
        if isinstance(action, enum.Enum):
            action = int(action.value)
        if isinstance(action, (int, np.integer)) and not isinstance(action, (bool, np.bool_)):
            action = np.full(n, action, dtype=self._adtype)                          # IVecEnv.Step(int): broadcast
        a = np.ascontiguousarray(np.asarray(action).reshape(-1).astype(self._adtype, copy=False))
        if a.shape[0] != n:
            raise ValueError("Number of actions passed should be equals to number of environments")
        obs, rew, done = np.empty((n, self.ObsDim), self._dtype), np.empty(n, np.float32), np.empty(n, np.uint8)
        capi.check(self._lib.gymnet_group_step(self._g, _host(a), _host(obs), _host(rew), _host(done)))
        return BatchStep(obs, rew, done.astype(bool), None, truncated=(done & 2) != 0)

    # device-resident path
    def ResetDevice(self):
        capi.check(self._lib.gymnet_group_reset_device(self._g))

    def _ptrs(self, d_actions):
        if len(d_actions) != self.NumMembers:
            raise ValueError("one device pointer per member")
        return (C.c_void_p * self.NumMembers)(*[_ptr(p) for p in d_actions])

    def StepDevice(self, d_actions):
        capi.check(self._lib.gymnet_group_step_device(self._g, self._ptrs(d_actions)))

    def RolloutDevice(self, d_actions, steps, action_stride, ring):
        capi.check(self._lib.gymnet_group_rollout_device(self._g, self._ptrs(d_actions), int(steps), int(action_stride), int(ring)))

    def AllGatherObs(self):
        capi.check(self._lib.gymnet_group_allgather_obs(self._g))

    def WaitGather(self):
        capi.check(self._lib.gymnet_group_wait_gather(self._g))

    def GlobalObsPtr(self, member):
        p = C.c_void_p()
        capi.check(self._lib.gymnet_group_global_obs(self._g, int(member), C.byref(p)))
        return p.value

    def ReadReplica(self, member):
        """Member's replica of all observations on the host, [G, D, N/G] (waits for the last gather)."""
        out = np.empty((self.NumMembers, self.ObsDim, self.LanesPerMember), self._dtype)
        capi.check(self._lib.gymnet_group_read_replica(self._g, int(member), _host(out)))
        return out

    def Sync(self):
        capi.check(self._lib.gymnet_group_sync(self._g))

    reset, step, seed, close = Reset, Step, Seed, Close


class DummyVecEnv(VectorEnv):
    """DummyVecEnv (src/Gym/Envs/DummyVecEnv.cs:2-4): a VecEnv around ONE environment — here a 1-lane batch."""

    def __init__(self, env="CartPole-v1", **kw):
        super().__init__(env, 1, **kw)


class GpuEnv:
    """Single-instance `Env` façade (Env.cs:13-41) over a 1-lane VectorEnv: Reset() -> NDArray[D],
    Step(object action) -> Step.  Exists so an existing per-instance loop (README.md:32-52,
    tests/Gym.Tests/Envs/Classic/CartpoleEnvironment.cs:14-35) runs unmodified on the engine."""
    ENV = "CartPole-v1"
    DTYPE = np.float32          # CartPoleEnv overrides: float64, the reference's own arithmetic

    def __init__(self, device=0, seed=0, validate_actions=False, max_episode_steps=0, dtype=None, resident=False):
        """max_episode_steps > 0 adds the TimeLimit wrapper upstream gym registers with the env (500 / 200; an extension: the
        reference has no time limit): the step that reaches the limit returns Done with
        Information["TimeLimit.truncated"] = True — the same shape as the C# GpuEnv (csharp/GpuEnv.cs).
        resident=True (opt-in since round 6): Step / Reset go through GYMNET_FLAG_RESIDENT — a resident single-wave kernel polling
        a mailbox in pinned host memory — instead of a kernel launch + synchronize per call: ~3x lower latency for a loop that does
        nothing but step (README.md:32-52), bit-identical results.  The price: while the kernel waits for the next command it occupies
        the handle's stream for up to ~5 ms, and a device-wide synchronize elsewhere in the process (torch.cuda.synchronize(), a
        caching allocator's hipFree) or serialised dispatch (rocprofv3 --pmc) waits for that — so a process that also trains on the
        GPU keeps the default, one launch per call."""
        self._v = VectorEnv(self.ENV, 1, device=device, seed=seed, auto_reset=False, validate_actions=validate_actions,
                            episode_stats=max_episode_steps > 0, max_episode_steps=max_episode_steps,
                            dtype=self.DTYPE if dtype is None else dtype, resident=resident)
        self.ActionSpace, self.ObservationSpace = self._v.ActionSpace, self._v.ObservationSpace
        self.Metadata, self.RewardRange = self._v.Metadata, self._v.RewardRange
        self._pending = None
        # The per-instance loop is latency-bound (README.md:32-52): long-lived buffers and their ctypes pointers, so a Step() is one
        # ABI call plus a 4-element copy — no array allocation, no argument marshalling per call.
        v = self._v
        self._discrete = isinstance(self.ActionSpace, Discrete)
        self._a = np.zeros(1, v._adtype)
        self._o = np.empty((1, v.ObsDim), v._dtype)
        self._r = np.empty(1, np.float32)
        self._d = np.empty(1, np.uint8)
        self._pa, self._po, self._pr, self._pd = (x.ctypes.data_as(C.c_void_p) for x in (self._a, self._o, self._r, self._d))
        self._native_step, self._native_reset = v._lib.gymnet_vecenv_step, v._lib.gymnet_vecenv_reset

    def Reset(self):                                                                 # CartPoleEnv.cs:63-67
        st = self._native_reset(self._v._h, self._po)
        if st:
            capi.check(st)
        return self._o[0].copy()                                                     # a COPY, like CartPoleEnv.cs:66

    def Step(self, action):                                                          # CartPoleEnv.cs:137-186
        if isinstance(action, enum.Enum):                                            # Env<TAction> where TAction : Enum (Env.cs:43-53)
            action = int(action.value)
        if self._discrete and (isinstance(action, (bool, np.bool_)) or not isinstance(action, (int, np.integer))):
            raise TypeError(f"Specified cast is not valid: {type(action).__name__} -> int")   # InvalidCastException, :138
        self._a[0] = action
        st = self._native_step(self._v._h, self._pa, self._po, self._pr, self._pd)
        if st:
            capi.check(st)
        d = int(self._d[0])
        # upstream gym's TimeLimit convention for the max_episode_steps extension (done byte, bit 1; the reference has none)
        return Step(self._o[0].copy(), float(self._r[0]), d != 0, {"TimeLimit.truncated": True} if d & 2 else None)

    def StepAsync(self, action):                                                     # Env.cs:23-25, 48-50
        if isinstance(action, enum.Enum):
            action = int(action.value)
        a = np.array([action], dtype=self._v._adtype)
        p = self._v.StepAsync(a)

        class _One:
            def Result(_self):
                return p.Result()[0]
        return _One()

    def Render(self, mode="human"):
        return None            # rendering is out of scope for the engine (NullEnvViewer semantics)

    def CloseEnvironment(self):                                                      # CartPoleEnv.cs:189-194
        self._v.Close()

    Close = Dispose = CloseEnvironment

    def Seed(self, seed):                                                            # CartPoleEnv.cs:196-198
        self._v.Seed(int(seed))

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.CloseEnvironment()


class CartPoleEnv(GpuEnv):
    """Drop-in for `new CartPoleEnv()` (README.md:32-52).  Defaults to dtype=float64 — GYMNET_FLAG_F64, the reference's own
    arithmetic (float64 state, the literal CartPoleEnv.cs:141-167 sequence) and its float64 observations (:166,185) — so an
    existing per-instance loop sees the reference's states to the last few ulps and its exact episode lengths, free-running.
    dtype=np.float32 selects the batched engine's float32 arithmetic (1e-5 per teacher-forced step)."""
    ENV = "CartPole-v1"
    DTYPE = np.float64


class PendulumEnv(GpuEnv):
    ENV = "Pendulum-v1"


class MountainCarEnv(GpuEnv):
    ENV = "MountainCar-v0"


class AcrobotEnv(GpuEnv):
    ENV = "Acrobot-v1"


__all__ = ["VectorEnv", "GroupVectorEnv", "DummyVecEnv", "BatchStep", "PendingStep", "GpuEnv", "CartPoleEnv", "PendulumEnv", "MountainCarEnv",
           "AcrobotEnv", "AlreadySteppingError", "NotSteppingError"]
