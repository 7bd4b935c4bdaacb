/*
 * gymnet_amd.h — C ABI of the MI355X-native batched classic-control environment engine.
 *
 * This is the drop-in boundary for ONE path of SciSharp/Gym.NET: the per-instance
 * Env.Step()/Reset() hot path of the classic-control environments, replaced by one HIP kernel
 * launch over a structure-of-arrays batch held in HBM (one environment instance per GPU lane).
 * Plain C types only: no torch types, no C++ types, no callbacks.  A C# host binds it with
 * [DllImport("gymnet_amd")]; a C++ or ctypes host binds it the same way.
 *
 * Every entry point cites the reference interface it replaces, as path:line relative to the
 * Gym.NET source tree (/root/reference in the build container):
 *   IEnv / Env            src/Gym/Envs/IEnv.cs:11-22, src/Gym/Envs/Env.cs:13-41
 *   IVecEnv / VecEnv      src/Gym/Envs/IVecEnv.cs:8-19, src/Gym/Envs/VecEnv.cs:12-93
 *   VecEnvWrapper         src/Gym/Envs/VecEnvWrapper.cs:9-30
 *   CartPoleEnv           src/Gym.Environments/Envs/Classic/CartPoleEnv.cs:24-67,137-198
 *   Space/Box/Discrete    src/Gym/Spaces/{Space.cs:5-18,Box.cs:25-96,Discrete.cs:11-44}
 *   Step                  src/Gym/Observations/Step.cs:7-20
 *   errors                src/Gym/Exceptions/{InvalidActionError,AlreadySteppingError,NotSteppingError}.cs
 *
 * Conventions
 *   - Every function returns a gymnet_status (0 = ok, negative = error) unless documented
 *     otherwise; nothing throws or aborts across the ABI.  The message for the most recent
 *     failure on the calling thread is gymnet_last_error().
 *   - "host" pointers are ordinary CPU memory owned by the caller; "device" pointers (names
 *     starting with d_) are HIP device memory on the handle's device.  The library never keeps a
 *     caller pointer after the call returns, except: gymnet_vecenv_step_async keeps `actions`
 *     until it returns (it copies), and *_device calls use d_ pointers until the work queued on
 *     the handle's stream has run (gymnet_vecenv_sync).
 *   - Layouts.  Inside the engine everything is structure-of-arrays: state[state_dim][stride],
 *     obs[obs_dim][stride], reward[n], done[n].  At the HOST boundary observations are
 *     row-major [num_envs, obs_dim] float32 (what an NDArray of shape (N, D) holds), reward is
 *     float32[num_envs], done is uint8[num_envs] (0 / 1).
 *   - Dtypes: Discrete action int32; Box action float32; observation float32 (the DECLARED dtype
 *     of CartPoleEnv.ObservationSpace, CartPoleEnv.cs:48); reward float32 (Step.cs:9); done uint8.
 *     A handle created with GYMNET_FLAG_F64 (CartPole) computes in the reference's ACTUAL arithmetic:
 *     float64 state, and float64 observations at the boundary — what CartPoleEnv.Step really returns
 *     (CartPoleEnv.cs:166,185: the float64 state NDArray itself).  Every `obs` / `state` buffer of
 *     such a handle holds doubles; the parameters are typed void* for that reason.
 *   - A handle is single-caller-at-a-time (like an Env instance, which has no re-entrancy guard);
 *     different handles may be used from different threads.  All work of a handle is ordered on
 *     one HIP stream.
 */
#ifndef GYMNET_AMD_H
#define GYMNET_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GYMNET_ABI_VERSION 6   /* 2: gymnet_config.d_ext_obs_alt, gymnet_device_view.{obs_buffer,d_obs_alt}, groups;
                                  3: gymnet_env_info.{traffic_bytes_per_step,state_row_in_obs}, per-element Box sampling, compact
                                     terminal observations, pinned host staging;
                                  4: GYMNET_FLAG_F64 (float64 CartPole: observation / state buffers typed by the handle,
                                     gymnet_device_view.state_dtype), GYMNET_FLAG_COMPACT_RECORDS_ONLY, gymnet_launch_policy +
                                     set / get, gymnet_vecenv_get_array / _set_array / _get_seed (checkpoint of every array);
                                  5: GYMNET_FLAG_F64 is a first-class mode — it combines with DONE_LIST / FINAL_OBS / DOUBLE_BUFFER /
                                     COMPACT_RECORDS_ONLY / d_ext_obs(_alt) and with groups; every parameter that carries observation
                                     values (gymnet_config.d_ext_obs*, terminal observations, group replicas and host batches) is
                                     typed void* = the handle's state scalar (binary layout of the calls unchanged).  A launch-policy
                                     value that would not take effect is rejected;
                                  6: ACTION STREAM v2 — no signature or struct changed, the VALUES every sampling entry point draws
                                     did (see "batched space sampling" below): one Philox4x32-10 call now serves the four consecutive
                                     global lanes of a group instead of one lane.  Reset draws, and therefore every result computed
                                     from caller-supplied actions, are bit-for-bit what ABI 5 produced */

typedef enum gymnet_status {
    GYMNET_OK = 0,
    GYMNET_ERR_INVALID_ARG = -1,     /* ArgumentException / ArgumentNullException (VecEnv.cs:49, CartPoleEnv.cs:57) */
    GYMNET_ERR_INVALID_ACTION = -2,  /* InvalidActionError (src/Gym/Exceptions/InvalidActionError.cs:7-10) */
    GYMNET_ERR_HIP = -3,             /* a HIP runtime call failed */
    GYMNET_ERR_OOM = -4,             /* device or host allocation failed */
    GYMNET_ERR_NO_DEVICE = -5,       /* no usable AMD GPU: the engine has NO CPU fallback */
    GYMNET_ERR_ALREADY_STEPPING = -6,/* AlreadySteppingError (src/Gym/Exceptions/AlreadySteppingError.cs:8-10) */
    GYMNET_ERR_NOT_STEPPING = -7,    /* NotSteppingError (src/Gym/Exceptions/NotSteppingError.cs:4-6) */
    GYMNET_ERR_UNSUPPORTED = -8,
    GYMNET_ERR_RCCL = -9             /* an RCCL call failed, or librccl could not be loaded (group gather mode RCCL) */
} gymnet_status;

typedef enum gymnet_env_id {
    GYMNET_ENV_CARTPOLE = 0,     /* CartPoleEnv.cs (the only classic env present in the reference) */
    GYMNET_ENV_PENDULUM = 1,     /* absent from the reference (README.md:69-76); upstream gym Pendulum-v1 */
    GYMNET_ENV_MOUNTAINCAR = 2,  /* absent from the reference; upstream gym MountainCar-v0 */
    GYMNET_ENV_ACROBOT = 3       /* absent from the reference; upstream gym Acrobot-v1 */
} gymnet_env_id;

/* gymnet_config.flags */
#define GYMNET_FLAG_AUTORESET        0x01u /* fuse the caller's `if (done) Reset()` (README.md:36-40) into the step kernel */
#define GYMNET_FLAG_VALIDATE_ACTIONS 0x02u /* reject actions outside Discrete(n) (Discrete.cs:38-40) before stepping, like
                                              LunarLanderEnv.cs:604-607; default mirrors Release-build CartPole (CartPoleEnv.cs:139,146) */
#define GYMNET_FLAG_DONE_LIST        0x04u /* emit the compacted list of lanes that finished in the last step */
#define GYMNET_FLAG_EPISODE_STATS    0x08u /* per-lane episode return / length bookkeeping (BasePlaySession.cs:58-69) */
#define GYMNET_FLAG_FINAL_OBS        0x10u /* with AUTORESET: keep the terminal observation of lanes that finished */
#define GYMNET_FLAG_F64              0x40u /* ABI 4, CartPole only: the reference's own arithmetic — float64 structure-of-arrays state, the literal
                                              operation sequence of CartPoleEnv.cs:141-167 in binary64 (float32-valued constants widened at use),
                                              reset draws with 53 random bits, float64 observations at the boundary.  Reproduces the reference's
                                              episode lengths free-running (the float32 engine guarantees 1e-5 per teacher-forced step only).
                                              73 B per env-step instead of 41.  Since ABI 5 it combines with every other flag, with d_ext_obs(_alt)
                                              (buffers of doubles) and with groups: the same kernels, instantiated for double (step_kernels.hpp) */
#define GYMNET_FLAG_COMPACT_RECORDS_ONLY 0x80u /* ABI 4, with DONE_LIST: the step kernel writes the finished lanes' episode records / terminal
                                              observations ONLY as compact records (gymnet_vecenv_done_records); the dense per-lane views
                                              (gymnet_vecenv_episode_stats / _final_obs, gymnet_device_view.d_finished_*) are then brought up to
                                              date on demand, for the most recent step only.  Default (flag clear): the step kernel keeps the
                                              dense views current itself, whatever the caller reads or skips */
#define GYMNET_FLAG_RESIDENT         0x100u /* ABI 5, num_envs <= 64 (the single-instance usage shape, README.md:32-52 / Env.cs:13-41): the host-boundary
                                              step / reset calls (gymnet_vecenv_step, _step_broadcast, _reset, _reset_where(NULL)) are served by a RESIDENT
                                              single-wave kernel that polls a mailbox in page-locked, device-mapped host memory: no kernel launch and no
                                              stream synchronize per step (two PCIe crossings instead of ~25 us).  Results are bit-identical to the
                                              launch path.  The kernel leaves by itself after ~50 ms without a command and is restarted on demand; every
                                              other entry point first tells it to leave.  While it is resident it occupies the handle's stream: a
                                              device-wide synchronize issued elsewhere in the process waits for it (up to the idle timeout).  Not with
                                              DONE_LIST / FINAL_OBS / DOUBLE_BUFFER / d_ext_obs / a caller's stream (GYMNET_ERR_UNSUPPORTED) */
#define GYMNET_FLAG_DOUBLE_BUFFER    0x20u /* two observation buffers, written alternately: the step launched after buffer A was
                                              written reads A and writes B, so a consumer (an all-gather of A over xGMI, a policy
                                              reading A) may still be using A while the next step runs.  For envs whose observation
                                              IS the state (CartPole, MountainCar) the state ping-pongs with it.  Which buffer holds
                                              the latest observation: gymnet_device_view.obs_buffer / d_obs */

typedef struct gymnet_vecenv gymnet_vecenv;   /* opaque handle: one batch ("VectorEnv") on one GPU */

typedef struct gymnet_config {
    uint32_t struct_size;       /* = sizeof(gymnet_config) */
    int32_t  env_id;            /* gymnet_env_id */
    int64_t  num_envs;          /* lanes owned by this handle (this rank's shard) — VecEnv.NumberOfEnvironments */
    int64_t  lane_offset;       /* global id of this handle's lane 0; reset draws are keyed by GLOBAL lane id, so a
                                   batch sharded over G handles/GPUs gives the same results as one handle */
    int32_t  device;            /* HIP device ordinal */
    uint32_t flags;             /* GYMNET_FLAG_* */
    uint64_t seed;              /* Env.Seed(int) (CartPoleEnv.cs:196-198): Philox key */
    void    *stream;            /* hipStream_t to order all work on; NULL = the library creates its own */
    void    *d_ext_obs;         /* optional device buffer [obs_dim][ext_obs_stride] (float32; float64 with GYMNET_FLAG_F64) to keep
                                   observations in (e.g. this
                                   rank's slice of an all-gather buffer).  It is LIVE STATE STORAGE, not an output copy: for
                                   envs whose observation IS the state (CartPole, MountainCar) every row, and for the others
                                   the rows listed in gymnet_env_info.state_row_in_obs (Pendulum obs[2]; Acrobot obs[4], obs[5]),
                                   are read back by the next step.  A consumer must not modify them in place (normalise /
                                   clip into its own buffer).  The same holds for d_ext_obs_alt.  NULL = library allocates */
    int64_t  ext_obs_stride;    /* elements between component arrays of d_ext_obs (>= num_envs) */
    int32_t  max_episode_steps; /* EXTENSION (the reference has no time limit): >0 truncates episodes;
                                   requires GYMNET_FLAG_EPISODE_STATS; done byte gets bit 1 (value 2) for truncation */
    int32_t  reserved;
    void    *d_ext_obs_alt;     /* with GYMNET_FLAG_DOUBLE_BUFFER and d_ext_obs: the caller's SECOND observation buffer (same
                                   stride), e.g. this rank's slice of a second all-gather buffer.  NULL = library allocates */
} gymnet_config;

typedef struct gymnet_env_info {
    uint32_t struct_size;
    int32_t  env_id;
    char     name[32];               /* "CartPole-v1", ... */
    int32_t  state_dim;
    int32_t  obs_dim;
    int32_t  obs_aliases_state;      /* 1: the observation arrays ARE the state arrays */
    int32_t  action_is_box;          /* 0: Discrete(action_n) int32; 1: Box(action_low, action_high, (1,)) float32 */
    int32_t  action_n;
    float    action_low, action_high;
    float    obs_low[8], obs_high[8];/* ObservationSpace bounds (CartPoleEnv.cs:46-48) */
    float    reward_low, reward_high;
    int32_t  algorithmic_bytes_per_step; /* bytes one env-step must move (CartPole: 41) */
    int32_t  traffic_bytes_per_step;     /* ABI 3: bytes one env-step really moves: less than the algorithmic figure where a state
                                            component the observation repeats verbatim is stored once (Pendulum 33 of 37,
                                            Acrobot 57 of 65) */
    int32_t  state_row_in_obs[8];        /* ABI 3: state component k is held in row state_row_in_obs[k] of the OBSERVATION array
                                            (Pendulum theta_dot = obs[2]; Acrobot dtheta1, dtheta2 = obs[4], obs[5]); -1 = in its
                                            own row of the state array.  Only matters to readers of gymnet_device_view.d_state;
                                            gymnet_vecenv_get_state / _set_state assemble the full [state_dim][num_envs] array */
} gymnet_env_info;

/* Device-side view of a handle: zero-copy access for a GPU-resident policy / trainer. */
typedef struct gymnet_device_view {
    uint32_t struct_size;
    int32_t  state_dim, obs_dim, obs_aliases_state;
    int64_t  num_envs, state_stride, obs_stride;
    void    *d_state;          /* [state_dim][state_stride] of state_dtype; rows listed in gymnet_env_info.state_row_in_obs are NOT
                                  kept here (they are rows of d_obs) */
    void    *d_obs;            /* [obs_dim][obs_stride] of state_dtype (== d_state when obs_aliases_state) */
    float   *d_reward;         /* [num_envs] */
    uint8_t *d_done;           /* [num_envs] */
    int32_t *d_steps_beyond_done; /* CartPole without AUTORESET: CartPoleEnv.cs:41 per lane; else NULL */
    void    *d_final_obs;      /* [obs_dim][num_envs] of state_dtype, FINAL_OBS only */
    int32_t *d_done_list;      /* [num_envs] compact list, valid after gymnet_vecenv_done_lanes(_device); DONE_LIST only */
    float   *d_episode_return; int32_t *d_episode_length;     /* running, EPISODE_STATS only */
    float   *d_finished_return; int32_t *d_finished_length;   /* last finished episode per lane */
    void    *stream;           /* hipStream_t all of the handle's work is ordered on */
    int32_t  obs_buffer;       /* GYMNET_FLAG_DOUBLE_BUFFER: index (0 / 1) of the buffer d_obs points at = the latest observation */
    int32_t  state_dtype;      /* ABI 4: gymnet_dtype of d_state / d_obs — GYMNET_DTYPE_F32, or GYMNET_DTYPE_F64 for a GYMNET_FLAG_F64 handle */
    void    *d_obs_alt;        /* GYMNET_FLAG_DOUBLE_BUFFER: the other buffer = what the NEXT step will write; else NULL */
} gymnet_device_view;

typedef enum gymnet_dtype { GYMNET_DTYPE_F32 = 0, GYMNET_DTYPE_F64 = 1 } gymnet_dtype;

/* ABI 4.  The step kernel's launch configuration.  The library chooses one at create from the batch size and the
 * buffers' alignment; gymnet_vecenv_set_launch_policy overrides fields (every field: -1 = leave as it is).  All configurations
 * of an env compute bit-identical results — this is a performance knob for probes, A/B timing and tests that pin a kernel form,
 * set through the ABI so that a host process's ENVIRONMENT never changes which kernel the library runs. */
typedef struct gymnet_launch_policy {
    uint32_t struct_size;          /* = sizeof(gymnet_launch_policy) */
    int32_t  vec;                  /* lanes per thread on wide accesses: 1, 4 (dwordx4 rows; not Acrobot), 2 (Acrobot packed-FP32 form; F64 handles) */
    int32_t  block;                /* threads per workgroup: 64 / 128 / 256 */
    int32_t  nt;                   /* non-temporal stream mask: 0 none, 12 action + reward / done, 15 every stream */
    int32_t  sequential_lanes;     /* multi-lane kernels (lean variant): Acrobot with vec 1 — lanes per thread, 1 (one-shot kernel) .. 5;
                                      Acrobot / F64 handles with vec 2 — lane PAIRS per thread, 1 .. 4; num_envs a multiple of
                                      2 * sequential_lanes * 256, except F64 handles with auto-reset (any batch size; their
                                      multi-pair kernel draws the fused reset once per thread-group of pairs whatever reset_form
                                      says).  A value > 1 the launcher would not resolve to -> GYMNET_ERR_INVALID_ARG */
    int32_t  reset_form;           /* fused auto-reset: 0 per-thread drain loop, 1 wave-compacted (wide kernels of the envs whose
                                      observation is the state: CartPole in both state scalars — F64: two lanes per reset —,
                                      MountainCar); also selects the per-step reset of the fused rollout */
    int32_t  lds_pipe;             /* Acrobot: 1 = producer / consumer form of the multi-lane kernel (needs num_envs % 512 == 0) */
    int32_t  occupancy_lds_bytes;  /* unused dynamic LDS per workgroup, for the one purpose of capping occupancy in probes */
    int32_t  graph;                /* gymnet_vecenv_rollout_device: 0 eager launches, 1 hipGraph replay, -2 back to "by batch size" */
} gymnet_launch_policy;

typedef struct gymnet_counters {
    uint32_t struct_size;
    uint32_t reserved;
    uint64_t tick;               /* engine tick: number of step/reset launches since seed (Philox counter word) */
    uint64_t lane_steps;         /* total env-steps executed */
    uint64_t stepped_after_done; /* lane-steps taken on an already-done lane: the reference's console warning
                                    (CartPoleEnv.cs:176-179), counted instead of printed */
    int64_t  last_done_count;    /* lanes that finished in the last step (DONE_LIST), else -1 */
} gymnet_counters;

/* ---- library-level ------------------------------------------------------------------------ */
int         gymnet_abi_version(void);
const char *gymnet_status_string(int status);
const char *gymnet_last_error(void);                 /* thread-local message of the last failure */
int         gymnet_device_count(int *count);         /* GYMNET_ERR_NO_DEVICE (count = 0) without a GPU */
/* Space descriptors an env's ctor builds (CartPoleEnv.cs:43-52): ActionSpace, ObservationSpace bounds. */
int         gymnet_env_describe(int env_id, gymnet_env_info *out);

/* ---- lifecycle: new VecEnv(...) / Close() / Dispose() ------------------------------------- */
/* VecEnv ctor (VecEnv.cs:13-18) + N x CartPoleEnv ctor (CartPoleEnv.cs:43-52).  State is undefined until reset. */
int gymnet_vecenv_create(const gymnet_config *cfg, gymnet_vecenv **out);
/* VecEnv.Close() (VecEnvWrapper.cs:26-30) / Env.Dispose() (Env.cs:38-40). */
int gymnet_vecenv_destroy(gymnet_vecenv *h);
/* VecEnv.Seed(int) (VecEnv.cs:44-46) -> Env.Seed (CartPoleEnv.cs:196-198).  DEVIATION: the reference hands every
 * env the SAME seed (identical reset streams); here lane i draws from Philox(key = seed, counter = (global lane, tick)).
 * Also rewinds the engine tick to 0. */
int gymnet_vecenv_seed(gymnet_vecenv *h, uint64_t seed);
/* VecEnv.Seed(int[]) (VecEnv.cs:48-53): one seed per lane; count != num_envs -> GYMNET_ERR_INVALID_ARG
 * (the reference throws ArgumentException).  Lane i then draws from Philox(key = seeds[i], ...). */
int gymnet_vecenv_seed_lanes(gymnet_vecenv *h, const uint64_t *seeds, int64_t count);

/* ---- host-boundary path (what an NDArray-based caller uses) --------------------------------- */
/* VecEnv.Reset() (VecEnvWrapper.cs:18-20) -> N x CartPoleEnv.Reset() (CartPoleEnv.cs:63-67): steps_beyond_done = -1,
 * state ~ U(-0.05,0.05)^4.  obs_out: host [num_envs, obs_dim] or NULL — float32, or float64 for a GYMNET_FLAG_F64 handle (here
 * and in every call below that takes obs_out). */
int gymnet_vecenv_reset(gymnet_vecenv *h, void *obs_out);
/* The caller's `if (done) Reset()` (README.md:36-40), batched: resets exactly the lanes with mask[i] != 0;
 * mask == NULL resets the lanes whose last returned done flag is set.  obs_out as above (all lanes).
 * With GYMNET_FLAG_AUTORESET the step itself already re-drew every finished lane, so mask == NULL is a NO-OP that only
 * returns the current observations (it used to draw those lanes a second time); an explicit mask still resets. */
int gymnet_vecenv_reset_where(gymnet_vecenv *h, const uint8_t *mask, void *obs_out);
/* EXTENSION of IVecEnv.Step (per-lane actions): N x CartPoleEnv.Step (CartPoleEnv.cs:137-186).
 * actions: host int32[num_envs] (Discrete) or float32[num_envs] (Box).  Outputs may each be NULL.  Blocks until
 * the outputs are written. */
int gymnet_vecenv_step(gymnet_vecenv *h, const void *actions, void *obs_out, float *reward_out, uint8_t *done_out);
/* IVecEnv.Step(int action) (IVecEnv.cs:15, VecEnvWrapper.cs:22-24): ONE scalar action broadcast to all lanes. */
int gymnet_vecenv_step_broadcast(gymnet_vecenv *h, int32_t action, void *obs_out, float *reward_out, uint8_t *done_out);
/* VecEnv.StepAsync (VecEnv.cs:63-65) / Env.StepAsync (Env.cs:23-25): queue the step, return immediately.
 * A second call before gymnet_vecenv_step_wait -> GYMNET_ERR_ALREADY_STEPPING. */
int gymnet_vecenv_step_async(gymnet_vecenv *h, const void *actions);
/* step_wait(): block until the queued step finished and copy its results out.  Without a pending
 * gymnet_vecenv_step_async -> GYMNET_ERR_NOT_STEPPING. */
int gymnet_vecenv_step_wait(gymnet_vecenv *h, void *obs_out, float *reward_out, uint8_t *done_out);
/* ABI 3.  Library-owned HOST buffers for this path, for a caller that can keep its NDArrays over unmanaged memory: actions
 * (int32 / float32 [num_envs]), obs (float32 [num_envs, obs_dim]), reward (float32 [num_envs]), done (uint8 [num_envs]) —
 * page-locked and mapped into the device, valid until gymnet_vecenv_destroy, allocated on the first call.  When the pointers
 * handed to gymnet_vecenv_step / _reset / _reset_where / _step_wait / _read ARE these buffers, the call runs without staging:
 * the actions are DMA'd straight out of the pinned buffer and ONE kernel writes observations (row-major), rewards and done flags
 * across PCIe into the others (no device-side pack buffer, no per-array memcpy).  Ordinary caller-owned memory keeps working
 * (staged copies).  Any out pointer may be NULL. */
int gymnet_vecenv_host_buffers(gymnet_vecenv *h, void **actions, void **obs, float **reward, uint8_t **done);
/* Copy the results of the most recent step/reset again (Step record, Step.cs:8-10). */
int gymnet_vecenv_read(gymnet_vecenv *h, void *obs_out, float *reward_out, uint8_t *done_out);

/* ---- device-resident path (no PCIe on the hot path; everything stream-ordered, non-blocking) -- */
int gymnet_vecenv_reset_device(gymnet_vecenv *h);
int gymnet_vecenv_reset_where_device(gymnet_vecenv *h, const uint8_t *d_mask);   /* NULL = own done flags */
/* One vector step = ONE kernel launch.  d_actions: device int32[num_envs] / float32[num_envs]. */
int gymnet_vecenv_step_device(gymnet_vecenv *h, const void *d_actions);
/* `steps` consecutive vector steps, one kernel launch each, replayed from a cached hipGraph: step t reads
 * d_actions + (t % ring) * action_stride elements.  (The caller's hot loop, README.md:34-47, without host hops.) */
int gymnet_vecenv_rollout_device(gymnet_vecenv *h, const void *d_actions, int64_t steps,
                                 int64_t action_stride, int64_t ring);
/* Device-side rollout buffers (the example's replay memory, batched: examples/ReinforcementLearning/
 * ReinforcementLearning/MemoryTypes/ReplayMemory.cs:25-67).  Any pointer may be NULL = do not record that stream. */
typedef struct gymnet_rollout_buffers {
    void    *d_obs;     /* [steps][obs_dim][num_envs]  observation AFTER step t (after auto-reset, like the step API); float32 —
                           float64 for a GYMNET_FLAG_F64 handle */
    float   *d_reward;  /* [steps][num_envs] */
    uint8_t *d_done;    /* [steps][num_envs] */
} gymnet_rollout_buffers;
/* The same `steps` vector steps as gymnet_vecenv_rollout_device — bit-identical state, reward, done — fused into ONE
 * kernel launch: every lane keeps its state in registers across the steps, so per env-step only the action is read and
 * (optionally, rec != NULL) the recorded streams are written.  For open-loop / pre-generated action sequences only: no
 * policy can look at step t's observation before step t+1.  (Since ABI 5 also on bookkeeping handles: see
 * gymnet_vecenv_rollout_fused_ex_device below, of which this is the ring-actions, no-episode-records form.) */
int gymnet_vecenv_rollout_fused_device(gymnet_vecenv *h, const void *d_actions, int64_t steps, int64_t action_stride,
                                       int64_t ring, const gymnet_rollout_buffers *rec);
/* ABI 5.  The fused rollout with what its CONSUMER needs (examples/ReinforcementLearning/ReinforcementLearning/PlaySessions/
 * BasePlaySession.cs:58-69 accumulates the episode reward and keeps the best episodes, MemoryTypes/ReplayMemory.cs:53-67 stores
 * them; TrainingPlaySession.cs:46-52 draws an epsilon-greedy action per step) — still ONE kernel launch for `steps` vector steps:
 *   - on a bookkeeping handle (EPISODE_STATS / DONE_LIST / FINAL_OBS / per-lane seeds; gymnet_vecenv_rollout_fused_device accepts
 *     those too now) the running episode return / length live in registers for the whole rollout, max_episode_steps truncates,
 *     per-lane seeds key the reset draws, the dense last-finished-episode views stay current, and the done list / records of
 *     "the most recent step" describe the rollout's last step: the handle ends in exactly the state `steps` single steps leave;
 *   - every episode that ends during the rollout leaves a compact record (step index t, lane, return, length) in the caller's
 *     arrays (wave ballot + one atomic per wave inside the kernel, gathered afterwards; unordered);
 *   - the ACTIONS can be drawn inside the kernel — GYMNET_ACTIONS_SAMPLE: ActionSpace.Sample() per lane and step, exactly the
 *     values gymnet_vecenv_sample_actions_device(seed = action_seed, tick = action_tick0 + t) would write; GYMNET_ACTIONS_
 *     EPSILON_GREEDY: gymnet_vecenv_compose_actions_device over the ring as the policy's actions — so a random rollout reads no
 *     action ring at all (0 B instead of 4 B per env-step), and d_rec_actions records what was taken.
 * Results are bit-identical to `steps` x (sample / compose, then gymnet_vecenv_step_device).  Stream-ordered, non-blocking. */
/* gymnet_rollout_spec.record_flags.  By default a rollout that keeps episode records loses none below ep_capacity.  NO_OVERFLOW selects the
 * kernel variant without the overflow path (8 % faster with records): episodes of lanes that finish very UNEVENLY — a few waves producing
 * most of them — can then be dropped below ep_capacity (a per-shard limit of 2 * ceil(ep_capacity / 256) + 64 records; d_ep_count[1] > [0]
 * says so).  Evenly finishing batches, e.g. random rollouts, never notice the difference. */
#define GYMNET_RECORDS_NO_OVERFLOW 1
typedef enum gymnet_action_source { GYMNET_ACTIONS_RING = 0, GYMNET_ACTIONS_SAMPLE = 1, GYMNET_ACTIONS_EPSILON_GREEDY = 2 } gymnet_action_source;
typedef struct gymnet_rollout_spec {
    uint32_t struct_size;        /* = sizeof(gymnet_rollout_spec) */
    int32_t  action_source;      /* gymnet_action_source */
    const void *d_actions;       /* RING: the actions; EPSILON_GREEDY: the policy's actions; SAMPLE: ignored (may be NULL) */
    int64_t  steps;
    int64_t  action_stride;      /* step t reads d_actions + (t % ring) * action_stride elements */
    int64_t  ring;
    uint64_t action_seed;        /* SAMPLE / EPSILON_GREEDY: Philox action stream key ... */
    uint64_t action_tick0;       /* ... and tick of step 0 (step t draws with tick action_tick0 + t) */
    float    epsilon;            /* EPSILON_GREEDY: exploration probability, [0, 1] */
    int32_t  record_flags;       /* episode records: 0, or GYMNET_RECORDS_NO_OVERFLOW (ABI 5 called this field `reserved`: 0 = the default) */
    /* dense per-step recording (the members of gymnet_rollout_buffers); any pointer NULL = not recorded */
    void    *d_rec_obs;          /* [steps][obs_dim][num_envs] observation AFTER step t; float32 — float64 for a GYMNET_FLAG_F64 handle */
    float   *d_rec_reward;       /* [steps][num_envs] */
    uint8_t *d_rec_done;         /* [steps][num_envs] */
    void    *d_rec_actions;      /* [steps][num_envs] the actions TAKEN (int32 / float32), or NULL */
    /* compact records of the episodes that ended during the rollout (bookkeeping handles); all NULL = none wanted */
    int32_t *d_ep_step;          /* [ep_capacity] step index t inside this rollout */
    int32_t *d_ep_lane;          /* [ep_capacity] lane */
    float   *d_ep_return;        /* [ep_capacity] episode return  (EPISODE_STATS) */
    int32_t *d_ep_length;        /* [ep_capacity] episode length  (EPISODE_STATS) */
    int64_t  ep_capacity;        /* records the arrays hold; a random-action CartPole rollout ends ~0.045 x num_envs episodes per step */
    uint32_t *d_ep_count;        /* [2]: [0] records written, [1] episodes that ended.  [1] > [0] exactly when more episodes ended than
                                    ep_capacity: then ep_capacity records are kept (which ones is unspecified) and the rest are only
                                    counted.  (Inside the kernel the records live in 256 per-shard segments — a wave appends to segment
                                    (wave index mod 256) with one atomic per flush — of 2 * ceil(ep_capacity / 256) + 64 records each;
                                    what a shard cannot hold, e.g. when a few waves produce most of the episodes, spills to one shared
                                    overflow segment of ep_capacity records, so an uneven batch loses nothing — unless record_flags asks
                                    for GYMNET_RECORDS_NO_OVERFLOW, round 5's behaviour.) */
} gymnet_rollout_spec;
int gymnet_vecenv_rollout_fused_ex_device(gymnet_vecenv *h, const gymnet_rollout_spec *spec);
/* Pack the SoA observations into row-major [num_envs, obs_dim] on the device (the NDArray layout; float32, or float64 for a
 * GYMNET_FLAG_F64 handle). */
int gymnet_vecenv_pack_obs_device(gymnet_vecenv *h, void *d_obs_rowmajor);
int gymnet_vecenv_sync(gymnet_vecenv *h);
int gymnet_vecenv_device_view(gymnet_vecenv *h, gymnet_device_view *out);
/* The launch configuration the handle chose for its step kernel (DESIGN.md §4 launch policy): lanes per thread on wide
 * accesses (1 / 2 / 4), threads per workgroup, non-temporal stream mask (0 / 12 / 15), and lanes per thread of the multi-lane
 * kernel that loads all of a thread's lanes first and then computes / stores them one after another (Acrobot; 1 = the
 * one-shot kernel).  Any out pointer may be NULL. */
int gymnet_vecenv_launch_policy(gymnet_vecenv *h, int32_t *vec, int32_t *block, int32_t *nt, int32_t *sequential_lanes);
/* ABI 4.  Override / read back the whole launch configuration (struct above).  A value the handle cannot run (dwordx4 lanes on an
 * unaligned external buffer, Acrobot's forms on another env, ...) -> GYMNET_ERR_INVALID_ARG and nothing changes.  Drops the
 * handle's captured graphs.  Synchronizes the handle's stream. */
int gymnet_vecenv_set_launch_policy(gymnet_vecenv *h, const gymnet_launch_policy *policy);
int gymnet_vecenv_get_launch_policy(gymnet_vecenv *h, gymnet_launch_policy *out);
/* ABI 3.  The kernel instantiation the handle's NEXT step launch runs, as text — e.g. "step_kernel<CartPole,4,true,false,15,1>"
 * (env, lanes per thread, AUTORESET, EXTRAS, non-temporal mask, reset form) or "step_kernel_pipe<Acrobot,4,true,15>".  It is
 * printed by the same function the launcher dispatches on, so a profile, a bench line or a parity test can name what ran
 * without copying the launch policy.  Diagnostic only: the spelling is not a stable interface. */
int gymnet_vecenv_kernel_name(gymnet_vecenv *h, char *buf, int32_t capacity);

/* ---- state access: teacher-forced parity tests, checkpoint / resume -------------------------- */
/* host [state_dim][num_envs] (structure-of-arrays), float32 — float64 for a GYMNET_FLAG_F64 handle.
 * CartPole: x, x_dot, theta, theta_dot (CartPoleEnv.cs:141-144). */
int gymnet_vecenv_get_state(gymnet_vecenv *h, void *state_soa);
int gymnet_vecenv_set_state(gymnet_vecenv *h, const void *state_soa);
/* steps_beyond_done per lane (CartPoleEnv.cs:41); only for CartPole without AUTORESET, else GYMNET_ERR_UNSUPPORTED. */
int gymnet_vecenv_get_steps_beyond_done(gymnet_vecenv *h, int32_t *out);
int gymnet_vecenv_set_steps_beyond_done(gymnet_vecenv *h, const int32_t *in);
int gymnet_vecenv_get_tick(gymnet_vecenv *h, uint64_t *tick);
int gymnet_vecenv_set_tick(gymnet_vecenv *h, uint64_t tick);
int gymnet_vecenv_counters(gymnet_vecenv *h, gymnet_counters *out);
/* ABI 4.  Every per-lane array a handle keeps, by id — together with the state, the tick and the seed this is a COMPLETE
 * checkpoint of any configuration (get_state / set_state double as checkpoint / resume): running episode return /
 * length (which drive the max_episode_steps truncation), the reward / done flags of the last step (which reset_where(NULL)
 * consumes), steps_beyond_done, the dense finished-episode views, the per-lane Philox keys of VecEnv.Seed(int[]).
 * `bytes` must equal the array's size (num_envs x element size; FINAL_OBS: obs_dim x num_envs x 4 — x 8 for a float64 handle —, structure-of-arrays).
 * An array the handle's configuration does not have -> GYMNET_ERR_UNSUPPORTED.  Both calls block.
 * set(LANE_SEEDS) installs the keys WITHOUT rewinding the tick (gymnet_vecenv_seed_lanes rewinds it); set(DONE) also forgets the
 * compacted done list of the step before (it described other flags). */
typedef enum gymnet_array_id {
    GYMNET_ARRAY_REWARD = 0,             /* float32 [num_envs] */
    GYMNET_ARRAY_DONE = 1,               /* uint8   [num_envs]  (bit 0 terminated, bit 1 truncated) */
    GYMNET_ARRAY_STEPS_BEYOND_DONE = 2,  /* int32   [num_envs]  CartPole without AUTORESET (CartPoleEnv.cs:41) */
    GYMNET_ARRAY_EPISODE_RETURN = 3,     /* float32 [num_envs]  EPISODE_STATS: running */
    GYMNET_ARRAY_EPISODE_LENGTH = 4,     /* int32   [num_envs] */
    GYMNET_ARRAY_FINISHED_RETURN = 5,    /* float32 [num_envs]  EPISODE_STATS: last finished episode per lane */
    GYMNET_ARRAY_FINISHED_LENGTH = 6,    /* int32   [num_envs] */
    GYMNET_ARRAY_FINAL_OBS = 7,          /* float32 (float64: F64 handle) [obs_dim][num_envs]  FINAL_OBS */
    GYMNET_ARRAY_LANE_SEEDS = 8          /* uint64  [num_envs]  after gymnet_vecenv_seed_lanes with distinct seeds */
} gymnet_array_id;
int gymnet_vecenv_get_array(gymnet_vecenv *h, int32_t which, void *out, int64_t bytes);
int gymnet_vecenv_set_array(gymnet_vecenv *h, int32_t which, const void *in, int64_t bytes);
/* The Philox key in use (Env.Seed(int), CartPoleEnv.cs:196-198) and whether per-lane keys are active.  Either out may be NULL. */
int gymnet_vecenv_get_seed(gymnet_vecenv *h, uint64_t *seed, int32_t *per_lane);

/* ---- episode bookkeeping (the step AFTER the path: BasePlaySession.cs:58-69) ------------------ */
/* Lanes that finished in the most recent step (unordered). Needs GYMNET_FLAG_DONE_LIST. */
int gymnet_vecenv_done_lanes(gymnet_vecenv *h, int32_t *lanes_out, int64_t capacity, int64_t *count);
/* Device-side form: writes the compact list to d_lanes_out (capacity num_envs int32; NULL = the handle's own buffer,
 * gymnet_device_view.d_done_list) and the count to *d_count_out.  Stream-ordered, does not block.  (Inside the step
 * kernel the list is built per wave with ballot + one atomic into one of 256 shard counters; this call gathers the
 * shards.) */
int gymnet_vecenv_done_lanes_device(gymnet_vecenv *h, int32_t *d_lanes_out, uint32_t *d_count_out);
/* ABI 3.  The RECORDS of the lanes that finished in the most recent step, compacted (unordered; all arrays in the same order):
 * lane ids, with EPISODE_STATS the finished episode's return and length, with FINAL_OBS its terminal observation as rows
 * [count][obs_dim] (float32; float64 for a GYMNET_FLAG_F64 handle — here and in gymnet_vecenv_final_obs) — what a trainer consumes per step (BasePlaySession.cs:58-69: accumulate reward, count steps per episode)
 * without shipping num_envs flags to the host.  Inside the step kernel every finished lane's record is written AT ITS POSITION
 * in the (sharded) done list, so a wave's ~11 finished lanes write a few contiguous cache lines instead of one scattered line
 * per lane and array.  Any out pointer may be NULL; at most `capacity` records are copied, *count is the true number.
 * Needs GYMNET_FLAG_DONE_LIST. */
int gymnet_vecenv_done_records(gymnet_vecenv *h, int32_t *lanes_out, float *return_out, int32_t *length_out, void *final_obs_out,
                               int64_t capacity, int64_t *count);
/* Device-side form (stream-ordered, does not block): caller-owned device arrays of `capacity` records each (d_final_obs:
 * [capacity][obs_dim] row-major); *d_count receives the true number.  Any array may be NULL. */
int gymnet_vecenv_done_records_device(gymnet_vecenv *h, int32_t *d_lanes, float *d_return, int32_t *d_length, void *d_final_obs,
                                      int64_t capacity, uint32_t *d_count);
/* DENSE views, one row per lane.  Last finished episode's return and length per lane (0 length = none finished yet); needs
 * EPISODE_STATS.  Terminal observations, host [num_envs, obs_dim], rows of lanes that never finished are 0; needs FINAL_OBS.
 * The step kernel maintains these arrays itself (scattered stores), so they are current after ANY sequence of steps, rollouts or
 * graph replays, read or not.  Only with GYMNET_FLAG_COMPACT_RECORDS_ONLY (an opt-in beside DONE_LIST) does the step write the
 * compact records alone; each call of these getters then first applies the records of the MOST RECENT step to the dense arrays,
 * and records of steps the caller did not read are not in the dense view (use gymnet_vecenv_done_records every step). */
int gymnet_vecenv_episode_stats(gymnet_vecenv *h, float *finished_return, int32_t *finished_length);
int gymnet_vecenv_final_obs(gymnet_vecenv *h, void *final_obs_out);

/* ---- batched space sampling (the step BEFORE the path: ActionSpace.Sample(), TrainingPlaySession.cs:46-49) -- */
/* All sampling below draws from the ACTION stream (version 2, ABI 6).  A sampled action consumes one 32-bit word and a
 * Philox4x32-10 call yields four, so the four consecutive GLOBAL lanes of a group share one call:
 *     word A of global lane L at tick t = word (L & 3) of Philox(key = seed ^ 0x9E3779B97F4A7C15, counter = (L >> 2, t))
 *     word B of global lane L at tick t = word (L & 3) of Philox(key = seed ^ 0xD6E8FEB86659FD93, counter = (L >> 2, t))
 * with L = lane_offset + i.  A is the ActionSpace.Sample() word; B is drawn only by the consumers that need a second word (the
 * epsilon-greedy coin, the second uniform of Box.cs:82's normal).  Neither key is the reset draws' (key = seed), so
 * ActionSpace.Sample() called with an env's own (seed, tick) never replays the words of that env's reset draws; values depend on
 * the GLOBAL lane only, so a sharded batch samples what the whole batch would (any lane_offset, aligned to a group or not).
 * (Version 1, ABI <= 5: a whole call per lane, counter (L, t), words 0 and 1.  The reference's own stream is NumSharp's
 * un-vendored generator — CartPoleEnv.cs:49, Discrete.cs:17-28 — and is pinned by nothing)
 * Discrete.Sample() without mask (Discrete.cs:17-28): start + randint(0, n).  Element i = start + hi32(word A * n). */
int gymnet_sample_discrete_device(int device, void *stream, int32_t *d_out, int64_t count, int32_t n, int32_t start,
                                  uint64_t seed, uint64_t lane_offset, uint64_t tick);
/* Discrete.Sample(mask) (Discrete.cs:18-26): valid = {k : mask[k] == 1}; none -> start, else start + valid[hi32(word A * |valid|)].
 * d_mask: device uint8, one row of n bytes per element (mask_stride = n) or ONE row shared by all elements (mask_stride = 0). */
int gymnet_sample_discrete_masked_device(int device, void *stream, int32_t *d_out, int64_t count, int32_t n, int32_t start,
                                         const uint8_t *d_mask, int64_t mask_stride, uint64_t seed, uint64_t lane_offset, uint64_t tick);
/* Box.Sample() (Box.cs:69-90), the reference's four regimes per element: bounded -> uniform(low, high);
 * low only -> low + Exp(1); high only -> high + Exp(1) (sic, Box.cs:84); unbounded -> Normal(0.5, 1) (sic, Box.cs:82).
 * low = -INFINITY / high = +INFINITY select the regime. */
int gymnet_sample_box_device(int device, void *stream, float *d_out, int64_t count, float low, float high,
                             uint64_t seed, uint64_t lane_offset, uint64_t tick);
/* ABI 3.  Box.Sample() for a Box built from Low / High ARRAYS (Box.cs:25-51: `new Box(NDArray low, NDArray high)`), e.g. an
 * observation space whose velocity components are unbounded: `count` samples of `dim` elements each, row-major [count][dim];
 * d_low / d_high are device arrays of `dim` floats and every element picks ITS OWN regime from its own bounds, like the
 * reference's boolean masks (Box.cs:74-85).  gymnet_sample_box_device above is the scalar-bounds form (one pair for every
 * element: the four envs' action spaces); for dim = 1 both draw the same values. */
int gymnet_sample_box_elementwise_device(int device, void *stream, float *d_out, int64_t count, int32_t dim, const float *d_low,
                                         const float *d_high, uint64_t seed, uint64_t lane_offset, uint64_t tick);
/* ActionSpace.Sample() for every lane of a handle into d_actions (int32 / float32 [num_envs]). */
int gymnet_vecenv_sample_actions_device(gymnet_vecenv *h, void *d_actions, uint64_t seed, uint64_t tick);
int gymnet_vecenv_sample_actions(gymnet_vecenv *h, void *actions_out, uint64_t seed, uint64_t tick);
/* ActionSpace.Sample(mask) for every lane of a handle (Discrete action spaces only; Box.Sample(mask) throws in the
 * reference, Box.cs:70 -> GYMNET_ERR_UNSUPPORTED). */
int gymnet_vecenv_sample_actions_masked_device(gymnet_vecenv *h, int32_t *d_actions, const uint8_t *d_mask, int64_t mask_stride,
                                               uint64_t seed, uint64_t tick);
/* The caller's epsilon-greedy composer, batched (examples/ReinforcementLearning/ReinforcementLearning/PlaySessions/
 * TrainingPlaySession.cs:46-52: `if (Random.NextDouble() <= _epsilon) return ActionSpace.Sample(); return policy action`).
 * Lane i: u = 24-bit uniform from word B of (seed, global lane, tick); out[i] = (u <= epsilon) ? the
 * Discrete.Sample() draw of gymnet_vecenv_sample_actions_device for the same (seed, tick) : d_policy_actions[i].
 * Discrete action spaces only. */
int gymnet_vecenv_compose_actions_device(gymnet_vecenv *h, const int32_t *d_policy_actions, float epsilon, int32_t *d_actions_out,
                                         uint64_t seed, uint64_t tick);

/* ---- multi-GPU group: ONE process (e.g. a C# host) driving G members, one per GPU ----------------------------
 * The reference has no multi-device code; what shards is the independence of VecEnvWrapper's map (VecEnvWrapper.cs:22-24).
 * Member m owns the contiguous global lanes [m*N/G, (m+1)*N/G) (reset draws keyed by GLOBAL lane id: results do not depend
 * on G).  Every member keeps a replica of the whole batch's observations on its own GPU, rank-major [G][obs_dim][N/G]; the
 * member's live observation arrays ARE slice [m] of its replica (zero-copy send side).  gymnet_group_allgather_obs completes
 * the replicas:
 *   DIRECT  hand-written full-mesh push: each member stores its slice into the G-1 peers' replicas through peer-mapped
 *           memory, one peer per xGMI link, all links concurrently (~110 us for 16 MiB at 8 GPUs vs ~770 us for a ring);
 *   RCCL    ncclAllGather (in place) on per-member communicators (ncclCommInitAll); needs G distinct devices.
 * With GYMNET_FLAG_DOUBLE_BUFFER the gather runs on per-member side streams and overlaps the next step.
 * All calls are stream-ordered and non-blocking unless they take host buffers. */
typedef struct gymnet_group gymnet_group;
typedef enum gymnet_gather_mode { GYMNET_GATHER_NONE = 0, GYMNET_GATHER_DIRECT = 1, GYMNET_GATHER_RCCL = 2 } gymnet_gather_mode;

typedef struct gymnet_group_config {
    uint32_t struct_size;       /* = sizeof(gymnet_group_config) */
    int32_t  env_id;
    int64_t  global_num_envs;   /* N: lanes of the whole batch; a multiple of num_members */
    int32_t  num_members;       /* G, 1..16 */
    uint32_t flags;             /* GYMNET_FLAG_* for every member */
    uint64_t seed;
    const int32_t *devices;     /* [G] HIP device ordinal per member; NULL = 0..G-1.  An ordinal may repeat (several logical
                                   members on one GPU — how a 1-GPU box tests the path); RCCL needs distinct ordinals */
    int32_t  gather;            /* gymnet_gather_mode */
    int32_t  max_episode_steps;
} gymnet_group_config;

int gymnet_group_create(const gymnet_group_config *cfg, gymnet_group **out);
int gymnet_group_destroy(gymnet_group *g);
int gymnet_group_size(gymnet_group *g, int32_t *num_members, int64_t *lanes_per_member);
/* The member's handle (borrowed: destroyed with the group).  Everything gymnet_vecenv_* offers works on it. */
int gymnet_group_member(gymnet_group *g, int32_t member, gymnet_vecenv **out);
int gymnet_group_seed(gymnet_group *g, uint64_t seed);                        /* VecEnv.Seed(int), VecEnv.cs:44-46 */
int gymnet_group_reset_device(gymnet_group *g);                               /* VecEnv.Reset() on every member */
/* One vector step of the whole batch = one kernel launch per member.  d_actions[m]: pointer ON member m's device to that
 * member's N/G actions. */
int gymnet_group_step_device(gymnet_group *g, const void *const *d_actions);
/* `steps` vector steps per member (gymnet_vecenv_rollout_device on each, launches interleaved across members). */
int gymnet_group_rollout_device(gymnet_group *g, const void *const *d_actions, int64_t steps, int64_t action_stride, int64_t ring);
/* Completes every member's replica with the observations of the most recent step / reset.  Work queued on a member's stream
 * after gymnet_group_wait_gather (or, without DOUBLE_BUFFER, after this call) sees all G slices. */
int gymnet_group_allgather_obs(gymnet_group *g);
int gymnet_group_wait_gather(gymnet_group *g);
/* Member m's replica that was gathered last: device [G][obs_dim][N/G] on m's GPU — float32, or float64 when the group was created
 * with GYMNET_FLAG_F64 (here and in the three calls below). */
int gymnet_group_global_obs(gymnet_group *g, int32_t member, void **d_obs_all);
/* The same replica copied to the host, [G][obs_dim][N/G]; waits for the last gather and blocks. */
int gymnet_group_read_replica(gymnet_group *g, int32_t member, void *replica_out);
int gymnet_group_sync(gymnet_group *g);
/* Host-boundary forms over the whole batch (NDArray-shaped: obs [N, obs_dim], reward [N], done [N]; any may be NULL). */
int gymnet_group_reset(gymnet_group *g, void *obs_out);
int gymnet_group_step(gymnet_group *g, const void *actions, void *obs_out, float *reward_out, uint8_t *done_out);

/* ---- peer buffers: the direct all-gather for ONE PROCESS PER GPU hosts ---------------------------------------------
 * gymnet_group_* needs all GPUs in one process.  A host that runs one process per GPU (torch.distributed, MPI, a .NET
 * launcher) gets the same hand-written push over xGMI with these four calls: every rank creates its replica buffer
 * [G][obs_dim][N/G] as a peer buffer, hands the 64-byte handle to the other ranks by whatever channel it has, opens theirs,
 * creates its handle with d_ext_obs = buffer + rank * obs_dim * (N/G), and after a step pushes its slice into every
 * peer's replica.  Cross-process ordering is the host's: synchronize the stream, then a node barrier, before anyone reads a
 * replica or pushes into it again.  Teardown in the same spirit: barrier, every rank closes what it opened, barrier,
 * then every rank destroys what it created (no exporter frees memory a peer still has mapped).
 * (HIP IPC; needs HSA_ENABLE_IPC_MODE_LEGACY=0 on this platform's driver.) */
typedef struct gymnet_ipc_handle { char bytes[64]; } gymnet_ipc_handle;
int gymnet_peer_buffer_create(int device, int64_t bytes, void **d_ptr, gymnet_ipc_handle *handle);   /* hipMalloc + zero + export */
int gymnet_peer_buffer_open(int device, const gymnet_ipc_handle *handle, void **d_ptr);              /* map a peer's buffer */
int gymnet_peer_buffer_close(int device, void *d_ptr);                                               /* unmap an opened buffer */
int gymnet_peer_buffer_destroy(int device, void *d_ptr);                                             /* free a created buffer */
/* Stores `count` 4-byte words (floats; a float64 slice counts two per element) from d_src into the same-shaped slice d_dst[p] of every peer (p < npeers <= 15), all peers
 * concurrently (one grid row per peer = one xGMI link each), on `stream` (a hipStream_t of `device`; NULL = default). */
int gymnet_push_obs_device(int device, void *stream, const float *d_src, float *const *d_dst, int32_t npeers, int64_t count);

#ifdef __cplusplus
}
#endif
#endif /* GYMNET_AMD_H */
