// kernels.hpp — launch interface between the C-ABI layer (capi.hip) and the HIP kernels (kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gymnet {

// Everything one vector-step launch needs.  Passed by value (kernarg segment).
// R is the STATE SCALAR of the handle: float (the batched engine's structure-of-arrays hot path) or double (GYMNET_FLAG_F64: the
// reference's own arithmetic, CartPoleEnv.cs:141-166,185 — float64 state that IS the float64 observation).  Everything that holds a
// state or observation value is typed R; reward (Step.cs:9: a C# float), done, the actions and all counters are the same for both.
template <class R>
struct StepArgsT {
    R *state;                // [S][state_stride]  structure-of-arrays, one env per lane (read)
    R *state_out;            // where the new state is written: == state, or the OTHER half of a double-buffered pair
                             // (GYMNET_FLAG_DOUBLE_BUFFER: step t+1 writes B while a gather of A is still in flight)
    R *obs;                  // [O][obs_stride]    (unused when the env's observation aliases its state)
    const R *obs_in;         // the CURRENT observation buffer (== obs unless double-buffered): state rows the observation repeats
                             // (Env::OBS_ROW_OF_STATE) are read from here — they have no row of their own in `state`
    const void *action;      // int32[n] (Discrete) or float32[n] (Box)
    float *reward;           // [n]
    uint8_t *done;           // [n]
    int32_t *sbd;            // [n] CartPole steps_beyond_done (CartPoleEnv.cs:41); NULL with auto-reset
    uint64_t *tick2;         // device tick, double-buffered: launch with tick t reads tick2[t&1], writes tick2[(t+1)&1]=t+1
    // extras (all NULL / 0 in the lean hot-path variant)
    R *final_obs;            // [O][n] dense terminal observations; NULL = none kept / compact records only
    // Done-lane compaction is SHARDED: 4096 waves hammering one counter word serialise at ~88 atomics/us (46 us per
    // step at 2^20 lanes, measured); wave w appends to shard w % kShards, each shard has its own counter (64-byte
    // stride) and its own segment of the list.  compact_done gathers the segments into one list on demand.
    int32_t *done_list;      // [kShards][done_cap] segmented
    uint32_t *done_count2;   // [2][kShards * kCountStride]: this launch fills half [cparity] and zeroes half [cparity^1]
    int64_t done_cap;        // entries per shard segment
    float *ep_ret; int32_t *ep_len;
    float *fin_ret; int32_t *fin_len;   // dense last-finished-episode arrays; NULL = compact records only (with done_list)
    // compact records, segmented like done_list (position p of shard s = the lane at done_list[s * done_cap + p]):
    float *rec_ret; int32_t *rec_len;   // [kShards][done_cap]     finished episode's return / length   (EPISODE_STATS + DONE_LIST)
    R *rec_obs;                         // [kShards][O][done_cap]  terminal observation                  (FINAL_OBS + DONE_LIST)
    const uint64_t *lane_seed;       // [n] per-lane Philox keys (VecEnv.Seed(int[])) or NULL
    unsigned long long *after_done;  // [kShards * kAfterStride] sharded counter: steps taken on already-done lanes (CartPoleEnv.cs:176-179)
    int64_t n, state_stride, obs_stride;
    uint64_t lane_offset, seed;
    int32_t parity;          // tick & 1 of this launch
    int32_t cparity;         // (number of step launches so far) & 1: which done_count2 slot this launch fills
    int32_t max_episode_steps;
};
typedef StepArgsT<float> StepArgs;
typedef StepArgsT<double> StepArgs64;

// vec: envs per thread (4 = dwordx4 streams, 1 = scalar); block: threads per workgroup;
// nt: non-temporal mask (0 none, 12 action + reward/done streams, 15 every stream)
constexpr int kShards = 256;        // power of two; the gather kernels' workgroup size (kernels.hip static_assert)
constexpr int kCountStride = 16;    // uint32 words between shard counters (64 bytes: one counter per cache line)
constexpr int kAfterStride = 8;     // uint64 words between after_done shards (64 bytes)

// lds_bytes: unused dynamic LDS requested per workgroup, for the ONE purpose of capping occupancy in probes (GYMNET_LDS)
// items: lanes per thread of the fully unrolled software-pipelined kernel (envs with PIPELINED only; 1 = one-shot kernel)
// reset_form: 1 = wave-compacted fused reset in the one-step kernel (reset_pending_wave: envs whose observation aliases the
// state, dwordx4 lanes)
// lds_pipe: 1 = the multi-lane kernel in its producer / consumer form (step_kernel_lds: `items` tiles per workgroup)
// simds: SIMD units of the handle's device (multiProcessorCount x 4; 1024 on MI355X) — sizes the resident generation of the looped
// multi-pair kernel (step_kernels.hpp pipe2_chunks) and the policy's waves-per-SIMD windows (capi.hip default_policy)
struct LaunchCfg { int vec; int block; int nt; int lds_bytes = 0; int items = 1; int reset_form = 0; int lds_pipe = 0; int simds = 1024; };

// ---- env-dependent launchers: one translation unit per env (env_*.hip, GYMNET_DEFINE_ENV in step_kernels.hpp) ---------------
#define GYMNET_DECLARE_ENV(tag, R)                                                                                              \
    hipError_t launch_step_##tag(bool autoreset, bool extras, const StepArgsT<R> &a, LaunchCfg cfg, hipStream_t st);            \
    int describe_step_##tag(bool autoreset, bool extras, LaunchCfg cfg, int64_t n, char *buf, size_t cap);                      \
    void resolved_shape_##tag(bool autoreset, bool extras, LaunchCfg cfg, int64_t n, int *vec, int *sequential);                \
    hipError_t launch_rollout_##tag(bool autoreset, bool extras, const StepArgsT<R> &a, const RolloutArgsT<R> &r, LaunchCfg cfg, hipStream_t st); \
    hipError_t launch_reset_##tag(const ResetArgsT<R> &a, hipStream_t st);                                                      \
    hipError_t launch_observe_##tag(const R *state, int64_t sstride, R *obs, int64_t ostride, int64_t n, hipStream_t st);            \
    hipError_t launch_resident_##tag(bool autoreset, bool extras, const StepArgsT<R> &a, const ResetArgsT<R> &r, Mailbox *mb, uint64_t idle_polls, hipStream_t st);

// Fused multi-step rollout: `steps` vector steps inside ONE launch; state stays in registers between steps.
// Round 5: the rollout carries what its consumer needs (examples/.../PlaySessions/BasePlaySession.cs:58-69 accumulates the episode
// reward and keeps the best episodes, MemoryTypes/ReplayMemory.cs:53-67; TrainingPlaySession.cs:46-52 draws epsilon-greedy actions
// per step): on a bookkeeping handle the running episode return / length live in registers, max_episode_steps truncates, per-lane
// seeds key the reset draws, and every finished episode leaves a compact (t, lane, return, length) record; and the actions can be
// DRAWN IN THE KERNEL — the words sample_discrete_kernel / compose_discrete_kernel would produce for (action_seed, action_tick0 + t)
// — so a random or epsilon-greedy rollout reads no action ring at all.
template <class R>
struct RolloutArgsT {
    int64_t steps;           // T
    int64_t action_stride;   // elements between consecutive action slices
    int64_t ring;            // step t reads slice t % ring
    R *rec_obs;              // optional [T][O][n]   (NULL = do not record)
    float *rec_reward;       // optional [T][n]
    uint8_t *rec_done;       // optional [T][n]
    void *rec_action;        // optional [T][n] the actions TAKEN (int32 / float32): what a replay memory stores beside obs / reward / done
    // action source: 0 = the ring (StepArgs::action); 1 = ActionSpace.Sample() per lane and step; 2 = epsilon-greedy — with
    // probability epsilon the sampled action, else the ring's (policy) action.  Philox action stream, counter (global lane, action_tick0 + t).
    int32_t action_source;
    float epsilon;
    uint64_t action_seed, action_tick0;
    // compact episode records of the WHOLE rollout, segmented like StepArgs::done_list: wave w appends to shard w % kShards
    // (ep_count[shard * kCountStride], one atomic per wave and step that has a finished lane), record p of shard s at s * ep_cap + p
    int32_t *ep_t, *ep_lane;      // step index inside the rollout, lane
    float *ep_ret; int32_t *ep_len;
    uint32_t *ep_count;           // [kShards * kCountStride], zeroed by the launcher; counts beyond ep_cap go to the overflow segment
    int64_t ep_cap;
    // Round 6: ONE overflow segment shared by all shards, for the records a shard's own segment cannot hold (lanes that finish
    // unevenly: a few waves producing most of the episodes).  It is "segment kShards" of the same four arrays — record p at
    // kShards * ep_cap + p, its counter at ep_count[kShards * kCountStride] — so the kernel needs no pointers of its own for it (the
    // rollout kernels are short of scalar registers: six more argument words cost the records variant 0.5 us per vector step).  The
    // counter counts every overflowed record; the first ov_cap are kept.  ov_cap = the caller's ep_capacity: records are dropped only
    // beyond ep_capacity in TOTAL.  records_no_overflow (GYMNET_RECORDS_NO_OVERFLOW): the kernel variant without the spill path.
    int64_t ov_cap;
    int32_t records_no_overflow;
};
typedef RolloutArgsT<float> RolloutArgs;
typedef RolloutArgsT<double> RolloutArgs64;

template <class R>
struct ResetArgsT {
    R *state; R *obs; int32_t *sbd; uint8_t *done;   // done flags cleared for reset lanes
    const uint8_t *mask;      // NULL = all lanes
    uint64_t *tick2; const uint64_t *lane_seed;
    float *ep_ret; int32_t *ep_len;
    int64_t n, state_stride, obs_stride;
    uint64_t lane_offset, seed;
    int32_t parity;
};
typedef ResetArgsT<float> ResetArgs;
typedef ResetArgsT<double> ResetArgs64;

// ---- resident single-wave kernel (GYMNET_FLAG_RESIDENT; N <= 64 lanes: the single-instance facade's latency path) -----------------
// The reference's usage shape is ONE env stepped in a host loop (README.md:32-52, Env.cs:13-41).  Served by one kernel launch +
// one stream synchronize per step that costs ~25 us; served by a RESIDENT kernel — one wave that stays on the GPU, polls a mailbox
// in page-locked, device-mapped, coherent HOST memory for commands and writes observation / reward / done back into it — a step
// is two PCIe crossings and no launch.  The kernel leaves by itself after `idle_polls` polls without a command (or on the exit
// command) and touches nothing afterwards; the host restarts it on the next command.  Same per-lane code (step_body / reset_lane)
// and Philox counters as the launch path: bit-identical.
constexpr int kMailboxLanes = 64;
struct Mailbox {
    uint64_t cmd_seq;                 // host -> device: sequence number of the newest command (the host writes it LAST, release)
    uint32_t cmd;                     // kMailboxStep / ResetAll / ResetDone / Exit
    uint32_t pad0;
    uint64_t done_seq;                // device -> host: the last command whose results are in the mailbox (written LAST, release)
    uint32_t exited;                  // device -> host: the kernel has left; it touches nothing afterwards
    uint32_t pad1;
    uint64_t tick;                    // device -> host: engine tick after the last command
    uint64_t pad2[3];
    int32_t actions[kMailboxLanes];   // int32 (Discrete) or float32 bits (Box), one per lane
    float reward[kMailboxLanes];
    uint8_t done[kMailboxLanes];
    // the observations follow at kMailboxObsOffset: row-major [lane][obs_dim] of the handle's state scalar (<= 64 * 6 * 8 bytes)
};
constexpr size_t kMailboxObsOffset = 1024;
constexpr size_t kMailboxBytes = kMailboxObsOffset + (size_t)kMailboxLanes * 8 * 8;
static_assert(sizeof(Mailbox) <= kMailboxObsOffset, "mailbox header overlaps the observations");
constexpr uint32_t kMailboxStep = 1, kMailboxResetAll = 2, kMailboxResetDone = 3, kMailboxExit = 4;

GYMNET_DECLARE_ENV(cartpole, float)
GYMNET_DECLARE_ENV(pendulum, float)
GYMNET_DECLARE_ENV(mountaincar, float)
GYMNET_DECLARE_ENV(acrobot, float)
GYMNET_DECLARE_ENV(cartpole64, double)      // GYMNET_FLAG_F64: CartPole in the reference's own binary64 arithmetic (cartpole64.hpp)

// env_id: gymnet_env_id; the state scalar of the arguments selects the float32 engine or (env 0 only) the float64 one.
// autoreset / extras select the compiled variant.  Returns hipError_t.
hipError_t launch_step(int env_id, bool autoreset, bool extras, const StepArgsT<float> &a, LaunchCfg cfg, hipStream_t st);
hipError_t launch_step(int env_id, bool autoreset, bool extras, const StepArgsT<double> &a, LaunchCfg cfg, hipStream_t st);

// The template instantiation launch_step would run for this configuration, as text ("step_kernel<CartPole,4,true,false,15,1>");
// returns the length, or < 0 for an unknown env.  n: the batch size (one form needs whole 2 * items * 256-lane groups).
int describe_step_kernel(int env_id, bool f64, bool autoreset, bool extras, LaunchCfg cfg, int64_t n, char *buf, size_t cap);
// what that instantiation is shaped like: lanes per thread on its wide accesses, and lanes (or lane pairs) a thread works through
// one after another in the multi-lane forms (1 = one-shot kernel)
void resolved_step_shape(int env_id, bool f64, bool autoreset, bool extras, LaunchCfg cfg, int64_t n, int *vec, int *sequential);

// Same results as `steps` launch_step calls (bitwise) — with r.action_source != 0: as `steps` x (sample / compose actions, then step).
hipError_t launch_rollout_fused(int env_id, bool autoreset, bool extras, const StepArgsT<float> &a, const RolloutArgsT<float> &r, LaunchCfg cfg, hipStream_t st);
hipError_t launch_rollout_fused(int env_id, bool autoreset, bool extras, const StepArgsT<double> &a, const RolloutArgsT<double> &r, LaunchCfg cfg, hipStream_t st);
// Gathers the kShards segments of a fused rollout's episode records (RolloutArgs::ep_*) into compact arrays out_*[0 .. *out_count);
// records beyond out_capacity are dropped, the count is the number of records kept by the rollout.  Any out array may be NULL.
struct EpisodeGatherArgs {
    const uint32_t *counts; int64_t cap;
    const int32_t *ep_t, *ep_lane; const float *ep_ret; const int32_t *ep_len;
    int64_t ov_cap;                // the shared overflow segment = segment kShards of the same arrays (RolloutArgs)
    int32_t *out_t, *out_lane; float *out_ret; int32_t *out_len;
    int64_t out_capacity; uint32_t *out_count;
};
hipError_t launch_gather_episodes(const EpisodeGatherArgs &a, hipStream_t st);

hipError_t launch_reset(int env_id, const ResetArgsT<float> &a, hipStream_t st);
hipError_t launch_reset(int env_id, const ResetArgsT<double> &a, hipStream_t st);
// starts the resident kernel of a small handle (a.n <= kMailboxLanes; a.action must point at mb->actions); mb: device-visible address
hipError_t launch_resident(int env_id, bool autoreset, bool extras, const StepArgsT<float> &a, const ResetArgsT<float> &r, Mailbox *mb, uint64_t idle_polls, hipStream_t st);
hipError_t launch_resident(int env_id, bool autoreset, bool extras, const StepArgsT<double> &a, const ResetArgsT<double> &r, Mailbox *mb, uint64_t idle_polls, hipStream_t st);

// observations: SoA [O][stride] -> row-major [n][O]; float32, or float64 for a GYMNET_FLAG_F64 handle (here and below)
hipError_t launch_pack_obs(int obs_dim, const float *obs, int64_t stride, float *out_rowmajor, int64_t n, hipStream_t st);
hipError_t launch_pack_obs(int obs_dim, const double *obs, int64_t stride, double *out_rowmajor, int64_t n, hipStream_t st);
// small batches at the host boundary: obs (row-major), reward and done written straight into ONE host-mapped buffer
// [n*O floats | n floats | n bytes] by a single kernel — no device-to-host memcpy calls on the latency path
hipError_t launch_export_small(int obs_dim, const float *obs, int64_t stride, const float *reward, const uint8_t *done,
                               float *out_obs, float *out_reward, uint8_t *out_done, int64_t n, hipStream_t st);
hipError_t launch_export_small(int obs_dim, const double *obs, int64_t stride, const float *reward, const uint8_t *done,
                               double *out_obs, float *out_reward, uint8_t *out_done, int64_t n, hipStream_t st);
// any batch size: obs (row-major), reward, done written straight into page-locked device-mapped HOST memory (the stores are the
// PCIe transfer); any out pointer may be NULL
hipError_t launch_export_host(int obs_dim, const float *obs, int64_t stride, const float *reward, const uint8_t *done,
                              float *out_obs, float *out_reward, uint8_t *out_done, int64_t n, hipStream_t st);
hipError_t launch_export_host(int obs_dim, const double *obs, int64_t stride, const float *reward, const uint8_t *done,
                              double *out_obs, float *out_reward, uint8_t *out_done, int64_t n, hipStream_t st);
// recompute obs from state (after set_state) for envs whose observation is derived
hipError_t launch_observe(int env_id, const float *state, int64_t state_stride, float *obs, int64_t obs_stride,
                          int64_t n, hipStream_t st);
hipError_t launch_fill_i32(int32_t *p, int32_t v, int64_t n, hipStream_t st);
// Gathers the sharded done list of one step (counter half `counts`) and the records written beside it into compact arrays
// out_*[0 .. *out_count) (entries beyond out_capacity are dropped; the count is the true one), and / or applies the records to
// the dense per-lane arrays.  Every out / dense / rec pointer may be NULL.
template <class R>
struct CompactArgsT {
    const uint32_t *counts; const int32_t *list; int64_t cap;
    const float *rec_ret; const int32_t *rec_len; const R *rec_obs; int32_t obs_dim;
    int32_t *out_list; float *out_ret; int32_t *out_len; R *out_obs;       // out_obs: row-major [count][obs_dim]
    int64_t out_capacity;
    uint32_t *out_count;
    float *dense_ret; int32_t *dense_len; R *dense_obs; int64_t n;         // dense_obs: [obs_dim][n]
};
typedef CompactArgsT<float> CompactArgs;
hipError_t launch_compact_done(const CompactArgsT<float> &a, hipStream_t st);
hipError_t launch_compact_done(const CompactArgsT<double> &a, hipStream_t st);
// counts actions outside [0, nvals) into *bad
hipError_t launch_validate_discrete(const int32_t *a, int64_t n, int32_t nvals, uint32_t *bad, hipStream_t st);
hipError_t launch_sample_discrete(int32_t *out, int64_t n, int32_t nvals, int32_t start, uint64_t seed,
                                  uint64_t lane_offset, uint64_t tick, hipStream_t st);
// Discrete.Sample(mask) (Discrete.cs:18-26): mask uint8 [n][nvals] (mask_stride = nvals) or one row for every lane (mask_stride = 0)
hipError_t launch_sample_discrete_masked(int32_t *out, int64_t n, int32_t nvals, int32_t start, const uint8_t *mask,
                                         int64_t mask_stride, uint64_t seed, uint64_t lane_offset, uint64_t tick, hipStream_t st);
// Direct all-gather, push form: copies `count` 4-byte words (floats; a float64 slice counts two per element) from src to the same offset inside each of `npeers` peer buffers
// (peer-mapped device memory reached over xGMI, or buffers on the same device for logical shards).
constexpr int kMaxPeers = 15;
struct PushArgs { const float *src; float *dst[kMaxPeers]; int64_t count; int32_t npeers; };
hipError_t launch_push_obs(const PushArgs &a, hipStream_t st);
hipError_t launch_compose_discrete(const int32_t *policy, int32_t *out, int64_t n, int32_t nvals, float epsilon, uint64_t seed,
                                   uint64_t lane_offset, uint64_t tick, hipStream_t st);
hipError_t launch_sample_box(float *out, int64_t n, float low, float high, uint64_t seed, uint64_t lane_offset,
                             uint64_t tick, hipStream_t st);

// Box.Sample() with per-element bounds (Box.cs:25-51,69-90): low / high device arrays of `dim` floats, out row-major [n][dim]
hipError_t launch_sample_box_elementwise(float *out, int64_t n, int32_t dim, const float *low, const float *high, uint64_t seed,
                                         uint64_t lane_offset, uint64_t tick, hipStream_t st);

}  // namespace gymnet
