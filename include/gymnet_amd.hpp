// gymnet_amd.hpp — header-only C++17 host mirror of the reference interface for this path, over the C ABI
// (gymnet_amd.h).  The reference is compiled C#; with no .NET toolchain in the build image this is the
// compiled-language host side: same member names, argument meaning and error behaviour as
//   VecEnv / IVecEnv   src/Gym/Envs/VecEnv.cs:12-93, src/Gym/Envs/IVecEnv.cs:8-19
//   Env / IEnv         src/Gym/Envs/Env.cs:13-41
//   Space/Box/Discrete src/Gym/Spaces/{Space.cs:5-18,Box.cs:15-96,Discrete.cs:5-44}
//   Step               src/Gym/Observations/Step.cs:7-29
//   exceptions         src/Gym/Exceptions/*.cs
// (paths relative to the Gym.NET tree).  NDArray is std::vector<float> here (row-major [N, D]).
// Nothing in this header computes: every call lands in libgymnet_amd.so's HIP kernels.
#pragma once
#include <cmath>
#include <cstdint>
#include <limits>
#include <memory>
#include <random>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

#include "gymnet_amd.h"

namespace gymnet {

// ---- exceptions (src/Gym/Exceptions) ---------------------------------------------------------------------
struct GymNetError : std::runtime_error { using std::runtime_error::runtime_error; };
struct NoDeviceError : GymNetError { using GymNetError::GymNetError; };
struct InvalidActionError : std::runtime_error {            // InvalidActionError.cs:7-10
    explicit InvalidActionError(const std::string &m = "Action is outside of the configured action space.") : std::runtime_error(m) {}
};
struct AlreadySteppingError : std::runtime_error {          // AlreadySteppingError.cs:8-10
    AlreadySteppingError() : std::runtime_error("already running an async step") {}
};
struct NotSteppingError : std::runtime_error {              // NotSteppingError.cs:4-6
    NotSteppingError() : std::runtime_error("not running an async step") {}
};

/// status -> the exception the reference throws for the same condition
inline void check(int status) {
    if (status == GYMNET_OK) return;
    const char *m = gymnet_last_error();
    std::string msg = (m && *m) ? m : gymnet_status_string(status);
    switch (status) {
        case GYMNET_ERR_INVALID_ARG: throw std::invalid_argument(msg);          // ArgumentException (VecEnv.cs:49)
        case GYMNET_ERR_INVALID_ACTION: throw InvalidActionError(msg);
        case GYMNET_ERR_ALREADY_STEPPING: throw AlreadySteppingError();
        case GYMNET_ERR_NOT_STEPPING: throw NotSteppingError();
        case GYMNET_ERR_NO_DEVICE: throw NoDeviceError(msg);
        case GYMNET_ERR_OOM: throw std::bad_alloc();
        case GYMNET_ERR_UNSUPPORTED: throw std::logic_error(msg);               // NotSupportedException
        default: throw GymNetError(msg + " (gymnet status " + std::to_string(status) + ")");
    }
}

// ---- spaces ----------------------------------------------------------------------------------------------
class Space {                                                // Space.cs:5-18
public:
    std::vector<int64_t> Shape;
    virtual ~Space() = default;
    virtual void Seed(int seed) = 0;
};

class Discrete : public Space {                              // Discrete.cs:5-44
public:
    int N, Start;
    explicit Discrete(int n, int seed = -1, int start = 0) : N(n), Start(start), rng_(seed == -1 ? std::random_device{}() : (unsigned)seed) { Shape = {n}; }
    int Sample() { return Start + (int)(rng_() % (unsigned)N); }                // Discrete.cs:27: Start + randint(0, N)
    int Sample(const std::vector<int> &mask) {                                  // Discrete.cs:18-26
        std::vector<int> valid;
        for (int i = 0; i < (int)mask.size(); ++i) if (mask[i] == 1) valid.push_back(i);
        return valid.empty() ? Start : Start + valid[rng_() % valid.size()];
    }
    bool Contains(int x) const { return x >= 0 && x < N; }                      // Discrete.cs:38-40 (ignores Start)
    void Seed(int seed) override { rng_.seed((unsigned)seed); }
private:
    std::mt19937 rng_;
};

class Box : public Space {                                   // Box.cs:15-96
public:
    enum class BoundedManner { Both, Below, Above };         // Box.cs:9-14
    std::vector<float> Low, High;
    std::vector<bool> BoundedLow, BoundedHigh;
    Box(float low, float high, int64_t n = 1, int seed = -1) : Box(std::vector<float>((size_t)n, low), std::vector<float>((size_t)n, high), seed) {}
    Box(std::vector<float> low, std::vector<float> high, int seed = -1)
        : Low(std::move(low)), High(std::move(high)), rng_(seed == -1 ? std::random_device{}() : (unsigned)seed) {
        if (Low.size() != High.size()) throw std::invalid_argument("low/high shape mismatch");
        Shape = {(int64_t)Low.size()};
        for (size_t i = 0; i < Low.size(); ++i) {                               // CheckBounded, Box.cs:49-54
            BoundedLow.push_back(Low[i] > -std::numeric_limits<float>::infinity());
            BoundedHigh.push_back(High[i] < std::numeric_limits<float>::infinity());
        }
    }
    bool IsBounded(BoundedManner manner = BoundedManner::Both) const {          // Box.cs:56-70
        bool below = true, above = true;
        for (size_t i = 0; i < Low.size(); ++i) { below = below && BoundedLow[i]; above = above && BoundedHigh[i]; }
        return manner == BoundedManner::Both ? (below && above) : manner == BoundedManner::Above ? above : below;
    }
    std::vector<float> Sample() {                                               // Box.cs:72-93, the reference's four regimes
        std::vector<float> s(Low.size());
        for (size_t i = 0; i < s.size(); ++i) {
            if (BoundedLow[i] && BoundedHigh[i]) s[i] = std::uniform_real_distribution<float>(Low[i], High[i])(rng_);
            else if (BoundedLow[i]) s[i] = std::exponential_distribution<float>(1.0f)(rng_) + Low[i];
            else if (BoundedHigh[i]) s[i] = std::exponential_distribution<float>(1.0f)(rng_) + High[i];   // sic, Box.cs:84
            else s[i] = std::normal_distribution<float>(0.5f, 1.0f)(rng_);                                // sic, Box.cs:82
        }
        return s;
    }
    bool Contains(const std::vector<float> &x) const {                          // Box.cs:95-99
        if (x.size() != Low.size()) return false;
        for (size_t i = 0; i < x.size(); ++i) if (!(x[i] >= Low[i] && x[i] <= High[i])) return false;
        return true;
    }
    void Seed(int seed) override { rng_.seed((unsigned)seed); }
private:
    std::mt19937 rng_;
};

// ---- Step records ------------------------------------------------------------------------------------------
struct Step {                                                // Step.cs:7-20; Information is always null on this path
    std::vector<float> Observation;
    float Reward = 0.0f;
    bool Done = false;
};

/// Step[] (IVecEnv.cs:15) as three arrays; operator[] materialises the reference's per-env record.
struct BatchStep {
    std::vector<float> Observation;      // row-major [N, D]
    std::vector<float> Reward;           // [N]
    std::vector<uint8_t> Done;           // [N]
    int64_t N = 0;
    int D = 0;
    Step operator[](int64_t i) const {
        Step s;
        s.Observation.assign(Observation.begin() + i * D, Observation.begin() + (i + 1) * D);
        s.Reward = Reward[(size_t)i];
        s.Done = Done[(size_t)i] != 0;
        return s;
    }
    int64_t size() const { return N; }
};

// ---- VectorEnv : VecEnv ------------------------------------------------------------------------------------
class VectorEnv {
public:
    VectorEnv(gymnet_env_id env, int64_t num_envs, int device = 0, uint64_t seed = 0, uint32_t flags = 0,
              int64_t lane_offset = 0, void *stream = nullptr) {
        // this class hands the library std::vector<float> observation buffers: a float64 handle (8 bytes per element) would overrun
        // them, so the flag is refused HERE; CartPoleEnv64 below is the float64 host class
        if (flags & GYMNET_FLAG_F64) throw std::logic_error("gymnet::VectorEnv is the float32 host class; GYMNET_FLAG_F64 handles: use gymnet::CartPoleEnv64 or the C ABI");
        check(gymnet_env_describe((int)env, &info_));
        // everything that can throw is built BEFORE the native handle exists: a constructor that throws runs no
        // destructor, so an allocation failing after gymnet_vecenv_create would leak the handle
        if (info_.action_is_box) action_box_ = std::make_unique<Box>(info_.action_low, info_.action_high, 1);
        else action_discrete_ = std::make_unique<Discrete>(info_.action_n);
        observation_space_ = std::make_unique<Box>(std::vector<float>(info_.obs_low, info_.obs_low + info_.obs_dim),
                                                   std::vector<float>(info_.obs_high, info_.obs_high + info_.obs_dim));
        gymnet_config cfg{};
        cfg.struct_size = sizeof cfg; cfg.env_id = (int)env; cfg.num_envs = num_envs; cfg.lane_offset = lane_offset;
        cfg.device = device; cfg.flags = flags; cfg.seed = seed; cfg.stream = stream;
        n_ = num_envs;
        check(gymnet_vecenv_create(&cfg, &h_));
    }
    VectorEnv(const VectorEnv &) = delete;
    VectorEnv &operator=(const VectorEnv &) = delete;
    ~VectorEnv() { Close(); }

    int64_t NumberOfEnvironments() const { return n_; }                         // VecEnv.cs:24
    int ObsDim() const { return info_.obs_dim; }
    int StateDim() const { return info_.state_dim; }
    Discrete *ActionSpaceDiscrete() const { return action_discrete_.get(); }    // CartPoleEnv.cs:47
    Box *ActionSpaceBox() const { return action_box_.get(); }
    const Box &ObservationSpace() const { return *observation_space_; }         // CartPoleEnv.cs:48
    std::pair<float, float> RewardRange() const { return {info_.reward_low, info_.reward_high}; }

    void Close() {                                                              // VecEnvWrapper.cs:26-30
        if (h_) { gymnet_vecenv_destroy(h_); h_ = nullptr; }
    }
    void Seed(uint64_t seed) { check(gymnet_vecenv_seed(h_, seed)); }           // VecEnv.cs:44-46
    void Seed(const std::vector<int> &seeds) {                                  // VecEnv.cs:48-53
        std::vector<uint64_t> s(seeds.begin(), seeds.end());
        check(gymnet_vecenv_seed_lanes(h_, s.data(), (int64_t)s.size()));
    }

    std::vector<float> Reset() {                                                // VecEnvWrapper.cs:18-20
        std::vector<float> obs((size_t)n_ * info_.obs_dim);
        check(gymnet_vecenv_reset(h_, obs.data()));
        return obs;
    }
    std::vector<float> ResetWhere(const std::vector<uint8_t> *mask = nullptr) { // README.md:36-40, batched
        if (mask && (int64_t)mask->size() != n_) throw std::invalid_argument("mask length must equal NumberOfEnvironments");
        std::vector<float> obs((size_t)n_ * info_.obs_dim);
        check(gymnet_vecenv_reset_where(h_, mask ? mask->data() : nullptr, obs.data()));
        return obs;
    }
    BatchStep Step(int action) {                                                // IVecEnv.Step(int), VecEnvWrapper.cs:22-24
        BatchStep b = make_batch();
        check(gymnet_vecenv_step_broadcast(h_, action, b.Observation.data(), b.Reward.data(), b.Done.data()));
        return b;
    }
    BatchStep Step(const std::vector<int32_t> &actions) {                       // extension: one action per lane
        if ((int64_t)actions.size() != n_) throw std::invalid_argument("Number of actions passed should be equals to number of environments");
        BatchStep b = make_batch();
        check(gymnet_vecenv_step(h_, actions.data(), b.Observation.data(), b.Reward.data(), b.Done.data()));
        return b;
    }
    BatchStep Step(const std::vector<float> &actions) {                         // Box actions
        if ((int64_t)actions.size() != n_) throw std::invalid_argument("Number of actions passed should be equals to number of environments");
        BatchStep b = make_batch();
        check(gymnet_vecenv_step(h_, actions.data(), b.Observation.data(), b.Reward.data(), b.Done.data()));
        return b;
    }
    void StepAsync(const std::vector<int32_t> &actions) {                       // VecEnv.cs:63-65
        if ((int64_t)actions.size() != n_) throw std::invalid_argument("Number of actions passed should be equals to number of environments");
        check(gymnet_vecenv_step_async(h_, actions.data()));
    }
    BatchStep StepWait() {
        BatchStep b = make_batch();
        check(gymnet_vecenv_step_wait(h_, b.Observation.data(), b.Reward.data(), b.Done.data()));
        return b;
    }

    std::vector<float> GetState() const {                                       // SoA [state_dim][N]
        std::vector<float> s((size_t)n_ * info_.state_dim);
        check(gymnet_vecenv_get_state(h_, s.data()));
        return s;
    }
    void SetState(const std::vector<float> &soa) {
        if ((int64_t)soa.size() != n_ * info_.state_dim) throw std::invalid_argument("state must hold state_dim * N floats");
        check(gymnet_vecenv_set_state(h_, soa.data()));
    }
    std::vector<int32_t> GetStepsBeyondDone() const {                           // CartPoleEnv.cs:41 per lane
        std::vector<int32_t> b((size_t)n_);
        check(gymnet_vecenv_get_steps_beyond_done(h_, b.data()));
        return b;
    }
    gymnet_counters Counters() const { gymnet_counters c{}; check(gymnet_vecenv_counters(h_, &c)); return c; }

    // ---- ABI 3 ------------------------------------------------------------------------------------------------------
    /// The kernel instantiation the next step launch runs, as the launcher itself resolves it.
    std::string KernelName() const { char buf[128] = {0}; check(gymnet_vecenv_kernel_name(h_, buf, (int32_t)sizeof buf)); return buf; }
    /// Library-owned page-locked, device-mapped host buffers (valid until Close): stepping THROUGH them needs no staging copies.
    struct PinnedBuffers { void *actions; float *obs; float *reward; uint8_t *done; };
    PinnedBuffers HostBuffers() {
        PinnedBuffers b{};
        void *obs = nullptr;                                                    // typed by the handle; this class creates float32 handles
        check(gymnet_vecenv_host_buffers(h_, &b.actions, &obs, &b.reward, &b.done));
        b.obs = static_cast<float *>(obs);
        return b;
    }
    // ---- ABI 4 ------------------------------------------------------------------------------------------------------
    /// Override fields of the step kernel's launch configuration (-1 = keep); every configuration is bit-identical.
    void SetLaunchPolicy(gymnet_launch_policy p) { p.struct_size = sizeof p; check(gymnet_vecenv_set_launch_policy(h_, &p)); }
    static gymnet_launch_policy KeepPolicy() { gymnet_launch_policy p; p.struct_size = sizeof p; p.vec = p.block = p.nt = p.sequential_lanes = p.reset_form = p.lds_pipe = p.occupancy_lds_bytes = p.graph = -1; return p; }
    gymnet_launch_policy GetLaunchPolicy() const { gymnet_launch_policy p{}; check(gymnet_vecenv_get_launch_policy(h_, &p)); return p; }
    /// Any per-lane array by id (checkpoint / resume of every configuration): T must be the array's element type.
    template <class T> std::vector<T> GetArray(gymnet_array_id which, size_t count) const {
        std::vector<T> a(count);
        check(gymnet_vecenv_get_array(h_, (int32_t)which, a.data(), (int64_t)(count * sizeof(T))));
        return a;
    }
    template <class T> void SetArray(gymnet_array_id which, const std::vector<T> &a) { check(gymnet_vecenv_set_array(h_, (int32_t)which, a.data(), (int64_t)(a.size() * sizeof(T)))); }
    /// gymnet_vecenv_step with caller-owned buffers (nothing is allocated): the pinned ones above, or any host memory.
    void StepInto(const void *actions, float *obs_out, float *reward_out, uint8_t *done_out) { check(gymnet_vecenv_step(h_, actions, obs_out, reward_out, done_out)); }
    /// Compact records of the lanes that finished in the most recent step (DONE_LIST [+ EPISODE_STATS] [+ FINAL_OBS]).
    struct DoneRecordSet { std::vector<int32_t> lanes; std::vector<float> episode_return; std::vector<int32_t> episode_length; std::vector<float> final_obs; };
    DoneRecordSet DoneRecords(bool episode, bool final_obs) {
        DoneRecordSet r;
        r.lanes.resize((size_t)n_);
        if (episode) { r.episode_return.resize((size_t)n_); r.episode_length.resize((size_t)n_); }
        if (final_obs) r.final_obs.resize((size_t)n_ * info_.obs_dim);
        int64_t c = 0;
        check(gymnet_vecenv_done_records(h_, r.lanes.data(), episode ? r.episode_return.data() : nullptr, episode ? r.episode_length.data() : nullptr,
                                         final_obs ? r.final_obs.data() : nullptr, n_, &c));
        r.lanes.resize((size_t)c);
        if (episode) { r.episode_return.resize((size_t)c); r.episode_length.resize((size_t)c); }
        if (final_obs) r.final_obs.resize((size_t)c * info_.obs_dim);
        return r;
    }

    // device-resident path
    void ResetDevice() { check(gymnet_vecenv_reset_device(h_)); }
    void StepDevice(const void *d_actions) { check(gymnet_vecenv_step_device(h_, d_actions)); }
    void RolloutDevice(const void *d_actions, int64_t steps, int64_t stride, int64_t ring) { check(gymnet_vecenv_rollout_device(h_, d_actions, steps, stride, ring)); }
    void Sync() { check(gymnet_vecenv_sync(h_)); }
    gymnet_device_view DeviceView() const { gymnet_device_view v{}; check(gymnet_vecenv_device_view(h_, &v)); return v; }
    gymnet_vecenv *handle() const { return h_; }

private:
    BatchStep make_batch() const {
        BatchStep b;
        b.N = n_; b.D = info_.obs_dim;
        b.Observation.resize((size_t)n_ * info_.obs_dim); b.Reward.resize((size_t)n_); b.Done.resize((size_t)n_);
        return b;
    }
    gymnet_vecenv *h_ = nullptr;
    gymnet_env_info info_{};
    int64_t n_ = 0;
    std::unique_ptr<Discrete> action_discrete_;
    std::unique_ptr<Box> action_box_, observation_space_;
};

/// ONE process driving G GPUs (gymnet_group_*): member m owns lanes [m*N/G, (m+1)*N/G); every member keeps a replica
/// [G][obs_dim][N/G] of all observations on its GPU, completed by AllGatherObs() (hand-written direct push over peer-mapped
/// memory, or RCCL).  The host-boundary Reset / Step take the whole batch in the NDArray layout, like VectorEnv.
class GroupVectorEnv {
public:
    GroupVectorEnv(gymnet_env_id env, int64_t global_num_envs, const std::vector<int32_t> &devices, uint64_t seed = 0,
                   uint32_t flags = GYMNET_FLAG_AUTORESET, gymnet_gather_mode gather = GYMNET_GATHER_DIRECT) {
        check(gymnet_env_describe((int)env, &info_));
        gymnet_group_config cfg{};
        cfg.struct_size = sizeof cfg; cfg.env_id = (int)env; cfg.global_num_envs = global_num_envs;
        cfg.num_members = (int32_t)devices.size(); cfg.flags = flags; cfg.seed = seed; cfg.devices = devices.data(); cfg.gather = gather;
        n_ = global_num_envs; g_members_ = (int)devices.size();
        check(gymnet_group_create(&cfg, &g_));
    }
    GroupVectorEnv(const GroupVectorEnv &) = delete;
    GroupVectorEnv &operator=(const GroupVectorEnv &) = delete;
    ~GroupVectorEnv() { Close(); }
    void Close() { if (g_) { gymnet_group_destroy(g_); g_ = nullptr; } }

    int64_t NumberOfEnvironments() const { return n_; }
    int NumMembers() const { return g_members_; }
    int ObsDim() const { return info_.obs_dim; }
    void Seed(uint64_t seed) { check(gymnet_group_seed(g_, seed)); }

    std::vector<float> Reset() {                                                // row-major [N, obs_dim]
        std::vector<float> obs((size_t)n_ * info_.obs_dim);
        check(gymnet_group_reset(g_, obs.data()));
        return obs;
    }
    BatchStep Step(const std::vector<int32_t> &actions) {
        if ((int64_t)actions.size() != n_) throw std::invalid_argument("Number of actions passed should be equals to number of environments");
        BatchStep b;
        b.N = n_; b.D = info_.obs_dim;
        b.Observation.resize((size_t)n_ * info_.obs_dim); b.Reward.resize((size_t)n_); b.Done.resize((size_t)n_);
        check(gymnet_group_step(g_, actions.data(), b.Observation.data(), b.Reward.data(), b.Done.data()));
        return b;
    }
    BatchStep Step(int action) { return Step(std::vector<int32_t>((size_t)n_, action)); }   // IVecEnv.Step(int): scalar broadcast

    void AllGatherObs() { check(gymnet_group_allgather_obs(g_)); }
    void WaitGather() { check(gymnet_group_wait_gather(g_)); }
    std::vector<float> ReadReplica(int member) {                               // [G][obs_dim][N/G] as member `member` holds it
        std::vector<float> r((size_t)n_ * info_.obs_dim);
        check(gymnet_group_read_replica(g_, member, r.data()));
        return r;
    }
    void Sync() { check(gymnet_group_sync(g_)); }
    gymnet_group *handle() const { return g_; }

private:
    gymnet_group *g_ = nullptr;
    gymnet_env_info info_{};
    int64_t n_ = 0;
    int g_members_ = 0;
};

/// Single-instance Env façade in the reference's OWN arithmetic (GYMNET_FLAG_F64): float64 state, float64 observations — what
/// `new CartPoleEnv().Step(a).Observation` really holds (CartPoleEnv.cs:141-166,185) — so the reference's loop
/// (README.md:32-52) sees the reference's states to the last few ulps and its exact episode lengths, free-running.
struct Step64 { std::vector<double> Observation; float Reward = 0.0f; bool Done = false; };
class CartPoleEnv64 {
public:
    explicit CartPoleEnv64(int device = 0, uint64_t seed = 0, uint32_t extra_flags = 0) {
        gymnet_config cfg{};
        cfg.struct_size = sizeof cfg; cfg.env_id = GYMNET_ENV_CARTPOLE; cfg.num_envs = 1; cfg.device = device;
        cfg.flags = GYMNET_FLAG_F64 | extra_flags; cfg.seed = seed;
        // (the resident latency path is opt-in: pass GYMNET_FLAG_RESIDENT in extra_flags — see CartPoleEnv below for when it pays)
        check(gymnet_vecenv_create(&cfg, &h_));
    }
    CartPoleEnv64(const CartPoleEnv64 &) = delete;
    CartPoleEnv64 &operator=(const CartPoleEnv64 &) = delete;
    ~CartPoleEnv64() { CloseEnvironment(); }
    std::vector<double> Reset() { std::vector<double> o(4); check(gymnet_vecenv_reset(h_, o.data())); return o; }   // CartPoleEnv.cs:63-67
    Step64 Step(int action) {                                                   // CartPoleEnv.cs:137-186
        Step64 s; s.Observation.resize(4);
        uint8_t d = 0;
        const int32_t a = action;
        check(gymnet_vecenv_step(h_, &a, s.Observation.data(), &s.Reward, &d));
        s.Done = d != 0;
        return s;
    }
    void SetState(const std::vector<double> &x) { if (x.size() != 4) throw std::invalid_argument("state must hold 4 doubles"); check(gymnet_vecenv_set_state(h_, x.data())); }
    std::vector<double> GetState() const { std::vector<double> x(4); check(gymnet_vecenv_get_state(h_, x.data())); return x; }
    void Seed(int seed) { check(gymnet_vecenv_seed(h_, (uint64_t)seed)); }      // CartPoleEnv.cs:196-198
    void CloseEnvironment() { if (h_) { gymnet_vecenv_destroy(h_); h_ = nullptr; } }   // CartPoleEnv.cs:189-194
    gymnet_vecenv *handle() const { return h_; }
private:
    gymnet_vecenv *h_ = nullptr;
};

/// Single-instance Env façade (Env.cs:13-41; CartPoleEnv.cs:43-198) over a 1-lane float32 VectorEnv (the batched engine's
/// arithmetic: 1e-5 per teacher-forced step; CartPoleEnv64 above is the reference-exact one).
class CartPoleEnv {
public:
    /// resident = true (opt-in since round 6): GYMNET_FLAG_RESIDENT — Step / Reset are served by a resident single-wave kernel through a
    /// mailbox in pinned host memory (no launch, no synchronize per call; ~3x lower latency for a loop that only steps; bit-identical).
    /// While it waits for the next command the kernel occupies the handle's stream for up to ~5 ms: a device-wide synchronize elsewhere
    /// in the process waits for that, which is why a host that also trains on the GPU keeps the default (one launch per call).
    explicit CartPoleEnv(int device = 0, uint64_t seed = 0, bool resident = false) : v_(GYMNET_ENV_CARTPOLE, 1, device, seed, resident ? GYMNET_FLAG_RESIDENT : 0u) {}
    std::vector<float> Reset() { return v_.Reset(); }                           // CartPoleEnv.cs:63-67
    gymnet::Step Step(int action) { return v_.Step(std::vector<int32_t>{action})[0]; }   // CartPoleEnv.cs:137-186
    /// Env<TAction>.Step(TAction) where TAction : Enum (Env.cs:43-53): the enum's integer value is the discrete action
    template <class TAction, std::enable_if_t<std::is_enum<TAction>::value, int> = 0>
    gymnet::Step Step(TAction action) { return Step(static_cast<int>(action)); }
    void Seed(int seed) { v_.Seed((uint64_t)seed); }                            // CartPoleEnv.cs:196-198
    void CloseEnvironment() { v_.Close(); }                                     // CartPoleEnv.cs:189-194
    Discrete &ActionSpace() { return *v_.ActionSpaceDiscrete(); }
    const Box &ObservationSpace() const { return v_.ObservationSpace(); }
    VectorEnv &vector() { return v_; }
private:
    VectorEnv v_;
};

}  // namespace gymnet
