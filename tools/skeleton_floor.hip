// tools/skeleton_floor.hip — what the data movement of each step kernel costs WITHOUT its arithmetic.
//
// The roofline fraction prices a launch against 8 TB/s.  This probe prices it against the kernel's own skeleton: the very same
// templates of csrc/step_kernels.hpp (loads, stores, stream masks, launch shape, done bookkeeping, the fused-reset code path) are
// instantiated with an env whose step() is one add per state word and whose done flag depends on the data but never fires, and
// timed beside the real env — same buffers, same launch configuration, same process, alternating so box noise hits both.
//   real - floor  = the part of a launch the arithmetic (and the resets that actually run) is NOT hidden under
//   floor / ideal = what this access pattern loses to ramp, drain and DRAM / Infinity-Cache efficiency at 2^20 lanes
// Not part of the product: built by tools/build_probes.sh into tools/build/ (modes: default = every env, "parts", "forms").
//
//   usage: skeleton_floor [lanes = 1048576] [launches = 2000] [rounds = 5]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include <string>
#include <chrono>

#include "../gym.net_amd/csrc/step_kernels.hpp"
#include "../gym.net_amd/csrc/envs.hpp"
#include "../gym.net_amd/csrc/cartpole64.hpp"

#define HIP_OK(x)                                                                                              \
    do {                                                                                                       \
        hipError_t e_ = (x);                                                                                   \
        if (e_ != hipSuccess) { std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); std::exit(2); } \
    } while (0)

namespace gymnet {

// Base's layout, action type, launch forms and reset — and no physics.
template <class Base>
struct DataOnly : Base {
    using Real = typename Base::Real;
    using Action = typename Base::Action;
    static constexpr int S = Base::S, O = Base::O;
    static constexpr bool HAS_SMALL_ANGLE_PATH = false;
    static constexpr bool PACKED2 = false;
    __device__ __forceinline__ static void step(Real (&s)[S], Action a, float &reward, bool &done) {
#pragma unroll
        for (int k = 0; k < S; ++k) s[k] = s[k] + (Real)a * (Real)1e-9f;
        reward = 1.0f;
        done = s[0] > (Real)1e30f;        // data-dependent, never true: the reset path stays compiled in and never runs
    }
    __device__ __forceinline__ static void observe(const Real (&s)[S], Real (&o)[O]) {
#pragma unroll
        for (int k = 0; k < O; ++k) o[k] = s[k % S];
    }
    __device__ __forceinline__ static void observe_fresh(const Real (&s)[S], Real (&o)[O]) { observe(s, o); }
    __device__ __forceinline__ static void step_observe(Real (&s)[S], Action a, float &reward, bool &done, Real (&o)[O]) {
        step(s, a, reward, done);
        observe(s, o);
    }
};

// Base's physics, and a reset that draws nothing (every episode starts at the same state): what the Philox passes cost.
template <class Base>
struct CheapReset : Base {
    using Real = typename Base::Real;
    static constexpr int S = Base::S;
    static constexpr bool RESET_TAKES_KEY = false;
    __device__ __forceinline__ static void reset(Real (&s)[S], const PhiloxWords &) {
#pragma unroll
        for (int k = 0; k < S; ++k) s[k] = (Real)0.01f * (Real)(k + 1);
    }
};

}  // namespace gymnet

using namespace gymnet;

// Probe form (NOT in the library): the multi-item kernel of step_kernel_pipe2 over items of VEC lanes — a thread owns ITEMS groups
// of VEC consecutive lanes (group k at thread index + k * T), every load first, then advance / store group after group.
template <class Env, int VEC, int ITEMS, int NT, int RESETF>
__global__ __launch_bounds__(256) void probe_multi(const StepArgsT<typename Env::Real> a) {
    ResetScratch<Env> *sc = nullptr;
    if constexpr (RESETF == 1) {
        __shared__ ResetScratch<Env> scratch[256 / 64];
        sc = &scratch[threadIdx.x >> 6];
    }
    const uint64_t tick = a.tick2[a.parity];
    if (blockIdx.x == 0 && threadIdx.x == 0) a.tick2[a.parity ^ 1] = tick + 1;
    const int64_t T = (int64_t)gridDim.x * blockDim.x;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    LaneInputs<Env, VEC> in[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) load_inputs<Env, VEC, true, NT, false>(a, (t + k * T) * VEC, in[k]);
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        if (k == 0) {
#pragma unroll
            for (int c = 0; c < Env::S; ++c)
#pragma unroll
                for (int j = 0; j < VEC; ++j) asm volatile("" : "+v"(in[0].s[c][j]));
        }
        advance_and_store<Env, VEC, true, false, NT, false, RESETF, false>(a, (t + k * T) * VEC, tick, in[k], sc);
    }
}

constexpr int64_t kRing = 32;        // action slices (iid per lane and slice: the bench's workload, ~4.5 % of CartPole lanes finish per step)
// Probe form (NOT in the library): step_kernel_pipe2's shape — ITEMS lane pairs per thread, all loads first — with the pair advanced by the env's
// PACKED two-lane arithmetic (v_pk_*_f32; the library's pair kernel advances lane after lane).  SQ counters say Acrobot's launch is VALU-issue
// bound (419 instructions per env-step x 4 cycles = 11.2 of its 12.6 us); the packed form is 287.
template <class Env, int ITEMS, int NT, bool PACK>
__global__ __launch_bounds__(256) void probe_pairs(const StepArgsT<typename Env::Real> a) {
    const uint64_t tick = a.tick2[a.parity];
    if (blockIdx.x == 0 && threadIdx.x == 0) a.tick2[a.parity ^ 1] = tick + 1;
    const int64_t T = (int64_t)gridDim.x * blockDim.x;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    LaneInputs<Env, 2> in[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) load_inputs<Env, 2, true, NT, false>(a, (t + k * T) * 2, in[k]);
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        if (k == 0) {
#pragma unroll
            for (int c = 0; c < Env::S; ++c) asm volatile("" : "+v"(in[0].s[c][0]), "+v"(in[0].s[c][1]));
        }
        advance_and_store<Env, 2, true, false, NT, false, 0, PACK>(a, (t + k * T) * 2, tick, in[k]);
    }
}

struct Buffers {
    void *state = nullptr, *obs = nullptr, *action = nullptr;
    float *reward = nullptr; uint8_t *done = nullptr; uint64_t *tick2 = nullptr;
};

template <class Env>
constexpr int32_t action_count() { if constexpr (Env::BOX_ACTION) return 0; else return Env::ACTION_N; }

// kRing slices of n actions: iid uniform over the Discrete(count) values, or uniform in [-1, 1) for a Box action (splitmix-style hash)
static std::vector<uint32_t> host_actions(int64_t n, bool box, int count) {
    std::vector<uint32_t> act((size_t)kRing * n);
    for (int64_t i = 0; i < kRing * n; ++i) {
        uint64_t z = (uint64_t)i * 0x9E3779B97F4A7C15ull + 0x5EED;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        if (box) { const float f = (float)(z >> 40) / 8388608.0f - 1.0f; std::memcpy(&act[i], &f, 4); }
        else act[i] = (uint32_t)((z >> 33) % (uint64_t)count);
    }
    return act;
}

template <class Env>
static StepArgsT<typename Env::Real> make_args(const Buffers &b, int64_t n) {
    using R = typename Env::Real;
    StepArgsT<R> a{};
    a.state = (R *)b.state; a.state_out = (R *)b.state;
    a.obs = Env::OBS_ALIASES_STATE ? (R *)b.state : (R *)b.obs;
    a.obs_in = a.obs;
    a.action = b.action; a.reward = b.reward; a.done = b.done; a.tick2 = b.tick2;
    a.n = n; a.state_stride = n; a.obs_stride = n; a.lane_offset = 0; a.seed = 0x5EED;
    return a;
}

template <class Env>
static double time_launches(const Buffers &b, int64_t n, LaunchCfg cfg, int launches, hipStream_t st, uint64_t &tick) {
    auto a = make_args<Env>(b, n);
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    HIP_OK(hipEventRecord(e0, st));
    for (int i = 0; i < launches; ++i) {
        a.parity = (int32_t)(tick & 1); a.cparity = a.parity;
        a.action = static_cast<const char *>(b.action) + (int64_t)(tick % kRing) * n * 4;
        HIP_OK((launch_step_env<Env>(true, false, a, cfg, st)));
        ++tick;
    }
    HIP_OK(hipEventRecord(e1, st));
    HIP_OK(hipEventSynchronize(e1));
    float ms = 0;
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    HIP_OK(hipEventDestroy(e0)); HIP_OK(hipEventDestroy(e1));
    return (double)ms * 1000.0 / launches;
}

template <class Env, int VEC, int ITEMS, int RESETF>
static double time_multi(const Buffers &b, int64_t n, int block, int launches, hipStream_t st, uint64_t &tick) {
    auto a = make_args<Env>(b, n);
    const dim3 grid((unsigned)(n / ((int64_t)VEC * ITEMS * block))), blk(block);
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    HIP_OK(hipEventRecord(e0, st));
    for (int i = 0; i < launches; ++i) {
        a.parity = (int32_t)(tick & 1); a.cparity = a.parity;
        a.action = static_cast<const char *>(b.action) + (int64_t)(tick % kRing) * n * 4;
        hipLaunchKernelGGL((probe_multi<Env, VEC, ITEMS, 15, RESETF>), grid, blk, 0, st, a);
        ++tick;
    }
    HIP_OK(hipGetLastError());
    HIP_OK(hipEventRecord(e1, st));
    HIP_OK(hipEventSynchronize(e1));
    float ms = 0;
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    HIP_OK(hipEventDestroy(e0)); HIP_OK(hipEventDestroy(e1));
    return (double)ms * 1000.0 / launches;
}

static double median(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

template <class Env>
static void run_env(const char *label, int moved_bytes, LaunchCfg cfg, int64_t n, int launches, int rounds, hipStream_t st, bool first) {
    using R = typename Env::Real;
    using Floor = DataOnly<Env>;
    Buffers b;
    const size_t esz = sizeof(R);
    HIP_OK(hipMalloc(&b.state, (size_t)Env::S * n * esz));
    HIP_OK(hipMalloc(&b.obs, (size_t)Env::O * n * esz));
    HIP_OK(hipMalloc(&b.action, (size_t)kRing * n * 4));
    HIP_OK(hipMalloc((void **)&b.reward, (size_t)n * 4));
    HIP_OK(hipMalloc((void **)&b.done, (size_t)n));
    HIP_OK(hipMalloc((void **)&b.tick2, 16));
    HIP_OK(hipMemsetAsync(b.state, 0, (size_t)Env::S * n * esz, st));
    HIP_OK(hipMemsetAsync(b.obs, 0, (size_t)Env::O * n * esz, st));
    std::vector<uint32_t> act = host_actions(n, Env::BOX_ACTION, Env::BOX_ACTION ? 0 : (int)action_count<Env>());
    HIP_OK(hipMemcpyAsync(b.action, act.data(), act.size() * 4, hipMemcpyHostToDevice, st));
    HIP_OK(hipMemsetAsync(b.tick2, 0, 16, st));
    HIP_OK(hipStreamSynchronize(st));
    uint64_t tick = 0;
    char name[160];
    describe_step_env<Env>(true, false, cfg, n, name, sizeof name);
    // warm-up of both, then alternate
    time_launches<Env>(b, n, cfg, 300, st, tick);
    time_launches<Floor>(b, n, cfg, 300, st, tick);
    std::vector<double> real, flo;
    for (int r = 0; r < rounds; ++r) {
        real.push_back(time_launches<Env>(b, n, cfg, launches, st, tick));
        flo.push_back(time_launches<Floor>(b, n, cfg, launches, st, tick));
    }
    const double mr = median(real), mf = median(flo);
    const double ideal = (double)moved_bytes * (double)n / 8.0e12 * 1e6;
    std::printf("%s{\"env\": \"%s\", \"kernel\": \"%s\", \"lanes\": %lld, \"moved_bytes_per_lane\": %d, \"ideal_us_at_8TBps\": %.3f, "
                "\"real_us\": %.3f, \"real_us_min\": %.3f, \"real_us_max\": %.3f, \"skeleton_us\": %.3f, \"skeleton_us_min\": %.3f, "
                "\"skeleton_us_max\": %.3f, \"real_over_skeleton\": %.4f, \"skeleton_frac_of_8TBps\": %.4f, \"real_frac_of_8TBps\": %.4f}",
                first ? "" : ",\n ", label, name, (long long)n, moved_bytes, ideal, mr, *std::min_element(real.begin(), real.end()),
                *std::max_element(real.begin(), real.end()), mf, *std::min_element(flo.begin(), flo.end()),
                *std::max_element(flo.begin(), flo.end()), mr / mf, ideal / mf, ideal / mr);
    std::fflush(stdout);
    HIP_OK(hipFree(b.state)); HIP_OK(hipFree(b.obs)); HIP_OK(hipFree(b.action)); HIP_OK(hipFree(b.reward)); HIP_OK(hipFree(b.done));
    HIP_OK(hipFree(b.tick2));
}

// multi-item forms of the dwordx4 kernels, real env and skeleton, against the library's one-shot default (same buffers)
template <class Env, int RESETF>
static void run_forms(const char *label, LaunchCfg dflt, int64_t n, int launches, int rounds, hipStream_t st) {
    using R = typename Env::Real;
    using Floor = DataOnly<Env>;
    Buffers b;
    const size_t esz = sizeof(R);
    HIP_OK(hipMalloc(&b.state, (size_t)Env::S * n * esz));
    HIP_OK(hipMalloc(&b.obs, (size_t)Env::O * n * esz));
    HIP_OK(hipMalloc(&b.action, (size_t)kRing * n * 4));
    HIP_OK(hipMalloc((void **)&b.reward, (size_t)n * 4));
    HIP_OK(hipMalloc((void **)&b.done, (size_t)n));
    HIP_OK(hipMalloc((void **)&b.tick2, 16));
    HIP_OK(hipMemsetAsync(b.state, 0, (size_t)Env::S * n * esz, st));
    HIP_OK(hipMemsetAsync(b.obs, 0, (size_t)Env::O * n * esz, st));
    std::vector<uint32_t> act = host_actions(n, Env::BOX_ACTION, Env::BOX_ACTION ? 0 : (int)action_count<Env>());
    HIP_OK(hipMemcpyAsync(b.action, act.data(), act.size() * 4, hipMemcpyHostToDevice, st));
    HIP_OK(hipMemsetAsync(b.tick2, 0, 16, st));
    HIP_OK(hipStreamSynchronize(st));
    uint64_t tick = 0;
    struct Row { const char *name; std::vector<double> real, flo; };
    std::vector<Row> rows;
    constexpr int V = 4;
#define FORM(NAME, REAL, FLOOR)                                                          \
    {                                                                                    \
        Row r{NAME, {}, {}};                                                             \
        (void)(REAL); (void)(FLOOR);                                                     \
        for (int q = 0; q < rounds; ++q) { r.real.push_back(REAL); r.flo.push_back(FLOOR); } \
        rows.push_back(r);                                                               \
    }
    FORM("one-shot (library default)", (time_launches<Env>(b, n, dflt, launches, st, tick)), (time_launches<Floor>(b, n, dflt, launches, st, tick)))
    FORM("1 quad, block 256 (probe kernel)", (time_multi<Env, V, 1, RESETF>(b, n, 256, launches, st, tick)), (time_multi<Floor, V, 1, RESETF>(b, n, 256, launches, st, tick)))
    FORM("2 quads, block 256", (time_multi<Env, V, 2, RESETF>(b, n, 256, launches, st, tick)), (time_multi<Floor, V, 2, RESETF>(b, n, 256, launches, st, tick)))
    FORM("2 quads, block 64", (time_multi<Env, V, 2, RESETF>(b, n, 64, launches, st, tick)), (time_multi<Floor, V, 2, RESETF>(b, n, 64, launches, st, tick)))
    FORM("3 quads, block 256 (n / 3072 groups)", (time_multi<Env, V, 3, RESETF>(b, n - n % 3072, 256, launches, st, tick)), (time_multi<Floor, V, 3, RESETF>(b, n - n % 3072, 256, launches, st, tick)))
    FORM("4 quads, block 256", (time_multi<Env, V, 4, RESETF>(b, n, 256, launches, st, tick)), (time_multi<Floor, V, 4, RESETF>(b, n, 256, launches, st, tick)))
    FORM("4 quads, block 64", (time_multi<Env, V, 4, RESETF>(b, n, 64, launches, st, tick)), (time_multi<Floor, V, 4, RESETF>(b, n, 64, launches, st, tick)))
    FORM("2 quads, block 256, drain-loop reset", (time_multi<Env, V, 2, 0>(b, n, 256, launches, st, tick)), (time_multi<Floor, V, 2, 0>(b, n, 256, launches, st, tick)))
    FORM("4 quads, block 256, drain-loop reset", (time_multi<Env, V, 4, 0>(b, n, 256, launches, st, tick)), (time_multi<Floor, V, 4, 0>(b, n, 256, launches, st, tick)))
#undef FORM
    std::printf("== %s, %lld lanes: us per launch, median of %d x %d launches (real | skeleton)\n", label, (long long)n, rounds, launches);
    for (auto &r : rows)
        std::printf("   %-42s real %7.3f [%.3f, %.3f]   skeleton %7.3f [%.3f, %.3f]\n", r.name, median(r.real),
                    *std::min_element(r.real.begin(), r.real.end()), *std::max_element(r.real.begin(), r.real.end()), median(r.flo),
                    *std::min_element(r.flo.begin(), r.flo.end()), *std::max_element(r.flo.begin(), r.flo.end()));
    std::fflush(stdout);
    HIP_OK(hipFree(b.state)); HIP_OK(hipFree(b.obs)); HIP_OK(hipFree(b.action)); HIP_OK(hipFree(b.reward)); HIP_OK(hipFree(b.done));
    HIP_OK(hipFree(b.tick2));
}

int main(int argc, char **argv) {
    const int64_t n = argc > 1 ? std::atoll(argv[1]) : (int64_t)1 << 20;
    const int launches = argc > 2 ? std::atoi(argv[2]) : 2000;
    const int rounds = argc > 3 ? std::atoi(argv[3]) : 5;
    HIP_OK(hipSetDevice(0));
    hipStream_t st;
    if (std::getenv("SKELETON_HIGH_PRIORITY")) {       // probe of the probe: does a high-priority stream shorten the launch-to-launch gap?
        int lo = 0, hi = 0;
        HIP_OK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        HIP_OK(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, hi));
        std::fprintf(stderr, "stream priority %d (range %d .. %d)\n", hi, lo, hi);
    } else HIP_OK(hipStreamCreate(&st));
    if (argc > 4 && std::strcmp(argv[4], "split") == 0) {
        // The step kernels are WRITE-bound at this size (profiles/write_path_probe_r02.txt: the write half of CartPole's pattern alone
        // takes 5.0 us, the read half 2.8, a copy of both 5.3) and a launch cannot store before its first loads are back.  Lanes are
        // independent, so the batch can run as K chains of n / K lanes on K streams: chain A's store phase then overlaps chain B's
        // load phase ACROSS launches.  us per vector step (all K launches), host clock around `launches` steps + one synchronize.
        const int parts_list[] = {1, 2, 3, 4};
        auto run_split = [&](auto env_tag, const char *label, LaunchCfg cfg) {
            using Env = decltype(env_tag);
            using R = typename Env::Real;
            Buffers b;
            HIP_OK(hipMalloc(&b.state, (size_t)Env::S * n * sizeof(R))); HIP_OK(hipMalloc(&b.obs, (size_t)Env::O * n * sizeof(R)));
            HIP_OK(hipMalloc(&b.action, (size_t)kRing * n * 4));
            HIP_OK(hipMalloc((void **)&b.reward, (size_t)n * 4)); HIP_OK(hipMalloc((void **)&b.done, (size_t)n));
            HIP_OK(hipMalloc((void **)&b.tick2, 16 * 8));
            HIP_OK(hipMemsetAsync(b.state, 0, (size_t)Env::S * n * sizeof(R), st)); HIP_OK(hipMemsetAsync(b.obs, 0, (size_t)Env::O * n * sizeof(R), st));
            std::vector<uint32_t> act = host_actions(n, Env::BOX_ACTION, Env::BOX_ACTION ? 0 : (int)action_count<Env>());
            HIP_OK(hipMemcpyAsync(b.action, act.data(), act.size() * 4, hipMemcpyHostToDevice, st));
            HIP_OK(hipStreamSynchronize(st));
            hipStream_t ss[4];
            for (auto &q : ss) HIP_OK(hipStreamCreateWithFlags(&q, hipStreamNonBlocking));
            std::printf("== %s, %lld lanes as K chains on K streams: us per vector step\n", label, (long long)n);
            for (int K : parts_list) {
                const int64_t m = n / K / 1024 * 1024;           // lanes per chain (whole workgroups; the remainder is ignored by the probe)
                std::vector<double> us;
                for (int r = 0; r < rounds + 1; ++r) {
                    HIP_OK(hipMemsetAsync(b.tick2, 0, 16 * 8, st));
                    HIP_OK(hipStreamSynchronize(st));
                    const auto t0 = std::chrono::steady_clock::now();
                    for (int l = 0; l < launches; ++l) {
                        for (int c = 0; c < K; ++c) {
                            StepArgsT<R> a{};
                            a.state = (R *)b.state + c * m; a.state_out = a.state;
                            a.obs = Env::OBS_ALIASES_STATE ? a.state : (R *)b.obs + c * m; a.obs_in = a.obs;
                            a.action = static_cast<const char *>(b.action) + ((int64_t)(l % kRing) * n + c * m) * 4;
                            a.reward = b.reward + c * m; a.done = b.done + c * m; a.tick2 = b.tick2 + 2 * c;
                            a.n = m; a.state_stride = n; a.obs_stride = n; a.lane_offset = (uint64_t)(c * m); a.seed = 0x5EED;
                            a.parity = l & 1; a.cparity = a.parity;
                            HIP_OK((launch_step_env<Env>(true, false, a, cfg, ss[c])));
                        }
                    }
                    for (int c = 0; c < K; ++c) HIP_OK(hipStreamSynchronize(ss[c]));
                    const auto t1 = std::chrono::steady_clock::now();
                    if (r) us.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count() / launches);
                }
                std::printf("   K = %d (%lld lanes per launch): eager %7.3f us per step  [%.3f, %.3f]", K, (long long)m, median(us),
                            *std::min_element(us.begin(), us.end()), *std::max_element(us.begin(), us.end()));
                // the same as ONE hipGraph of K parallel chains x 64 steps, replayed (no host cost per launch)
                {
                    constexpr int L = 64;
                    hipEvent_t fork, join[4];
                    HIP_OK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
                    for (auto &e : join) HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                    HIP_OK(hipMemsetAsync(b.tick2, 0, 16 * 8, st));
                    HIP_OK(hipStreamSynchronize(st));
                    hipGraph_t graph; hipGraphExec_t exec;
                    HIP_OK(hipStreamBeginCapture(ss[0], hipStreamCaptureModeGlobal));
                    HIP_OK(hipEventRecord(fork, ss[0]));
                    for (int c = 1; c < K; ++c) HIP_OK(hipStreamWaitEvent(ss[c], fork, 0));
                    for (int l = 0; l < L; ++l) {
                        for (int c = 0; c < K; ++c) {
                            StepArgsT<R> a{};
                            a.state = (R *)b.state + c * m; a.state_out = a.state;
                            a.obs = Env::OBS_ALIASES_STATE ? a.state : (R *)b.obs + c * m; a.obs_in = a.obs;
                            a.action = static_cast<const char *>(b.action) + ((int64_t)(l % kRing) * n + c * m) * 4;
                            a.reward = b.reward + c * m; a.done = b.done + c * m; a.tick2 = b.tick2 + 2 * c;
                            a.n = m; a.state_stride = n; a.obs_stride = n; a.lane_offset = (uint64_t)(c * m); a.seed = 0x5EED;
                            a.parity = l & 1; a.cparity = a.parity;
                            HIP_OK((launch_step_env<Env>(true, false, a, cfg, ss[c])));
                        }
                    }
                    for (int c = 1; c < K; ++c) { HIP_OK(hipEventRecord(join[c], ss[c])); HIP_OK(hipStreamWaitEvent(ss[0], join[c], 0)); }
                    HIP_OK(hipStreamEndCapture(ss[0], &graph));
                    HIP_OK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
                    std::vector<double> gus;
                    const int replays = launches / L > 0 ? launches / L : 1;
                    for (int r = 0; r < rounds + 1; ++r) {
                        const auto t0 = std::chrono::steady_clock::now();
                        for (int q = 0; q < replays; ++q) HIP_OK(hipGraphLaunch(exec, ss[0]));
                        HIP_OK(hipStreamSynchronize(ss[0]));
                        const auto t1 = std::chrono::steady_clock::now();
                        if (r) gus.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count() / (replays * L));
                    }
                    std::printf("   graph of K chains %7.3f us per step  [%.3f, %.3f]\n", median(gus), *std::min_element(gus.begin(), gus.end()),
                                *std::max_element(gus.begin(), gus.end()));
                    HIP_OK(hipGraphExecDestroy(exec)); HIP_OK(hipGraphDestroy(graph));
                    HIP_OK(hipEventDestroy(fork)); for (auto &e : join) HIP_OK(hipEventDestroy(e));
                }
                std::fflush(stdout);
            }
            for (auto &q : ss) HIP_OK(hipStreamDestroy(q));
            HIP_OK(hipFree(b.state)); HIP_OK(hipFree(b.obs)); HIP_OK(hipFree(b.action)); HIP_OK(hipFree(b.reward)); HIP_OK(hipFree(b.done)); HIP_OK(hipFree(b.tick2));
        };
        run_split(CartPole{}, "CartPole-v1 (float32), one-shot 16-byte lanes", LaunchCfg{4, 256, 15, 0, 1, 1, 0});
        run_split(MountainCar{}, "MountainCar-v0", LaunchCfg{4, 64, 15, 0, 1, 1, 0});
        run_split(Pendulum{}, "Pendulum-v1", LaunchCfg{4, 64, 15, 0, 1, 0, 0});
        run_split(CartPole64{}, "CartPole-v1 float64, one-shot", LaunchCfg{2, 256, 15, 0, 1, 1, 0});
        return 0;
    }
    if (argc > 4 && std::strcmp(argv[4], "acrobot_packed") == 0) {
        Buffers b;
        HIP_OK(hipMalloc(&b.state, (size_t)4 * n * 4)); HIP_OK(hipMalloc(&b.obs, (size_t)6 * n * 4)); HIP_OK(hipMalloc(&b.action, (size_t)kRing * n * 4));
        HIP_OK(hipMalloc((void **)&b.reward, (size_t)n * 4)); HIP_OK(hipMalloc((void **)&b.done, (size_t)n)); HIP_OK(hipMalloc((void **)&b.tick2, 16));
        HIP_OK(hipMemsetAsync(b.state, 0, (size_t)4 * n * 4, st)); HIP_OK(hipMemsetAsync(b.obs, 0, (size_t)6 * n * 4, st));
        std::vector<uint32_t> act = host_actions(n, false, 3);
        HIP_OK(hipMemcpyAsync(b.action, act.data(), act.size() * 4, hipMemcpyHostToDevice, st));
        HIP_OK(hipMemsetAsync(b.tick2, 0, 16, st));
        HIP_OK(hipStreamSynchronize(st));
        uint64_t tick = 0;
        auto time_pairs = [&](auto items_tag, auto pack_tag) {
            constexpr int I = decltype(items_tag)::value;
            constexpr bool P = decltype(pack_tag)::value;
            auto a = make_args<Acrobot>(b, n);
            const dim3 grid((unsigned)(n / (2 * I * 256))), blk(256);
            hipEvent_t e0, e1;
            HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
            HIP_OK(hipEventRecord(e0, st));
            for (int i = 0; i < launches; ++i) {
                a.parity = (int32_t)(tick & 1); a.cparity = a.parity;
                a.action = static_cast<const char *>(b.action) + (int64_t)(tick % kRing) * n * 4;
                hipLaunchKernelGGL((probe_pairs<Acrobot, I, 15, P>), grid, blk, 0, st, a);
                ++tick;
            }
            HIP_OK(hipEventRecord(e1, st)); HIP_OK(hipEventSynchronize(e1));
            float ms = 0; HIP_OK(hipEventElapsedTime(&ms, e0, e1));
            HIP_OK(hipEventDestroy(e0)); HIP_OK(hipEventDestroy(e1));
            return (double)ms * 1000.0 / launches;
        };
        std::vector<double> t[7];
        const LaunchCfg dflt{1, 256, 15, 0, 4, 0, 0};
        time_launches<Acrobot>(b, n, dflt, 300, st, tick);
        for (int q = 0; q < rounds; ++q) {
            t[0].push_back(time_launches<Acrobot>(b, n, dflt, launches, st, tick));
            t[1].push_back(time_pairs(std::integral_constant<int, 2>{}, std::false_type{}));
            t[2].push_back(time_pairs(std::integral_constant<int, 2>{}, std::true_type{}));
            t[3].push_back(time_pairs(std::integral_constant<int, 4>{}, std::false_type{}));
            t[4].push_back(time_pairs(std::integral_constant<int, 4>{}, std::true_type{}));
            t[5].push_back(time_pairs(std::integral_constant<int, 1>{}, std::true_type{}));
            t[6].push_back(time_pairs(std::integral_constant<int, 8>{}, std::true_type{}));
        }
        const char *names[] = {"library default: step_kernel_pipe<Acrobot,4> (4 scalar lanes per thread)", "2 pairs per thread, lane after lane", "2 pairs per thread, PACKED",
                               "4 pairs per thread, lane after lane", "4 pairs per thread, PACKED", "1 pair per thread, PACKED", "8 pairs per thread, PACKED"};
        std::printf("Acrobot, %lld lanes, us per launch:\n", (long long)n);
        for (int i = 0; i < 7; ++i) std::printf("   %-78s %7.3f  [%.3f, %.3f]\n", names[i], median(t[i]), *std::min_element(t[i].begin(), t[i].end()), *std::max_element(t[i].begin(), t[i].end()));
        return 0;
    }
    if (argc > 4 && std::strcmp(argv[4], "ntmask") == 0) {
        // every non-temporal mask of the shipped CartPole kernel (1 state loads, 2 state stores, 4 action load, 8 reward / done stores);
        // the launcher offers 0, 12 and 15 — tools/store_flavour_probe.hip suggested nt loads + PLAIN state stores (13) for the in-place copy
        Buffers b;
        HIP_OK(hipMalloc(&b.state, (size_t)4 * n * 4)); HIP_OK(hipMalloc(&b.obs, 64)); HIP_OK(hipMalloc(&b.action, (size_t)kRing * n * 4));
        HIP_OK(hipMalloc((void **)&b.reward, (size_t)n * 4)); HIP_OK(hipMalloc((void **)&b.done, (size_t)n)); HIP_OK(hipMalloc((void **)&b.tick2, 16));
        HIP_OK(hipMemsetAsync(b.state, 0, (size_t)4 * n * 4, st));
        std::vector<uint32_t> act = host_actions(n, false, 2);
        HIP_OK(hipMemcpyAsync(b.action, act.data(), act.size() * 4, hipMemcpyHostToDevice, st));
        HIP_OK(hipMemsetAsync(b.tick2, 0, 16, st));
        HIP_OK(hipStreamSynchronize(st));
        uint64_t tick = 0;
        const dim3 grid((unsigned)((n / 4 + 255) / 256)), blk(256);
        auto time_mask = [&](auto mask_tag) {
            constexpr int M = decltype(mask_tag)::value;
            auto a = make_args<CartPole>(b, n);
            hipEvent_t e0, e1;
            HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
            HIP_OK(hipEventRecord(e0, st));
            for (int i = 0; i < launches; ++i) {
                a.parity = (int32_t)(tick & 1); a.cparity = a.parity;
                a.action = static_cast<const char *>(b.action) + (int64_t)(tick % kRing) * n * 4;
                hipLaunchKernelGGL((step_kernel<CartPole, 4, true, false, M, 1>), grid, blk, 0, st, a);
                ++tick;
            }
            HIP_OK(hipEventRecord(e1, st)); HIP_OK(hipEventSynchronize(e1));
            float ms = 0; HIP_OK(hipEventElapsedTime(&ms, e0, e1));
            HIP_OK(hipEventDestroy(e0)); HIP_OK(hipEventDestroy(e1));
            return (double)ms * 1000.0 / launches;
        };
        std::vector<double> t[8];
        time_mask(std::integral_constant<int, 15>{});
        for (int q = 0; q < rounds; ++q) {
            t[0].push_back(time_mask(std::integral_constant<int, 15>{})); t[1].push_back(time_mask(std::integral_constant<int, 13>{}));
            t[2].push_back(time_mask(std::integral_constant<int, 12>{})); t[3].push_back(time_mask(std::integral_constant<int, 14>{}));
            t[4].push_back(time_mask(std::integral_constant<int, 9>{}));  t[5].push_back(time_mask(std::integral_constant<int, 5>{}));
            t[6].push_back(time_mask(std::integral_constant<int, 7>{}));  t[7].push_back(time_mask(std::integral_constant<int, 0>{}));
        }
        const int masks[] = {15, 13, 12, 14, 9, 5, 7, 0};
        std::printf("CartPole step kernel, %lld lanes, us per launch by non-temporal mask (1 state loads, 2 state stores, 4 action load, 8 reward / done stores):\n", (long long)n);
        for (int i = 0; i < 8; ++i) std::printf("   mask %2d   %7.3f  [%.3f, %.3f]\n", masks[i], median(t[i]), *std::min_element(t[i].begin(), t[i].end()), *std::max_element(t[i].begin(), t[i].end()));
        return 0;
    }
    if (argc > 4 && std::strcmp(argv[4], "parts") == 0) {
        // float64 CartPole: where the time above the skeleton goes — physics + Philox (real), physics only (constant reset), nothing
        Buffers b;
        HIP_OK(hipMalloc(&b.state, (size_t)4 * n * 8)); HIP_OK(hipMalloc(&b.obs, (size_t)4 * n * 8)); HIP_OK(hipMalloc(&b.action, (size_t)kRing * n * 4));
        HIP_OK(hipMalloc((void **)&b.reward, (size_t)n * 4)); HIP_OK(hipMalloc((void **)&b.done, (size_t)n)); HIP_OK(hipMalloc((void **)&b.tick2, 16));
        HIP_OK(hipMemsetAsync(b.state, 0, (size_t)4 * n * 8, st));
        std::vector<uint32_t> act = host_actions(n, false, 2);
        HIP_OK(hipMemcpyAsync(b.action, act.data(), act.size() * 4, hipMemcpyHostToDevice, st));
        HIP_OK(hipMemsetAsync(b.tick2, 0, 16, st));
        HIP_OK(hipStreamSynchronize(st));
        uint64_t tick = 0;
        for (int items : {4, 2, 1}) {
            const LaunchCfg cfg{2, 256, 15, 0, items, items == 1 ? 1 : 0, 0};
            std::vector<double> r, c, f;
            time_launches<CartPole64>(b, n, cfg, 300, st, tick);
            for (int q = 0; q < rounds; ++q) {
                r.push_back(time_launches<CartPole64>(b, n, cfg, launches, st, tick));
                c.push_back(time_launches<CheapReset<CartPole64>>(b, n, cfg, launches, st, tick));
                f.push_back(time_launches<DataOnly<CartPole64>>(b, n, cfg, launches, st, tick));
            }
            std::printf("CartPole64, %d pair(s) per thread: real %.3f   physics with a constant reset %.3f   skeleton %.3f us\n", items, median(r), median(c), median(f));
        }
        {
            HIP_OK(hipMemsetAsync(b.state, 0, (size_t)4 * n * 8, st));
            const LaunchCfg cfg{4, 256, 15, 0, 1, 1, 0};
            std::vector<double> r, c, f;
            time_launches<CartPole>(b, n, cfg, 300, st, tick);
            for (int q = 0; q < rounds; ++q) {
                r.push_back(time_launches<CartPole>(b, n, cfg, launches, st, tick));
                c.push_back(time_launches<CheapReset<CartPole>>(b, n, cfg, launches, st, tick));
                f.push_back(time_launches<DataOnly<CartPole>>(b, n, cfg, launches, st, tick));
            }
            std::printf("CartPole (float32), one-shot: real %.3f   physics with a constant reset %.3f   skeleton %.3f us\n", median(r), median(c), median(f));
        }
        return 0;
    }
    if (argc > 4 && std::strcmp(argv[4], "forms") == 0) {
        run_forms<CartPole, 1>("CartPole-v1", LaunchCfg{4, 256, 15, 0, 1, 1, 0}, n, launches, rounds, st);
        run_forms<MountainCar, 1>("MountainCar-v0", LaunchCfg{4, 64, 15, 0, 1, 1, 0}, n, launches, rounds, st);
        run_forms<Pendulum, 0>("Pendulum-v1", LaunchCfg{4, 64, 15, 0, 1, 0, 0}, n, launches, rounds, st);
        if (argc > 5) run_forms<Acrobot, 0>("Acrobot-v1 (16-byte lanes: not a library form)", LaunchCfg{1, 256, 15, 0, 4, 0, 0}, n, launches, rounds, st);
        return 0;
    }
    // launch configurations: the library's defaults at 2^20 lanes (capi.hip default_policy); bytes MOVED per lane and step as in
    // profiles/roofline_r05.json (Pendulum keeps theta_dot in the observation array: 33, Acrobot 57)
    std::printf("[");
    run_env<CartPole>("CartPole-v1", 41, LaunchCfg{4, 256, 15, 0, 1, 1, 0}, n, launches, rounds, st, true);
    run_env<CartPole64>("CartPole-v1-f64", 73, LaunchCfg{2, 256, 15, 0, 4, 0, 0}, n, launches, rounds, st, false);
    run_env<CartPole64>("CartPole-v1-f64 (one-shot)", 73, LaunchCfg{2, 256, 15, 0, 1, 1, 0}, n, launches, rounds, st, false);
    run_env<Pendulum>("Pendulum-v1", 33, LaunchCfg{4, 64, 15, 0, 1, 0, 0}, n, launches, rounds, st, false);
    run_env<MountainCar>("MountainCar-v0", 25, LaunchCfg{4, 64, 15, 0, 1, 1, 0}, n, launches, rounds, st, false);
    run_env<Acrobot>("Acrobot-v1", 57, LaunchCfg{1, 256, 15, 0, 4, 0, 0}, n, launches, rounds, st, false);
    std::printf("]\n");
    HIP_OK(hipStreamDestroy(st));
    return 0;
}
