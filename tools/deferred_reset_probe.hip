// tools/deferred_reset_probe.hip — float64 CartPole: the fused reset of the multi-pair kernel, drawn ONCE per group of pairs.
//
// tools/skeleton_floor.hip ("parts") shows where the float64 kernel's time above its data movement goes: not into the physics
// (constant reset: 12.1 us against a skeleton of 11.7) but into the Philox passes of the fused reset (real: 14.0 on the same box).
// step_kernel_pipe2 drains every pair's finished sub-lanes with the per-thread loop: ~1.9 trips per pair, two Philox calls (the
// float64 draw needs eight words) per trip, ~3 of 64 lanes active — 15 call-passes per wave, each 20 quarter-rate v_mad_u64_u32.
// Here the wave's finished sub-lanes of a GROUP of pairs are ranked (ballot + mbcnt), handed through LDS to the wave's lanes TWO
// LANES PER RESET (lane 2r draws the words of key, lane 2r + 1 those of key ^ kStreamReset64: one call-pass serves 32 resets),
// combined by a DPP exchange, converted by the even lane and handed back before the group's state rows are stored (whole rows,
// written once).  Same counters, same words, same conversions: bit-identical — the probe checks that against the library kernel.
//
//   usage: deferred_reset_probe [lanes = 1048576] [launches = 1500] [rounds = 5]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

#include "../gym.net_amd/csrc/step_kernels.hpp"
#include "../gym.net_amd/csrc/envs.hpp"
#include "../gym.net_amd/csrc/cartpole64.hpp"

#define HIP_OK(x)                                                                                              \
    do {                                                                                                       \
        hipError_t e_ = (x);                                                                                   \
        if (e_ != hipSuccess) { std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); std::exit(2); } \
    } while (0)

using namespace gymnet;

struct ProbeScratch {
    uint32_t slot[32];          // rank -> owner lane * 8 + (pair-in-group * 2 + sub-lane)
    double draw[32][4];         // rank -> the drawn state
};

template <class Env, int ITEMS, int GROUP, int NT>
__global__ __launch_bounds__(256) void pipe2_deferred(const StepArgsT<double> a) {
    static_assert(ITEMS % GROUP == 0 && GROUP * 2 <= 8, "");
    constexpr bool NT_SS = (NT & 2) != 0, NT_O = (NT & 8) != 0;
    __shared__ ProbeScratch scratch[256 / 64];
    ProbeScratch *sc = &scratch[threadIdx.x >> 6];
    const uint64_t tick = a.tick2[a.parity];
    if (blockIdx.x == 0 && threadIdx.x == 0) a.tick2[a.parity ^ 1] = tick + 1;
    const int64_t T = (int64_t)gridDim.x * blockDim.x;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = lane_id();
    const int64_t t_wave0 = t - (int64_t)lane;
    LaneInputs<Env, 2> in[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) load_inputs<Env, 2, true, NT, false>(a, (t + k * T) * 2, in[k]);
#pragma unroll
    for (int g = 0; g < ITEMS / GROUP; ++g) {
        uint32_t pending = 0;
#pragma unroll
        for (int kk = 0; kk < GROUP; ++kk) {
            const int k = g * GROUP + kk;
            if (k == 0) {
#pragma unroll
                for (int c = 0; c < Env::S; ++c) asm volatile("" : "+v"(in[0].s[c][0]), "+v"(in[0].s[c][1]));
            }
            const int64_t i0 = (t + k * T) * 2;
            double o[Env::O][2];
            float rw[2];
            bool dn[2], after[2] = {false, false};
            advance_all<Env, 2, true, false, false>(in[k].s, in[k].act, in[k].sbd, rw, dn, after, o, i0, a.n);
            uint8_t db[2] = {(uint8_t)(dn[0] ? 1 : 0), (uint8_t)(dn[1] ? 1 : 0)};
            store_f32<2, NT_O, false>(a.reward, i0, a.n, rw);
            store_u8<2, NT_O, false>(a.done, i0, a.n, db);
            pending |= (dn[0] ? 1u : 0u) << (kk * 2) | (dn[1] ? 1u : 0u) << (kk * 2 + 1);
        }
        // ---- one compacted reset for the group's 2 * GROUP sub-lane positions --------------------------------------------------
        constexpr int Q = 2 * GROUP;
        uint32_t rank[Q];
        uint32_t total = 0;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const uint64_t m = __ballot((pending >> q) & 1u);
            rank[q] = total + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            total += (uint32_t)__popcll(m);
        }
        for (uint32_t base = 0; base < total; base += 32) {            // wave-uniform; 32 resets per pass
#pragma unroll
            for (int q = 0; q < Q; ++q)
                if (((pending >> q) & 1u) && rank[q] - base < 32u) sc->slot[rank[q] - base] = lane * 8u + (uint32_t)q;
            wave_lds_fence();
            const uint32_t r = lane >> 1, c = lane & 1u;
            const bool draws = r < total - base;
            PhiloxWords w{};
            if (draws) {
                const uint32_t sl = sc->slot[r];
                const uint32_t owner = sl >> 3, q = sl & 7u;
                const int64_t gl = ((t_wave0 + owner) + (int64_t)(g * GROUP + (int)(q >> 1)) * T) * 2 + (q & 1u);
                w = lane_words(c ? (a.seed ^ kStreamReset64) : a.seed, a.lane_offset + (uint64_t)gl, tick);
            }
            // the even lane takes its odd neighbour's words (the second call's) and converts
            uint32_t other[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) other[k] = (uint32_t)__builtin_amdgcn_mov_dpp((int)w.w[k], 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true);
            if (draws && c == 0) {
#pragma unroll
                for (int k = 0; k < 4; ++k) sc->draw[r][k] = -0.05 + (0.05 - -0.05) * u01_53(w.w[k], other[k]);
            }
            wave_lds_fence();
            double got[Q][4];
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const uint32_t rr = rank[q] - base;
#pragma unroll
                for (int k = 0; k < 4; ++k) got[q][k] = sc->draw[rr < 32u ? rr : 0u][k];
            }
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const bool mine = ((pending >> q) & 1u) && rank[q] - base < 32u;
#pragma unroll
                for (int k = 0; k < 4; ++k) in[g * GROUP + q / 2].s[k][q % 2] = mine ? got[q][k] : in[g * GROUP + q / 2].s[k][q % 2];
            }
            wave_lds_fence();
        }
#pragma unroll
        for (int kk = 0; kk < GROUP; ++kk) {
            const int k = g * GROUP + kk;
            const int64_t i0 = (t + k * T) * 2;
#pragma unroll
            for (int row = 0; row < Env::S; ++row) store_row<double, 2, NT_SS, false>(a.state_out + row * a.state_stride, i0, a.n, in[k].s[row]);
        }
    }
}

// the kernel this form replaced: every pair drains its finished sub-lanes with the per-thread loop (two Philox calls per trip)
template <class Env, int ITEMS, int NT>
__global__ __launch_bounds__(256) void pipe2_drain(const StepArgsT<double> a) {
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
        advance_and_store<Env, 2, true, false, NT, false, 0, false>(a, (t + k * T) * 2, tick, in[k]);
    }
}

constexpr int64_t kRing = 64;
struct Buffers {
    double *state = nullptr; int32_t *action = nullptr; float *reward = nullptr; uint8_t *done = nullptr; uint64_t *tick2 = nullptr;
};

static Buffers make(int64_t n, hipStream_t st) {
    Buffers b;
    HIP_OK(hipMalloc((void **)&b.state, (size_t)4 * n * 8)); HIP_OK(hipMalloc((void **)&b.action, (size_t)kRing * n * 4));
    HIP_OK(hipMalloc((void **)&b.reward, (size_t)n * 4)); HIP_OK(hipMalloc((void **)&b.done, (size_t)n)); HIP_OK(hipMalloc((void **)&b.tick2, 16));
    HIP_OK(hipMemsetAsync(b.state, 0, (size_t)4 * n * 8, st));
    HIP_OK(hipMemsetAsync(b.tick2, 0, 16, st));
    // a ring of iid {0, 1} actions (splitmix-style hash of (slice, lane)): the bench's workload, ~4.5 % of the lanes finish per step
    std::vector<uint32_t> act((size_t)kRing * n);
    for (int64_t i = 0; i < kRing * n; ++i) {
        uint64_t z = (uint64_t)i * 0x9E3779B97F4A7C15ull + 0x5EED;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        act[i] = (uint32_t)((z >> 40) & 1u);
    }
    HIP_OK(hipMemcpyAsync(b.action, act.data(), act.size() * 4, hipMemcpyHostToDevice, st));
    HIP_OK(hipStreamSynchronize(st));
    return b;
}

static StepArgsT<double> args_of(const Buffers &b, int64_t n) {
    StepArgsT<double> a{};
    a.state = b.state; a.state_out = b.state; a.obs = b.state; a.obs_in = b.state;
    a.action = b.action; a.reward = b.reward; a.done = b.done; a.tick2 = b.tick2;
    a.n = n; a.state_stride = n; a.obs_stride = n; a.seed = 0x5EED;
    return a;
}

// form: 0 = the per-pair drain loop (the library's kernel until this probe); 1, 2, 4 = pipe2_deferred with that GROUP;
// 9 = the library's step_kernel_pipe2<CartPole64, 4> as it is now (one deferred reset for the thread's four pairs)
static void launch_form(int form, StepArgsT<double> &a, const void *ring0, uint64_t &tick, hipStream_t st) {
    a.parity = (int32_t)(tick & 1); a.cparity = a.parity;
    a.action = static_cast<const int32_t *>(ring0) + (int64_t)(tick % kRing) * a.n;
    const dim3 grid((unsigned)(a.n / (2 * 4 * 256))), blk(256);
    switch (form) {
        case 0: hipLaunchKernelGGL((pipe2_drain<CartPole64, 4, 15>), grid, blk, 0, st, a); break;
        case 9: HIP_OK((launch_step_env<CartPole64>(true, false, a, LaunchCfg{2, 256, 15, 0, 4, 0, 0}, st))); break;
        case 1: hipLaunchKernelGGL((pipe2_deferred<CartPole64, 4, 1, 15>), grid, blk, 0, st, a); break;
        case 2: hipLaunchKernelGGL((pipe2_deferred<CartPole64, 4, 2, 15>), grid, blk, 0, st, a); break;
        case 4: hipLaunchKernelGGL((pipe2_deferred<CartPole64, 4, 4, 15>), grid, blk, 0, st, a); break;
        default: std::exit(3);
    }
    ++tick;
}

static double time_form(int form, const Buffers &b, int64_t n, int launches, uint64_t &tick, hipStream_t st) {
    auto a = args_of(b, n);
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    HIP_OK(hipEventRecord(e0, st));
    for (int i = 0; i < launches; ++i) launch_form(form, a, b.action, tick, st);
    HIP_OK(hipGetLastError());
    HIP_OK(hipEventRecord(e1, st));
    HIP_OK(hipEventSynchronize(e1));
    float ms = 0;
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    HIP_OK(hipEventDestroy(e0)); HIP_OK(hipEventDestroy(e1));
    return (double)ms * 1000.0 / launches;
}

static double median(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main(int argc, char **argv) {
    const int64_t n = argc > 1 ? std::atoll(argv[1]) : (int64_t)1 << 20;
    const int launches = argc > 2 ? std::atoi(argv[2]) : 1500;
    const int rounds = argc > 3 ? std::atoi(argv[3]) : 5;
    HIP_OK(hipSetDevice(0));
    hipStream_t st;
    HIP_OK(hipStreamCreate(&st));
    const int forms[] = {0, 1, 2, 4, 9};
    const char *names[] = {"per-pair drain loop (round 4 / early round 5)", "deferred reset, group of 1 pair", "deferred reset, group of 2 pairs",
                           "deferred reset, group of 4 pairs", "library step_kernel_pipe2<CartPole64,4>"};
    constexpr int NF = 5;
    // ---- bit-identity: 300 steps from the same start (all-zero state: every lane falls and is reset several times) ---------------
    std::vector<std::vector<double>> finals;
    std::vector<std::vector<uint8_t>> dones;
    for (int f : forms) {
        Buffers b = make(n, st);
        auto a = args_of(b, n);
        uint64_t tick = 0;
        for (int i = 0; i < 300; ++i) launch_form(f, a, b.action, tick, st);
        HIP_OK(hipStreamSynchronize(st));
        std::vector<double> h((size_t)4 * n);
        std::vector<uint8_t> d((size_t)n);
        HIP_OK(hipMemcpy(h.data(), b.state, h.size() * 8, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(d.data(), b.done, d.size(), hipMemcpyDeviceToHost));
        finals.push_back(h); dones.push_back(d);
        HIP_OK(hipFree(b.state)); HIP_OK(hipFree(b.action)); HIP_OK(hipFree(b.reward)); HIP_OK(hipFree(b.done)); HIP_OK(hipFree(b.tick2));
    }
    bool same = true;
    for (size_t f = 1; f < finals.size(); ++f) {
        const bool eq = std::memcmp(finals[0].data(), finals[f].data(), finals[0].size() * 8) == 0 && dones[0] == dones[f];
        std::printf("%s vs the drain loop after 300 steps: %s\n", names[f], eq ? "bit-identical" : "DIFFERENT");
        same = same && eq;
    }
    size_t nd = 0; for (uint8_t d : dones[0]) nd += d;
    std::printf("(lanes done in the last step: %zu of %lld)\n", nd, (long long)n);
    // ---- timing ------------------------------------------------------------------------------------------------------------------
    Buffers b = make(n, st);
    uint64_t tick = 0;
    for (int f : forms) time_form(f, b, n, 300, tick, st);
    std::vector<std::vector<double>> tm(NF);
    for (int q = 0; q < rounds; ++q)
        for (int fi = 0; fi < NF; ++fi) tm[fi].push_back(time_form(forms[fi], b, n, launches, tick, st));
    for (int fi = 0; fi < NF; ++fi)
        std::printf("%-46s %7.3f us per launch  [%.3f, %.3f]\n", names[fi], median(tm[fi]), *std::min_element(tm[fi].begin(), tm[fi].end()), *std::max_element(tm[fi].begin(), tm[fi].end()));
    return same ? 0 : 1;
}
