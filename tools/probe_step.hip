// tools/probe_step.hip — within-process A/B probe for the CartPole step kernel (NOT part of the product).
//
// Builds variants of the hot kernel from the same dynamics (gym.net_amd/csrc/envs.hpp) and times them
// interleaved, round-robin, with HIP events on one stream, so that deltas are not cross-process noise
// (cdna_hip_programming.md §5.4 rule 24).  Variants answer "where do the 10 us at 2^20 lanes go?":
//   copy-only (memory floor for the 41 B/lane access pattern), no-reset (Philox cost), VEC / block /
//   non-temporal / software-pipelined chunks.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/probe_step.hip -o tools/probe_step
//   ./tools/probe_step [log2_lanes=20] [rounds=20] [steps_per_round=200]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../gym.net_amd/csrc/envs.hpp"
#include "../gym.net_amd/csrc/kernels.hpp"

using namespace gymnet;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Args {
    float *s0, *s1, *s2, *s3;
    const int32_t *action;
    float *reward;
    uint8_t *done;
    const uint64_t *tick;
    int64_t n;
    uint64_t seed;
};

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int32_t i32x2 __attribute__((ext_vector_type(2)));
typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
template <int VEC> struct VT;
template <> struct VT<1> { using f = float; using i = int32_t; };
template <> struct VT<2> { using f = f32x2; using i = i32x2; };
template <> struct VT<4> { using f = f32x4; using i = i32x4; };

template <int VEC, bool NT>
__device__ __forceinline__ void ldf(const float *p, int64_t i0, float (&v)[VEC]) {
    using T = typename VT<VEC>::f;
    T t;
    if constexpr (NT) t = __builtin_nontemporal_load(reinterpret_cast<const T *>(p + i0));
    else t = *reinterpret_cast<const T *>(p + i0);
    const float *q = reinterpret_cast<const float *>(&t);
#pragma unroll
    for (int j = 0; j < VEC; ++j) v[j] = q[j];
}
template <int VEC, bool NT>
__device__ __forceinline__ void stf(float *p, int64_t i0, const float (&v)[VEC]) {
    using T = typename VT<VEC>::f;
    T t;
    float *q = reinterpret_cast<float *>(&t);
#pragma unroll
    for (int j = 0; j < VEC; ++j) q[j] = v[j];
    if constexpr (NT) __builtin_nontemporal_store(t, reinterpret_cast<T *>(p + i0));
    else *reinterpret_cast<T *>(p + i0) = t;
}
template <int VEC, bool NT>
__device__ __forceinline__ void ldi(const int32_t *p, int64_t i0, int32_t (&v)[VEC]) {
    using T = typename VT<VEC>::i;
    T t;
    if constexpr (NT) t = __builtin_nontemporal_load(reinterpret_cast<const T *>(p + i0));
    else t = *reinterpret_cast<const T *>(p + i0);
    const int32_t *q = reinterpret_cast<const int32_t *>(&t);
#pragma unroll
    for (int j = 0; j < VEC; ++j) v[j] = q[j];
}
template <int VEC, bool NT>
__device__ __forceinline__ void stu8(uint8_t *p, int64_t i0, const uint8_t (&v)[VEC]) {
    if constexpr (VEC == 4) {
        uint32_t w = (uint32_t)v[0] | ((uint32_t)v[1] << 8) | ((uint32_t)v[2] << 16) | ((uint32_t)v[3] << 24);
        if constexpr (NT) __builtin_nontemporal_store(w, reinterpret_cast<uint32_t *>(p + i0));
        else *reinterpret_cast<uint32_t *>(p + i0) = w;
    } else if constexpr (VEC == 2) {
        uint16_t w = (uint16_t)((uint16_t)v[0] | ((uint16_t)v[1] << 8));
        *reinterpret_cast<uint16_t *>(p + i0) = w;
    } else {
        p[i0] = v[0];
    }
}

// MODE 0 = full step; 1 = copy only (same loads/stores, trivial math); 2 = full math but no auto-reset
// RESET 0 = divergent per-sub-lane Philox (baseline); 1 = compacted loop: one Philox per iteration per lane
// NT mask: 1 = state loads, 2 = state stores, 4 = action load, 8 = reward/done stores
// RESET 2 = block-level compaction through LDS: the block's finished (lane, sub-lane) pairs are appended to an
//           LDS list (wave ballot + one LDS atomic per wave), the first cnt threads evaluate ONE Philox each,
//           results go back through LDS.  ~46 of 1024 envs finish per step => one wave-pass per block.
template <int VEC, int BLOCK, int MODE, int RESET, int NT>
__global__ __launch_bounds__(BLOCK) void k_step(const Args a) {
    constexpr bool NT_SL = NT & 1, NT_SS = NT & 2, NT_A = NT & 4, NT_O = NT & 8;
    const int64_t i0 = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) * VEC;
    if (RESET != 2 && i0 >= a.n) return;
    const uint64_t tick = a.tick[0];
    float s[4][VEC];
    ldf<VEC, NT_SL>(a.s0, i0, s[0]);
    ldf<VEC, NT_SL>(a.s1, i0, s[1]);
    ldf<VEC, NT_SL>(a.s2, i0, s[2]);
    ldf<VEC, NT_SL>(a.s3, i0, s[3]);
    int32_t act[VEC];
    ldi<VEC, NT_A>(a.action, i0, act);
    float reward[VEC];
    uint8_t done[VEC];
    uint32_t pending = 0;
    // RESET 3 = speculative: every sub-lane's reset draw is computed while the loads are still in flight
    float spec[VEC][4];
    if constexpr (MODE == 0 && RESET == 3) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const PhiloxWords r = lane_words(a.seed, (uint64_t)(i0 + j), tick);
            CartPole::reset(spec[j], r);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        if constexpr (MODE == 1) {
            reward[j] = 1.0f; done[j] = (uint8_t)(act[j] & 1);
            s[0][j] += 1.0f;
        } else {
            float sj[4] = {s[0][j], s[1][j], s[2][j], s[3][j]};
            bool dn; float rw;
            CartPole::step(sj, act[j], rw, dn);
            if constexpr (MODE >= 3) { CartPole::step(sj, act[j], rw, dn); }      // ALU-scaling experiment: 2x physics
            if constexpr (MODE >= 4) { CartPole::step(sj, act[j], rw, dn); }      // 3x physics
            reward[j] = rw; done[j] = dn ? 1 : 0;
            if constexpr (MODE == 0 && RESET == 0) {
                if (dn) {
                    const PhiloxWords r = lane_words(a.seed, (uint64_t)(i0 + j), tick);
                    CartPole::reset(sj, r);
                }
            }
            if constexpr (MODE == 0 && RESET == 1) pending |= dn ? (1u << j) : 0u;
            if constexpr (MODE == 0 && RESET == 3) {
                sj[0] = dn ? spec[j][0] : sj[0]; sj[1] = dn ? spec[j][1] : sj[1];
                sj[2] = dn ? spec[j][2] : sj[2]; sj[3] = dn ? spec[j][3] : sj[3];
            }
            s[0][j] = sj[0]; s[1][j] = sj[1]; s[2][j] = sj[2]; s[3][j] = sj[3];
        }
    }
    if constexpr (MODE == 0 && RESET == 1) {
        // one Philox evaluation per loop trip per lane: trips per wave = max over lanes of #finished sub-lanes
        while (pending) {
            const int j = __ffs(pending) - 1;
            pending &= pending - 1;
            const PhiloxWords r = lane_words(a.seed, (uint64_t)(i0 + j), tick);
            float sj[4];
            CartPole::reset(sj, r);
#pragma unroll
            for (int jj = 0; jj < VEC; ++jj)
                if (jj == j) { s[0][jj] = sj[0]; s[1][jj] = sj[1]; s[2][jj] = sj[2]; s[3][jj] = sj[3]; }
        }
    }
    if constexpr (MODE == 0 && RESET == 2) {
        __shared__ uint32_t s_cnt;
        __shared__ uint32_t s_pair[BLOCK * VEC];
        __shared__ float s_val[BLOCK * VEC][4];
        if (threadIdx.x == 0) s_cnt = 0;
        __syncthreads();
        const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        const uint64_t below = (1ull << lane) - 1ull;
        uint32_t off[VEC], total = 0;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const uint64_t m = __ballot(done[j] != 0);
            off[j] = total + (uint32_t)__popcll(m & below);
            total += (uint32_t)__popcll(m);
        }
        uint32_t base = 0;
        if (total) {
            if (lane == 0) base = atomicAdd(&s_cnt, total);
            base = __shfl(base, 0);
#pragma unroll
            for (int j = 0; j < VEC; ++j)
                if (done[j]) s_pair[base + off[j]] = threadIdx.x * VEC + j;
        }
        __syncthreads();
        const uint32_t cnt = s_cnt;
        for (uint32_t k = threadIdx.x; k < cnt; k += BLOCK) {
            const uint32_t pr = s_pair[k];
            const PhiloxWords r = lane_words(a.seed, (uint64_t)blockIdx.x * BLOCK * VEC + pr, tick);
            float sj[4];
            CartPole::reset(sj, r);
            s_val[k][0] = sj[0]; s_val[k][1] = sj[1]; s_val[k][2] = sj[2]; s_val[k][3] = sj[3];
        }
        __syncthreads();
        if (total) {
#pragma unroll
            for (int j = 0; j < VEC; ++j)
                if (done[j]) {
                    const uint32_t k = base + off[j];
                    s[0][j] = s_val[k][0]; s[1][j] = s_val[k][1]; s[2][j] = s_val[k][2]; s[3][j] = s_val[k][3];
                }
        }
    }
    stf<VEC, NT_SS>(a.s0, i0, s[0]);
    stf<VEC, NT_SS>(a.s1, i0, s[1]);
    stf<VEC, NT_SS>(a.s2, i0, s[2]);
    stf<VEC, NT_SS>(a.s3, i0, s[3]);
    stf<VEC, NT_O>(a.reward, i0, reward);
    stu8<VEC, NT_O>(a.done, i0, done);
}

// software-pipelined persistent variant: each thread walks CH chunks, loading chunk c+1 before computing chunk c
template <int BLOCK, int CH, int RESET>
__global__ __launch_bounds__(BLOCK) void k_step_pipe(const Args a) {
    constexpr int VEC = 4;
    const int64_t nthreads = (int64_t)gridDim.x * BLOCK;
    const int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const uint64_t tick = a.tick[0];
    float s[2][4][VEC];
    int32_t act[2][VEC];
    auto load = [&](int b, int64_t i0) {
        ldf<VEC, false>(a.s0, i0, s[b][0]); ldf<VEC, false>(a.s1, i0, s[b][1]);
        ldf<VEC, false>(a.s2, i0, s[b][2]); ldf<VEC, false>(a.s3, i0, s[b][3]);
        ldi<VEC, false>(a.action, i0, act[b]);
    };
    load(0, t * VEC);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int b = c & 1;
        const int64_t i0 = (t + (int64_t)c * nthreads) * VEC;
        if (c + 1 < CH) load(b ^ 1, (t + (int64_t)(c + 1) * nthreads) * VEC);
        float reward[VEC]; uint8_t done[VEC];
        uint32_t pending = 0;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            float sj[4] = {s[b][0][j], s[b][1][j], s[b][2][j], s[b][3][j]};
            bool dn; float rw;
            CartPole::step(sj, act[b][j], rw, dn);
            reward[j] = rw; done[j] = dn ? 1 : 0;
            if constexpr (RESET == 0) {
                if (dn) { const PhiloxWords r = lane_words(a.seed, (uint64_t)(i0 + j), tick); CartPole::reset(sj, r); }
            } else pending |= dn ? (1u << j) : 0u;
            s[b][0][j] = sj[0]; s[b][1][j] = sj[1]; s[b][2][j] = sj[2]; s[b][3][j] = sj[3];
        }
        if constexpr (RESET == 1) {
            while (pending) {
                const int j = __ffs(pending) - 1;
                pending &= pending - 1;
                const PhiloxWords r = lane_words(a.seed, (uint64_t)(i0 + j), tick);
                float sj[4];
                CartPole::reset(sj, r);
#pragma unroll
                for (int jj = 0; jj < VEC; ++jj)
                    if (jj == j) { s[b][0][jj] = sj[0]; s[b][1][jj] = sj[1]; s[b][2][jj] = sj[2]; s[b][3][jj] = sj[3]; }
            }
        }
        stf<VEC, false>(a.s0, i0, s[b][0]); stf<VEC, false>(a.s1, i0, s[b][1]);
        stf<VEC, false>(a.s2, i0, s[b][2]); stf<VEC, false>(a.s3, i0, s[b][3]);
        stf<VEC, false>(a.reward, i0, reward);
        stu8<VEC, false>(a.done, i0, done);
    }
}

__global__ void k_init(float *s0, float *s1, float *s2, float *s3, int64_t n, uint64_t seed) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const PhiloxWords r = lane_words(seed, (uint64_t)i, 0);
    float s[4];
    CartPole::reset(s, r);
    s0[i] = s[0]; s1[i] = s[1]; s2[i] = s[2]; s3[i] = s[3];
}
__global__ void k_actions(int32_t *a, int64_t n, uint64_t seed, uint64_t tick) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    a[i] = (int32_t)(lane_words(seed, (uint64_t)i, tick).w[0] >> 31);
}

struct Variant {
    std::string name;
    void (*launch)(const Args &, hipStream_t);
    std::vector<float> us;
};

// occupancy-limited launch: LDSKB KiB of (unused) dynamic LDS per workgroup caps the resident workgroups per CU, so
// the grid runs in several GENERATIONS and one generation's store phase can overlap the next one's load phase
template <int VEC, int BLOCK, int MODE, int RESET, int NT, int LDSKB>
void LOCC(const Args &a, hipStream_t st) {
    const int64_t threads = a.n / VEC;
    hipLaunchKernelGGL((k_step<VEC, BLOCK, MODE, RESET, NT>), dim3((unsigned)((threads + BLOCK - 1) / BLOCK)), dim3(BLOCK), LDSKB * 1024, st, a);
}

template <int VEC, int BLOCK, int MODE, int RESET, int NT>
void L(const Args &a, hipStream_t st) {
    const int64_t threads = a.n / VEC;
    hipLaunchKernelGGL((k_step<VEC, BLOCK, MODE, RESET, NT>), dim3((unsigned)((threads + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, st, a);
}
template <int BLOCK, int CH, int RESET>
void LP(const Args &a, hipStream_t st) {
    const int64_t threads = a.n / 4 / CH;
    hipLaunchKernelGGL((k_step_pipe<BLOCK, CH, RESET>), dim3((unsigned)((threads + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, st, a);
}

// the SHIPPED kernel (gym.net_amd/csrc/kernels.hip), launched through its own launcher, on the same buffers
static uint64_t *g_tick2 = nullptr;
static uint64_t g_host_tick = 0;
template <int VEC, int NT>
void LPROD(const Args &a, hipStream_t st) {
    StepArgs s{};
    s.state = a.s0; s.state_out = a.s0; s.obs = a.s0; s.action = a.action; s.reward = a.reward; s.done = a.done;
    s.tick2 = g_tick2; s.n = a.n; s.state_stride = a.s1 - a.s0; s.obs_stride = s.state_stride;
    s.seed = a.seed; s.parity = (int32_t)(g_host_tick & 1); s.cparity = s.parity;
    ++g_host_tick;
    launch_step(0, true, false, s, LaunchCfg{VEC, 256, NT}, st);
}

int main(int argc, char **argv) {
    const int lg = argc > 1 ? atoi(argv[1]) : 20;
    const int rounds = argc > 2 ? atoi(argv[2]) : 20;
    const int steps = argc > 3 ? atoi(argv[3]) : 200;
    const int64_t n = (int64_t)1 << lg;
    const int ring = argc > 4 ? atoi(argv[4]) : (lg <= 20 ? 64 : (lg <= 22 ? 16 : 4));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    Args a{};
    CK(hipMalloc(&a.s0, n * 16)); a.s1 = a.s0 + n; a.s2 = a.s1 + n; a.s3 = a.s2 + n;   // one SoA block, like the product
    CK(hipMalloc(&g_tick2, 16)); CK(hipMemset(g_tick2, 0, 16));
    CK(hipMalloc(&a.reward, n * 4)); CK(hipMalloc(&a.done, n));
    int32_t *acts; CK(hipMalloc(&acts, (size_t)ring * n * 4));
    uint64_t *tick; CK(hipMalloc(&tick, 8)); CK(hipMemset(tick, 0, 8));
    a.tick = tick; a.n = n; a.seed = 0x5EED;
    for (int r = 0; r < ring; ++r) hipLaunchKernelGGL(k_actions, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, acts + (size_t)r * n, n, 77, (uint64_t)r);

    std::vector<Variant> V;
#define ADD(nm, fn) V.push_back(Variant{nm, fn, {}})
    ADD("base    vec4 b256 reset=divergent ", (L<4, 256, 0, 0, 0>));
    ADD("loop    vec4 b256                 ", (L<4, 256, 0, 1, 0>));
    ADD("SHIPPED vec4 nt=all               ", (LPROD<4, 15>));
    ADD("loop vec4 b256 nt=all lds 40K (4/CU)", (LOCC<4, 256, 0, 1, 15, 40>));
    ADD("loop vec4 b256 nt=all lds 53K (3/CU)", (LOCC<4, 256, 0, 1, 15, 53>));
    ADD("loop vec4 b256 nt=all lds 80K (2/CU)", (LOCC<4, 256, 0, 1, 15, 80>));
    ADD("loop vec4 b256 nt=all lds 100K(1/CU)", (LOCC<4, 256, 0, 1, 15, 100>));
    ADD("loop vec4 b128 nt=all lds 40K (4/CU)", (LOCC<4, 128, 0, 1, 15, 40>));
    ADD("loop vec4 b128 nt=all lds 26K (6/CU)", (LOCC<4, 128, 0, 1, 15, 26>));
    ADD("loop vec4 b64  nt=all lds 20K (8/CU)", (LOCC<4, 64, 0, 1, 15, 20>));
    ADD("loop vec4 b64  nt=all lds 13K (12/CU)", (LOCC<4, 64, 0, 1, 15, 13>));
    ADD("loop vec2 b256 nt=all lds 40K (4/CU)", (LOCC<2, 256, 0, 1, 15, 40>));
    ADD("loop vec2 b256 nt=all lds 26K (6/CU)", (LOCC<2, 256, 0, 1, 15, 26>));
    ADD("loop vec1 b256 nt=all lds 26K (6/CU)", (LOCC<1, 256, 0, 1, 15, 26>));
    ADD("loop vec1 b256 nt=all lds 40K (4/CU)", (LOCC<1, 256, 0, 1, 15, 40>));
    ADD("copy vec4 b256 lds 80K (2/CU)       ", (LOCC<4, 256, 1, 0, 0, 80>));
    ADD("copy vec4 b128 lds 40K (4/CU)       ", (LOCC<4, 128, 1, 0, 0, 40>));
    ADD("noreset vec4 b256 nt=all 2x phys  ", (L<4, 256, 3, 0, 15>));
    ADD("noreset vec4 b256 nt=all 3x phys  ", (L<4, 256, 4, 0, 15>));
    ADD("spec    vec4 b256 nt=all          ", (L<4, 256, 0, 3, 15>));
    ADD("spec    vec2 b256 nt=all          ", (L<2, 256, 0, 3, 15>));
    ADD("spec    vec4 b256 nt=none         ", (L<4, 256, 0, 3, 0>));
    ADD("SHIPPED vec4 nt=streams           ", (LPROD<4, 12>));
    ADD("SHIPPED vec4 nt=none              ", (LPROD<4, 0>));
    ADD("SHIPPED vec1 nt=all               ", (LPROD<1, 15>));
    ADD("copy    vec4 b256 (memory floor)  ", (L<4, 256, 1, 0, 0>));
    ADD("copy    vec4 b256 nt=all-stores   ", (L<4, 256, 1, 0, 10>));
    ADD("noreset vec4 b256 nt=all          ", (L<4, 256, 2, 0, 15>));
    ADD("loop    vec4 b256 nt=state-ld+st  ", (L<4, 256, 0, 1, 3>));
    ADD("loop    vec4 b256 nt=all-stores   ", (L<4, 256, 0, 1, 10>));
    ADD("loop    vec4 b256 nt=streams      ", (L<4, 256, 0, 1, 12>));
    ADD("loop    vec4 b256 nt=st+streams   ", (L<4, 256, 0, 1, 14>));
    ADD("loop    vec4 b256 nt=all          ", (L<4, 256, 0, 1, 15>));
    ADD("loop    vec4 b128 nt=all          ", (L<4, 128, 0, 1, 15>));
    ADD("loop    vec2 b256 nt=all          ", (L<2, 256, 0, 1, 15>));
    ADD("loop    vec1 b256 nt=all          ", (L<1, 256, 0, 1, 15>));
    ADD("loop    vec1 b256 nt=streams      ", (L<1, 256, 0, 1, 12>));
    ADD("loop    vec2 b256 nt=streams      ", (L<2, 256, 0, 1, 12>));
    ADD("lds     vec4 b256 nt=all          ", (L<4, 256, 0, 2, 15>));
    ADD("lds     vec4 b256 nt=streams      ", (L<4, 256, 0, 2, 12>));

    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto reinit = [&]() { hipLaunchKernelGGL(k_init, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a.s0, a.s1, a.s2, a.s3, n, a.seed); };
    for (int r = 0; r < rounds + 1; ++r) {
        for (auto &v : V) {
            reinit();
            // 30 untimed steps bring the population to its steady-state done rate (~4.5 %/step)
            for (int t = 0; t < 30; ++t) { a.action = acts + (size_t)(t % ring) * n; v.launch(a, st); }
            CK(hipEventRecord(e0, st));
            for (int t = 0; t < steps; ++t) { a.action = acts + (size_t)(t % ring) * n; v.launch(a, st); }
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r > 0) v.us.push_back(ms * 1e3f / steps);
        }
    }
    CK(hipGetLastError());
    // ---- experiment: the batch split in K parts, each part on its own stream (load phase of one part can overlap
    //      the store phase of another); time per FULL vector step
    for (int parts : {1, 2, 4}) {
        std::vector<hipStream_t> ss(parts);
        for (auto &s : ss) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        std::vector<float> us;
        for (int r = 0; r < rounds; ++r) {
            reinit();
            CK(hipStreamSynchronize(st));
            hipEvent_t b0, b1; CK(hipEventCreate(&b0)); CK(hipEventCreate(&b1));
            CK(hipEventRecord(b0, ss[0]));
            for (int p = 1; p < parts; ++p) CK(hipStreamWaitEvent(ss[p], b0, 0));
            const int64_t np = n / parts;
            for (int t = 0; t < steps; ++t)
                for (int p = 0; p < parts; ++p) {
                    StepArgs s{};
                    s.state = a.s0 + p * np; s.obs = s.state; s.action = acts + (size_t)(t % ring) * n + p * np;
                    s.reward = a.reward + p * np; s.done = a.done + p * np;
                    s.tick2 = g_tick2; s.n = np; s.state_stride = n; s.obs_stride = n; s.lane_offset = (uint64_t)(p * np);
                    s.seed = a.seed; s.parity = 0; s.cparity = 0;
                    launch_step(0, true, false, s, LaunchCfg{4, 256, 15}, ss[p]);
                }
            std::vector<hipEvent_t> ends(parts);
            for (int p = 0; p < parts; ++p) { CK(hipEventCreate(&ends[p])); CK(hipEventRecord(ends[p], ss[p])); }
            for (int p = 1; p < parts; ++p) CK(hipStreamWaitEvent(ss[0], ends[p], 0));
            CK(hipEventRecord(b1, ss[0]));
            CK(hipEventSynchronize(b1));
            float ms; CK(hipEventElapsedTime(&ms, b0, b1));
            us.push_back(ms * 1e3f / steps);
        }
        std::sort(us.begin(), us.end());
        printf("SHIPPED split over %d stream(s): median %8.3f us per full step  %8.1f GB/s\n", parts, us[us.size() / 2], 41.0 * n / (us[us.size() / 2] * 1e-6) / 1e9);
    }
    printf("lanes=2^%d rounds=%d steps/round=%d   (us per launch; GB/s = 41 B x lanes / median)\n", lg, rounds, steps);
    for (auto &v : V) {
        std::sort(v.us.begin(), v.us.end());
        const float med = v.us[v.us.size() / 2], mn = v.us.front();
        printf("%-36s median %8.3f us  min %8.3f us  %8.1f GB/s\n", v.name.c_str(), med, mn, 41.0 * n / (med * 1e-6) / 1e9);
    }
    return 0;
}
