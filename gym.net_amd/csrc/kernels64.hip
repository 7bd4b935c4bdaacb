// kernels64.hip — GYMNET_FLAG_F64: the CartPole step / reset / export kernels over binary64 structure-of-arrays state.
//
// Same design rules as kernels.hip (one env per lane, SoA rows, 16-byte accesses where the row allows, static block -> lane map,
// one ballot + one atomic per wave for the step-after-done counter), for the path whose ARITHMETIC is the reference's own
// (cartpole64.hpp): 73 B per env-step against ~45 binary64 operations, four of them IEEE divisions (three by the constant total_mass).  A thread owns VEC = 2
// consecutive lanes: one dwordx4 per state row and direction, one dwordx2 for the actions and the rewards, one 16-bit store
// for the done flags.  Compiled with -ffp-contract=off.
#include "kernels.hpp"

#include <cstdio>

#include "cartpole64.hpp"

namespace gymnet {

namespace {

template <class T, int VEC> struct VecOf { typedef T type __attribute__((ext_vector_type(VEC))); };

// VEC consecutive elements of one SoA row; `full` = all VEC lanes are inside the batch (else element by element)
template <class T, int VEC, bool NT>
__device__ __forceinline__ void load_row(const T *__restrict__ p, int64_t i0, int64_t n, bool full, T (&v)[VEC]) {
    if constexpr (VEC > 1) {
        if (full) {
            typedef typename VecOf<T, VEC>::type V;
            V t;
            if constexpr (NT) t = __builtin_nontemporal_load(reinterpret_cast<const V *>(p + i0));
            else t = *reinterpret_cast<const V *>(p + i0);
#pragma unroll
            for (int j = 0; j < VEC; ++j) v[j] = t[j];
            return;
        }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        if (i0 + j < n) { if constexpr (NT) v[j] = __builtin_nontemporal_load(p + i0 + j); else v[j] = p[i0 + j]; }
        else v[j] = T(0);
    }
}

template <class T, int VEC, bool NT>
__device__ __forceinline__ void store_row(T *__restrict__ p, int64_t i0, int64_t n, bool full, const T (&v)[VEC]) {
    if constexpr (VEC > 1) {
        if (full) {
            typedef typename VecOf<T, VEC>::type V;
            V t;
#pragma unroll
            for (int j = 0; j < VEC; ++j) t[j] = v[j];
            if constexpr (NT) __builtin_nontemporal_store(t, reinterpret_cast<V *>(p + i0));
            else *reinterpret_cast<V *>(p + i0) = t;
            return;
        }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j)
        if (i0 + j < n) { if constexpr (NT) __builtin_nontemporal_store(v[j], p + i0 + j); else p[i0 + j] = v[j]; }
}

__device__ __forceinline__ uint32_t lane_id64() {
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

}  // namespace

// What one thread reads for its VEC lanes before it can advance them
template <int VEC>
struct Lanes64 {
    double s[CartPole64::S][VEC];
    int32_t act[VEC], sbd[VEC], ep_len[VEC];
    float ep_ret[VEC];
};

template <int VEC, bool AUTORESET, bool EXTRAS, int NT>
__device__ __forceinline__ void load_lanes64(const StepArgs64 &a, int64_t i0, bool full, bool stats, Lanes64<VEC> &L) {
    constexpr bool NT_SL = (NT & 1) != 0, NT_A = (NT & 4) != 0;
    const int64_t n = a.n;
#pragma unroll
    for (int k = 0; k < CartPole64::S; ++k) load_row<double, VEC, NT_SL>(a.state + k * a.stride, i0, n, full, L.s[k]);
    load_row<int32_t, VEC, NT_A>(a.action, i0, n, full, L.act);
#pragma unroll
    for (int j = 0; j < VEC; ++j) L.sbd[j] = -1;
    if constexpr (!AUTORESET) load_row<int32_t, VEC, NT_SL>(a.sbd, i0, n, full, L.sbd);
    if constexpr (EXTRAS) {
        if (stats) { load_row<float, VEC, NT_SL>(a.ep_ret, i0, n, full, L.ep_ret); load_row<int32_t, VEC, NT_SL>(a.ep_len, i0, n, full, L.ep_len); }
    }
}

// advance the VEC lanes and write everything back.  NT as in kernels.hip: 1 state loads, 2 state stores, 4 action load,
// 8 reward / done stores.  EXTRAS: episode return / length bookkeeping, the max_episode_steps truncation, per-lane seeds.
template <int VEC, bool AUTORESET, bool EXTRAS, int NT>
__device__ __forceinline__ void advance_and_store64(const StepArgs64 &a, int64_t i0, bool full, bool stats, uint64_t tick, Lanes64<VEC> &L) {
    constexpr int S = CartPole64::S;
    constexpr bool NT_SS = (NT & 2) != 0, NT_O = (NT & 8) != 0;
    const int64_t n = a.n;
    float reward[VEC];
    uint8_t done[VEC];
    uint32_t after = 0;            // lanes of this thread stepped although they had already returned done (CartPoleEnv.cs:176-179)
    // wave-uniform: every pole angle of the wave inside the range where sin / cos need no reduction (identical bits either way)
    bool small = true;
#pragma unroll
    for (int j = 0; j < VEC; ++j) small = small && (__builtin_fabs(L.s[2][j]) <= kSmallAngle64);
    const bool wave_small = __ballot(!small) == 0;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        double sj[S];
#pragma unroll
        for (int k = 0; k < S; ++k) sj[k] = L.s[k][j];
        bool dn;
        if (wave_small) CartPole64::step<true>(sj, L.act[j], dn);
        else CartPole64::step<false>(sj, L.act[j], dn);
        float rw = 1.0f;                                                             // :168-183
        if constexpr (!AUTORESET) {
            if (dn) {
                if (L.sbd[j] == -1) L.sbd[j] = 0;
                else { after += (i0 + j < n) ? 1u : 0u; L.sbd[j] += 1; rw = 0.0f; }
            }
        }
        uint8_t db = dn ? 1 : 0;
        if constexpr (EXTRAS) {
            if (stats) {
                L.ep_ret[j] += rw;
                L.ep_len[j] += 1;
                if (a.max_episode_steps > 0 && L.ep_len[j] >= a.max_episode_steps) db |= 2;      // truncated (extension)
                if (db && i0 + j < n) {
                    a.fin_ret[i0 + j] = L.ep_ret[j];
                    a.fin_len[i0 + j] = L.ep_len[j];
                    if constexpr (AUTORESET) { L.ep_ret[j] = 0.0f; L.ep_len[j] = 0; }
                }
            }
        }
        if constexpr (AUTORESET) {
            if (db) {                                                                // the caller's `if (done) Reset()`, fused
                uint64_t key = a.seed;
                if constexpr (EXTRAS) { if (a.lane_seed && i0 + j < n) key = a.lane_seed[i0 + j]; }
                CartPole64::reset(sj, key, a.lane_offset + (uint64_t)(i0 + j), tick);
            }
        }
#pragma unroll
        for (int k = 0; k < S; ++k) L.s[k][j] = sj[k];
        reward[j] = rw;
        done[j] = db;
    }

    if constexpr (!AUTORESET) {       // one 64-bit atomic per wave, and only if the wave has something to report
        uint32_t total = 0;
#pragma unroll
        for (uint32_t c = 1; c <= (uint32_t)VEC; ++c) total += c * (uint32_t)__popcll(__ballot(after == c));
        if (total && lane_id64() == (uint32_t)(__ffsll((unsigned long long)__ballot(1)) - 1)) {
            const uint32_t shard = (uint32_t)((((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6) & (kShards - 1));
            atomicAdd(&a.after_done[shard * kAfterStride], (unsigned long long)total);
        }
    }

    store_row<float, VEC, NT_O>(a.reward, i0, n, full, reward);
    store_row<uint8_t, VEC, NT_O>(a.done, i0, n, full, done);
#pragma unroll
    for (int k = 0; k < S; ++k) store_row<double, VEC, NT_SS>(a.state + k * a.stride, i0, n, full, L.s[k]);
    if constexpr (!AUTORESET) store_row<int32_t, VEC, NT_SS>(a.sbd, i0, n, full, L.sbd);
    if constexpr (EXTRAS) {
        if (stats) { store_row<float, VEC, NT_SS>(a.ep_ret, i0, n, full, L.ep_ret); store_row<int32_t, VEC, NT_SS>(a.ep_len, i0, n, full, L.ep_len); }
    }
}

// ONE launch advances every lane by one env-step: a thread owns VEC consecutive lanes.
template <int VEC, bool AUTORESET, bool EXTRAS, int NT>
__global__ __launch_bounds__(256) void step_kernel_f64(const StepArgs64 a) {
    const int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * VEC;
    const uint64_t tick = a.tick2[a.parity];
    if (blockIdx.x == 0 && threadIdx.x == 0) a.tick2[a.parity ^ 1] = tick + 1;
    if (i0 >= a.n) return;
    const bool full = i0 + VEC <= a.n;
    bool stats = false;
    if constexpr (EXTRAS) stats = a.ep_ret != nullptr;
    Lanes64<VEC> L;
    load_lanes64<VEC, AUTORESET, EXTRAS, NT>(a, i0, full, stats, L);
    advance_and_store64<VEC, AUTORESET, EXTRAS, NT>(a, i0, full, stats, tick, L);
}

// Multi-item form (the float64 kernel has real arithmetic: ~270 binary64 VALU per env-step against 73 B): a thread owns ITEMS
// lane pairs (pair k at thread index + k * T, coalesced per item), issues the loads of ALL its pairs first, then advances and
// stores pair after pair — pair k's stores drain under pair k + 1's arithmetic, the shape that works for Acrobot
// (kernels.hip step_kernel_pipe; every load before the first store, so the one full vmcnt wait sits after pair 0's
// arithmetic).  Whole batches only (n a multiple of 2 * ITEMS * 256: the launcher falls back to the one-shot kernel
// otherwise); lean variant.  Same per-lane code and Philox counters: bit-identical.
template <int ITEMS, bool AUTORESET, int NT>
__global__ __launch_bounds__(256) void step_kernel_f64_pipe(const StepArgs64 a) {
    constexpr int VEC = 2;
    const uint64_t tick = a.tick2[a.parity];
    if (blockIdx.x == 0 && threadIdx.x == 0) a.tick2[a.parity ^ 1] = tick + 1;
    const int64_t T = (int64_t)gridDim.x * blockDim.x;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    Lanes64<VEC> L[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) load_lanes64<VEC, AUTORESET, false, NT>(a, (t + k * T) * VEC, true, false, L[k]);
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        if (k == 0) {   // pair 0's inputs are needed now (their first uses must not be hoisted into the load block)
#pragma unroll
            for (int c = 0; c < CartPole64::S; ++c) asm volatile("" : "+v"(L[0].s[c][0]), "+v"(L[0].s[c][1]));
        }
        advance_and_store64<VEC, AUTORESET, false, NT>(a, (t + k * T) * VEC, true, false, tick, L[k]);
    }
}

// Fused rollout in the float64 mode: T vector steps in ONE launch.  A thread keeps its VEC lanes in registers for
// all T steps, so per env-step only the action is read (4 B) and — when recording — observation / reward / done are written
// (37 B): the launch is bound by the ~270 binary64 VALU per env-step, not by memory.  The next step's action is loaded before
// the current step's arithmetic.  Same per-lane code, same Philox counters (tick0 + t): bit-identical to T one-step launches.
template <int VEC, bool AUTORESET>
__global__ __launch_bounds__(256) void rollout_kernel_f64(const StepArgs64 a, const RolloutArgs64 ro) {
    constexpr int S = CartPole64::S;
    const int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * VEC;
    const uint64_t tick0 = a.tick2[a.parity];
    if (blockIdx.x == 0 && threadIdx.x == 0) a.tick2[a.parity ^ 1] = tick0 + (uint64_t)ro.steps;
    const int64_t n = a.n;
    if (i0 >= n) return;
    const bool full = i0 + VEC <= n;
    double s[S][VEC];
#pragma unroll
    for (int k = 0; k < S; ++k) load_row<double, VEC, true>(a.state + k * a.stride, i0, n, full, s[k]);
    int32_t sbd[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) sbd[j] = -1;
    if constexpr (!AUTORESET) load_row<int32_t, VEC, true>(a.sbd, i0, n, full, sbd);
    int32_t act[VEC], act_next[VEC];
    load_row<int32_t, VEC, true>(a.action, i0, n, full, act);
    int64_t slice = 0;
    float reward[VEC];
    uint8_t done[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { reward[j] = 0.0f; done[j] = 0; act_next[j] = 0; }
    for (int64_t t = 0; t < ro.steps; ++t) {
        int64_t nslice = slice + 1;
        if (nslice == ro.ring) nslice = 0;
        if (t + 1 < ro.steps) load_row<int32_t, VEC, true>(a.action + nslice * ro.action_stride, i0, n, full, act_next);   // in flight during this step's arithmetic
        bool small = true;
#pragma unroll
        for (int j = 0; j < VEC; ++j) small = small && (__builtin_fabs(s[2][j]) <= kSmallAngle64);
        const bool wave_small = __ballot(!small) == 0;
        uint32_t after = 0;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            double sj[S];
#pragma unroll
            for (int k = 0; k < S; ++k) sj[k] = s[k][j];
            bool dn;
            if (wave_small) CartPole64::step<true>(sj, act[j], dn);
            else CartPole64::step<false>(sj, act[j], dn);
            float rw = 1.0f;
            if constexpr (!AUTORESET) {
                if (dn) {
                    if (sbd[j] == -1) sbd[j] = 0;
                    else { after += (i0 + j < n) ? 1u : 0u; sbd[j] += 1; rw = 0.0f; }
                }
            }
            if constexpr (AUTORESET) {
                if (dn) CartPole64::reset(sj, a.seed, a.lane_offset + (uint64_t)(i0 + j), tick0 + (uint64_t)t);
            }
#pragma unroll
            for (int k = 0; k < S; ++k) s[k][j] = sj[k];
            reward[j] = rw;
            done[j] = dn ? 1 : 0;
        }
        if constexpr (!AUTORESET) {
            uint32_t total = 0;
#pragma unroll
            for (uint32_t c = 1; c <= (uint32_t)VEC; ++c) total += c * (uint32_t)__popcll(__ballot(after == c));
            if (total && lane_id64() == (uint32_t)(__ffsll((unsigned long long)__ballot(1)) - 1)) {
                const uint32_t shard = (uint32_t)((((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6) & (kShards - 1));
                atomicAdd(&a.after_done[shard * kAfterStride], (unsigned long long)total);
            }
        }
        if (ro.rec_reward) store_row<float, VEC, true>(ro.rec_reward + t * n, i0, n, full, reward);
        if (ro.rec_done) store_row<uint8_t, VEC, true>(ro.rec_done + t * n, i0, n, full, done);
        if (ro.rec_obs) {
#pragma unroll
            for (int k = 0; k < S; ++k) store_row<double, VEC, true>(ro.rec_obs + (t * S + k) * n, i0, n, full, s[k]);
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) act[j] = act_next[j];
        slice = nslice;
    }
#pragma unroll
    for (int k = 0; k < S; ++k) store_row<double, VEC, false>(a.state + k * a.stride, i0, n, full, s[k]);
    store_row<float, VEC, false>(a.reward, i0, n, full, reward);
    store_row<uint8_t, VEC, false>(a.done, i0, n, full, done);
    if constexpr (!AUTORESET) store_row<int32_t, VEC, false>(a.sbd, i0, n, full, sbd);
}

// Reset: all lanes, or the lanes selected by a byte mask (which may alias a.done: a lane's flag is read before it is cleared)
__global__ __launch_bounds__(256) void reset_kernel_f64(const ResetArgs64 a) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t tick = a.tick2[a.parity];
    if (blockIdx.x == 0 && threadIdx.x == 0) a.tick2[a.parity ^ 1] = tick + 1;
    if (i >= a.n) return;
    if (a.mask && !a.mask[i]) return;
    double s[4];
    CartPole64::reset(s, a.lane_seed ? a.lane_seed[i] : a.seed, a.lane_offset + (uint64_t)i, tick);
#pragma unroll
    for (int k = 0; k < 4; ++k) a.state[k * a.stride + i] = s[k];
    if (a.sbd) a.sbd[i] = -1;            // CartPoleEnv.cs:64
    if (a.done) a.done[i] = 0;
    if (a.ep_ret) { a.ep_ret[i] = 0.0f; a.ep_len[i] = 0; }
}

// SoA [4][stride] -> row-major [n][4] float64 (what an NDArray<double> of shape (N, 4) holds); optionally reward / done beside
// it (the host boundary: `out_*` may be host-mapped memory, in which case the stores are the PCIe transfer)
__global__ __launch_bounds__(256) void export_f64_kernel(const double *__restrict__ obs, int64_t stride, const float *__restrict__ reward,
                                                         const uint8_t *__restrict__ done, double *__restrict__ out_obs,
                                                         float *__restrict__ out_reward, uint8_t *__restrict__ out_done, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (out_obs) {
        typedef double d2 __attribute__((ext_vector_type(2)));
        d2 lo, hi;
        lo.x = obs[i]; lo.y = obs[stride + i]; hi.x = obs[2 * stride + i]; hi.y = obs[3 * stride + i];
        d2 *dst = reinterpret_cast<d2 *>(out_obs + i * 4);            // 32 contiguous bytes per lane, 2 KiB per wave
        dst[0] = lo; dst[1] = hi;
    }
    if (out_reward) out_reward[i] = reward[i];
    if (out_done) out_done[i] = done[i];
}

// ---------------------------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------------------------
static inline unsigned grid64(int64_t items, int block) { return (unsigned)((items + block - 1) / block); }

// items > 1: the multi-item kernel (lean variant, vec 2, whole batches only)
static bool f64_pipe_ok(bool extras, const StepArgs64 &a, int vec, int items) {
    return items > 1 && items <= 4 && !extras && vec == 2 && a.n > 0 && (a.n % (2 * (int64_t)items * 256)) == 0;
}

hipError_t launch_step_f64(bool autoreset, bool extras, const StepArgs64 &a, int vec, int nt, int items, hipStream_t st) {
    if (vec != 2) vec = 1;
    if (nt != 12 && nt != 15) nt = 0;
    if (f64_pipe_ok(extras, a, vec, items)) {
        const dim3 pgrid((unsigned)(a.n / (2 * (int64_t)items * 256))), pblk(256);
#define GYMNET_P64(I, NTM)                                                                                     \
    do {                                                                                                       \
        if (autoreset) hipLaunchKernelGGL((step_kernel_f64_pipe<I, true, NTM>), pgrid, pblk, 0, st, a);         \
        else hipLaunchKernelGGL((step_kernel_f64_pipe<I, false, NTM>), pgrid, pblk, 0, st, a);                  \
    } while (0)
#define GYMNET_P64_NT(I)                                          \
    do {                                                          \
        if (nt == 15) GYMNET_P64(I, 15);                          \
        else if (nt == 12) GYMNET_P64(I, 12);                     \
        else GYMNET_P64(I, 0);                                    \
    } while (0)
        switch (items) {
            case 2: GYMNET_P64_NT(2); break;
            case 3: GYMNET_P64_NT(3); break;
            default: GYMNET_P64_NT(4); break;
        }
#undef GYMNET_P64_NT
#undef GYMNET_P64
        return hipGetLastError();
    }
    const int64_t threads = (a.n + vec - 1) / vec;
    const dim3 grid(grid64(threads > 0 ? threads : 1, 256)), blk(256);
#define GYMNET_L64(V, AR, EX, NTM) hipLaunchKernelGGL((step_kernel_f64<V, AR, EX, NTM>), grid, blk, 0, st, a)
#define GYMNET_L64_NT(V, AR, EX)                                  \
    do {                                                          \
        if (nt == 15) GYMNET_L64(V, AR, EX, 15);                  \
        else if (nt == 12) GYMNET_L64(V, AR, EX, 12);             \
        else GYMNET_L64(V, AR, EX, 0);                            \
    } while (0)
#define GYMNET_L64_EX(V, AR)                                      \
    do {                                                          \
        if (extras) GYMNET_L64_NT(V, AR, true);                   \
        else GYMNET_L64_NT(V, AR, false);                         \
    } while (0)
    if (vec == 2) { if (autoreset) GYMNET_L64_EX(2, true); else GYMNET_L64_EX(2, false); }
    else          { if (autoreset) GYMNET_L64_EX(1, true); else GYMNET_L64_EX(1, false); }
#undef GYMNET_L64_EX
#undef GYMNET_L64_NT
#undef GYMNET_L64
    return hipGetLastError();
}

hipError_t launch_rollout_fused_f64(bool autoreset, const StepArgs64 &a, const RolloutArgs64 &r, int vec, hipStream_t st) {
    if (vec != 2) vec = 1;
    const int64_t threads = (a.n + vec - 1) / vec;
    const dim3 grid(grid64(threads > 0 ? threads : 1, 256)), blk(256);
    if (vec == 2) {
        if (autoreset) hipLaunchKernelGGL((rollout_kernel_f64<2, true>), grid, blk, 0, st, a, r);
        else hipLaunchKernelGGL((rollout_kernel_f64<2, false>), grid, blk, 0, st, a, r);
    } else {
        if (autoreset) hipLaunchKernelGGL((rollout_kernel_f64<1, true>), grid, blk, 0, st, a, r);
        else hipLaunchKernelGGL((rollout_kernel_f64<1, false>), grid, blk, 0, st, a, r);
    }
    return hipGetLastError();
}

int describe_step_kernel_f64(bool autoreset, bool extras, int vec, int nt, int items, int64_t n, char *buf, size_t cap) {
    if (vec != 2) vec = 1;
    if (nt != 12 && nt != 15) nt = 0;
    StepArgs64 probe{};
    probe.n = n;
    if (f64_pipe_ok(extras, probe, vec, items)) return std::snprintf(buf, cap, "step_kernel_f64_pipe<%d,%s,%d>", items, autoreset ? "true" : "false", nt);
    return std::snprintf(buf, cap, "step_kernel_f64<%d,%s,%s,%d>", vec, autoreset ? "true" : "false", extras ? "true" : "false", nt);
}

hipError_t launch_reset_f64(const ResetArgs64 &a, hipStream_t st) {
    hipLaunchKernelGGL(reset_kernel_f64, dim3(grid64(a.n > 0 ? a.n : 1, 256)), dim3(256), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_pack_obs_f64(int obs_dim, const double *obs, int64_t stride, double *out, int64_t n, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    if (obs_dim != 4) return hipErrorInvalidValue;
    hipLaunchKernelGGL(export_f64_kernel, dim3(grid64(n, 256)), dim3(256), 0, st, obs, stride, (const float *)nullptr,
                       (const uint8_t *)nullptr, out, (float *)nullptr, (uint8_t *)nullptr, n);
    return hipGetLastError();
}

hipError_t launch_export_small_f64(int obs_dim, const double *obs, int64_t stride, const float *reward, const uint8_t *done,
                                   double *out_obs, float *out_reward, uint8_t *out_done, int64_t n, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    if (obs_dim != 4) return hipErrorInvalidValue;
    hipLaunchKernelGGL(export_f64_kernel, dim3(grid64(n, 256)), dim3(256), 0, st, obs, stride, reward, done, out_obs, out_reward, out_done, n);
    return hipGetLastError();
}

hipError_t launch_export_host_f64(int obs_dim, const double *obs, int64_t stride, const float *reward, const uint8_t *done,
                                  double *out_obs, float *out_reward, uint8_t *out_done, int64_t n, hipStream_t st) {
    return launch_export_small_f64(obs_dim, obs, stride, reward, done, out_obs, out_reward, out_done, n, st);
}

}  // namespace gymnet
