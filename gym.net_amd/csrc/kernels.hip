// kernels.hip — HIP kernels of the batched classic-control engine, written for gfx950 (CDNA4, wave64).
//
// The hot path is HBM-bound streaming (CartPole: 41 algorithmic bytes per env-step against ~105 VALU), so the
// design rules are the memory ones: structure-of-arrays, one env per lane, 16-byte (dwordx4) accesses per lane
// on every stream, a static block->lane map, no LDS staging of the streams (there is no reuse to stage: each state word is read
// once and written once per launch; LDS carries only the lane-to-lane hand-offs: reset_pending_wave, compact_done_kernel,
// step_kernel_lds), no MFMA (no dense contraction exists on this path).
// What the counters say (profiles/rocprof_pmc_r01.txt): every launch fetches all of its input bytes through the
// fabric again — the per-XCD L2s keep nothing across a kernel boundary — so the only cross-launch reuse level
// is the 256 MiB Infinity Cache, and an XCD-aware block remap would buy nothing here; what matters instead is
// which streams are marked non-temporal (the NT template parameter, chosen from the batch size in capi.hip).
// At 2^20 lanes all waves are resident at once and run load -> math -> store in lock-step, so beyond the memory
// floor every VALU instruction is exposed: hence the in-house sincos, the fma-pair constant division and the
// loop-compacted Philox reset (envs.hpp, below).  Done-lane compaction is one wave ballot + one atomic per wave.
//
// Compiled with -ffp-contract=off (see envs.hpp).
#include "kernels.hpp"

#include <cstdio>
#include <type_traits>

#include "envs.hpp"

namespace gymnet {

// ---------------------------------------------------------------------------------------------
// VEC-wide lane access helpers.  i0 is a multiple of VEC; arrays are 16-byte aligned (capi.hip
// checks this before it picks VEC = 4), so the full-vector path is one dwordx4 per lane.
// NT = non-temporal (streaming) access: `global_load/store ... nt`.  Which streams get it is a
// measured policy (tools/probe_step.hip): it decides what stays in the 256 MiB
// Infinity Cache between two launches.
// ---------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int32_t i32x4 __attribute__((ext_vector_type(4)));

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int32_t i32x2 __attribute__((ext_vector_type(2)));

template <int VEC, bool NT, bool GUARD>
__device__ __forceinline__ void load_f32(const float *__restrict__ p, int64_t i0, int64_t n, float (&v)[VEC]) {
    if constexpr (VEC == 2) {
        if (!GUARD || i0 + 2 <= n) {
            f32x2 t;
            if constexpr (NT) t = __builtin_nontemporal_load(reinterpret_cast<const f32x2 *>(p + i0));
            else t = *reinterpret_cast<const f32x2 *>(p + i0);
            v[0] = t.x; v[1] = t.y;
            return;
        }
    }
    if constexpr (VEC == 4) {
        if (!GUARD || i0 + 4 <= n) {
            f32x4 t;
            if constexpr (NT) t = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p + i0));
            else t = *reinterpret_cast<const f32x4 *>(p + i0);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
            return;
        }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        if (!GUARD || i0 + j < n) { if constexpr (NT) v[j] = __builtin_nontemporal_load(p + i0 + j); else v[j] = p[i0 + j]; }
        else v[j] = 0.0f;
    }
}

template <int VEC, bool NT, bool GUARD>
__device__ __forceinline__ void store_f32(float *__restrict__ p, int64_t i0, int64_t n, const float (&v)[VEC]) {
    if constexpr (VEC == 2) {
        if (!GUARD || i0 + 2 <= n) {
            f32x2 t; t.x = v[0]; t.y = v[1];
            if constexpr (NT) __builtin_nontemporal_store(t, reinterpret_cast<f32x2 *>(p + i0));
            else *reinterpret_cast<f32x2 *>(p + i0) = t;
            return;
        }
    }
    if constexpr (VEC == 4) {
        if (!GUARD || i0 + 4 <= n) {
            f32x4 t; t.x = v[0]; t.y = v[1]; t.z = v[2]; t.w = v[3];
            if constexpr (NT) __builtin_nontemporal_store(t, reinterpret_cast<f32x4 *>(p + i0));
            else *reinterpret_cast<f32x4 *>(p + i0) = t;
            return;
        }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j)
        if (!GUARD || i0 + j < n) { if constexpr (NT) __builtin_nontemporal_store(v[j], p + i0 + j); else p[i0 + j] = v[j]; }
}

template <int VEC, bool NT, bool GUARD>
__device__ __forceinline__ void load_i32(const int32_t *__restrict__ p, int64_t i0, int64_t n, int32_t (&v)[VEC]) {
    if constexpr (VEC == 2) {
        if (!GUARD || i0 + 2 <= n) {
            i32x2 t;
            if constexpr (NT) t = __builtin_nontemporal_load(reinterpret_cast<const i32x2 *>(p + i0));
            else t = *reinterpret_cast<const i32x2 *>(p + i0);
            v[0] = t.x; v[1] = t.y;
            return;
        }
    }
    if constexpr (VEC == 4) {
        if (!GUARD || i0 + 4 <= n) {
            i32x4 t;
            if constexpr (NT) t = __builtin_nontemporal_load(reinterpret_cast<const i32x4 *>(p + i0));
            else t = *reinterpret_cast<const i32x4 *>(p + i0);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
            return;
        }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        if (!GUARD || i0 + j < n) { if constexpr (NT) v[j] = __builtin_nontemporal_load(p + i0 + j); else v[j] = p[i0 + j]; }
        else v[j] = 0;
    }
}

template <int VEC, bool NT, bool GUARD>
__device__ __forceinline__ void store_i32(int32_t *__restrict__ p, int64_t i0, int64_t n, const int32_t (&v)[VEC]) {
    if constexpr (VEC == 2) {
        if (!GUARD || i0 + 2 <= n) {
            i32x2 t; t.x = v[0]; t.y = v[1];
            if constexpr (NT) __builtin_nontemporal_store(t, reinterpret_cast<i32x2 *>(p + i0));
            else *reinterpret_cast<i32x2 *>(p + i0) = t;
            return;
        }
    }
    if constexpr (VEC == 4) {
        if (!GUARD || i0 + 4 <= n) {
            i32x4 t; t.x = v[0]; t.y = v[1]; t.z = v[2]; t.w = v[3];
            if constexpr (NT) __builtin_nontemporal_store(t, reinterpret_cast<i32x4 *>(p + i0));
            else *reinterpret_cast<i32x4 *>(p + i0) = t;
            return;
        }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j)
        if (!GUARD || i0 + j < n) { if constexpr (NT) __builtin_nontemporal_store(v[j], p + i0 + j); else p[i0 + j] = v[j]; }
}

template <int VEC, bool NT, bool GUARD>
__device__ __forceinline__ void store_u8(uint8_t *__restrict__ p, int64_t i0, int64_t n, const uint8_t (&v)[VEC]) {
    if constexpr (VEC == 2) {
        if (!GUARD || i0 + 2 <= n) {
            const uint16_t w = (uint16_t)((uint16_t)v[0] | ((uint16_t)v[1] << 8));
            if constexpr (NT) __builtin_nontemporal_store(w, reinterpret_cast<uint16_t *>(p + i0));
            else *reinterpret_cast<uint16_t *>(p + i0) = w;
            return;
        }
    }
    if constexpr (VEC == 4) {
        if (!GUARD || i0 + 4 <= n) {
            const uint32_t w = (uint32_t)v[0] | ((uint32_t)v[1] << 8) | ((uint32_t)v[2] << 16) | ((uint32_t)v[3] << 24);
            if constexpr (NT) __builtin_nontemporal_store(w, reinterpret_cast<uint32_t *>(p + i0));
            else *reinterpret_cast<uint32_t *>(p + i0) = w;
            return;
        }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j)
        if (!GUARD || i0 + j < n) { if constexpr (NT) __builtin_nontemporal_store(v[j], p + i0 + j); else p[i0 + j] = v[j]; }
}

__device__ __forceinline__ uint32_t lane_id() {
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

// Where state row k lives.  A state component that the observation repeats verbatim (Pendulum: theta_dot = obs[2]; Acrobot:
// dtheta1, dtheta2 = obs[4], obs[5]; Env::OBS_ROW_OF_STATE) is stored ONCE, in the observation array: the step reads it from
// there and never writes its row of the state array — 8 of Acrobot's 45 written bytes per env-step, 4 of Pendulum's 21.  At
// 2^20 lanes both kernels are short of WRITE bandwidth, and the 18 % fewer written bytes are worth 10 % of the launch
// (Acrobot 13.9 -> 12.6 us, Pendulum 6.68 -> 6.03 us, profiles/dedup_probe_r03.txt).
template <class Env>
__device__ __forceinline__ constexpr bool state_row_own(int k) {
    if constexpr (Env::OBS_ALIASES_STATE) return true; else return Env::OBS_ROW_OF_STATE[k] < 0;
}
template <class Env>
__device__ __forceinline__ const float *state_row_src(const float *state, int64_t state_stride, const float *obs_in, int64_t obs_stride, int k) {
    if constexpr (Env::OBS_ALIASES_STATE) return state + k * state_stride;
    else return Env::OBS_ROW_OF_STATE[k] < 0 ? state + k * state_stride : obs_in + Env::OBS_ROW_OF_STATE[k] * obs_stride;
}

// shard of the calling wave for the sharded counters / done list (StepArgs)
__device__ __forceinline__ uint32_t wave_shard() {
    return (uint32_t)((((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6) & (kShards - 1));
}

// one 64-bit atomic per WAVE (and only if the wave has something to report) into the wave's shard
template <int VEC>
__device__ __forceinline__ void count_after_done(const StepArgs &a, const bool (&after)[VEC]) {
    uint32_t total = 0;
#pragma unroll
    for (int j = 0; j < VEC; ++j) total += (uint32_t)__popcll(__ballot(after[j]));
    if (total && lane_id() == (uint32_t)(__ffsll((unsigned long long)__ballot(1)) - 1))
        atomicAdd(&a.after_done[wave_shard() * kAfterStride], (unsigned long long)total);
}

// One env-step of ONE sub-lane: the dynamics, then — without auto-reset, for envs that carry it — the reference's
// steps_beyond_done rule (CartPoleEnv.cs:168-183): reward 1 until and including the step the pole falls, 0 afterwards.
// `after` reports a step taken on a lane that had already returned done (the reference's console warning, :176-179).
template <class Env, bool AUTORESET, bool SMALL_ANGLE = false>
__device__ __forceinline__ void advance_sublane(float (&sj)[Env::S], typename Env::Action act, int32_t &sbd, float &rw,
                                                bool &dn, bool &after, bool in_range, float (&oj)[Env::O]) {
    if constexpr (Env::HAS_SMALL_ANGLE_PATH) Env::template step<SMALL_ANGLE>(sj, act, rw, dn);
    else if constexpr (Env::OBS_ALIASES_STATE) Env::step(sj, act, rw, dn);
    else Env::step_observe(sj, act, rw, dn, oj);          // observation of the new (pre-reset) state
    if constexpr (!AUTORESET && Env::HAS_SBD) {
        if (dn) {
            if (sbd == -1) { sbd = 0; }
            else { after = in_range; sbd += 1; rw = 0.0f; }
        }
    }
}

// Envs with a small-angle path (CartPole): true when EVERY sub-lane of EVERY lane of the wave holds an angle inside the range
// where the trigonometry needs no reduction (envs.hpp sincos_tiny).  Wave-uniform, so the two code paths never diverge; with
// the fused auto-reset the pole angle is below the termination threshold at every entry and the fast path is the only one run.
template <class Env, int VEC>
__device__ __forceinline__ bool wave_angles_small(const float (&s)[Env::S][VEC]) {
    if constexpr (!Env::HAS_SMALL_ANGLE_PATH) return false;
    else {
        bool small = true;
#pragma unroll
        for (int j = 0; j < VEC; ++j) small = small && (fabsf(s[Env::ANGLE_ROW][j]) <= kSmallAngle);
        return __ballot(!small) == 0;
    }
}

// One env-step of ALL VEC sub-lanes of a thread.  Generic: sub-lane after sub-lane.  Envs that provide a two-lane packed
// form (Acrobot: both envs of a thread ride the v_pk_*_f32 instructions, envs.hpp) take it when VEC == 2; per element the
// arithmetic is the same IEEE sequence, so the results are bit-identical to the sub-lane loop.
template <class Env, int VEC, bool AUTORESET, bool GUARD, bool PACK = true>
__device__ __forceinline__ void advance_all(float (&s)[Env::S][VEC], typename Env::Action (&act)[VEC], int32_t (&sbd)[VEC],
                                            float (&rw)[VEC], bool (&dn)[VEC], bool (&after)[VEC], float (&o)[Env::O][VEC],
                                            int64_t i0, int64_t n) {
    constexpr int S = Env::S, O = Env::O;
#ifndef GYMNET_PROBE_NO_PACK      // probe builds only: two lanes per thread, scalar arithmetic
    constexpr bool kPack = true;
#else
    constexpr bool kPack = false;
#endif
    if constexpr (Env::PACKED2 && VEC == 2 && kPack && PACK) {
        Env::step_observe_x2(s, act, rw, dn, o);
    } else {
        auto all_sublanes = [&](auto small_tag) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                float sj[S], oj[O];
#pragma unroll
                for (int k = 0; k < S; ++k) sj[k] = s[k][j];
                advance_sublane<Env, AUTORESET, decltype(small_tag)::value>(sj, act[j], sbd[j], rw[j], dn[j], after[j], !GUARD || i0 + j < n, oj);
#pragma unroll
                for (int k = 0; k < S; ++k) s[k][j] = sj[k];
                if constexpr (!Env::OBS_ALIASES_STATE) {
#pragma unroll
                    for (int k = 0; k < O; ++k) o[k][j] = oj[k];
                }
            }
        };
        if constexpr (Env::HAS_SMALL_ANGLE_PATH) {
            if (wave_angles_small<Env, VEC>(s)) all_sublanes(std::true_type{});
            else all_sublanes(std::false_type{});
        } else {
            all_sublanes(std::false_type{});
        }
    }
}

// Fused auto-reset of the sub-lanes flagged in `pending`.  ~4.5 % of CartPole lanes finish per step, so ~95 % of
// 64-lane waves hold a finished lane in EVERY sub-lane position: a per-sub-lane `if (done) philox()` would make every
// wave pay VEC Philox passes.  Instead each thread drains its finished sub-lanes one per loop trip; the trips a wave
// pays are max over its lanes of #finished sub-lanes (1.6 on average instead of 3.8), and waves with no finished lane
// skip the loop through the exec mask.
template <class Env, int VEC, bool LANE_SEEDS>
__device__ __forceinline__ void reset_pending(uint32_t pending, float (&s)[Env::S][VEC], float (&o)[Env::O][VEC], const StepArgs &a,
                                              int64_t i0, int64_t n, uint64_t tick) {
    constexpr int S = Env::S, O = Env::O;
    while (pending) {
        const int j = __ffs(pending) - 1;
        pending &= pending - 1;
        uint64_t key = a.seed;
        if constexpr (LANE_SEEDS) {
            if (a.lane_seed && i0 + j < n) key = a.lane_seed[i0 + j];
        }
        const PhiloxWords r = lane_words(key, a.lane_offset + (uint64_t)(i0 + j), tick);
        float sj[S];
        Env::reset(sj, r);
        float oj[O];
        if constexpr (!Env::OBS_ALIASES_STATE) Env::observe_fresh(sj, oj);
#pragma unroll
        for (int jj = 0; jj < VEC; ++jj) {
            if (jj == j) {
#pragma unroll
                for (int k = 0; k < S; ++k) s[k][jj] = sj[k];
                if constexpr (!Env::OBS_ALIASES_STATE) {
#pragma unroll
                    for (int k = 0; k < O; ++k) o[k][jj] = oj[k];
                }
            }
        }
    }
}

// Wave-compacted form of the fused auto-reset (RESETF = 1; envs whose observation aliases the state, dwordx4 lanes; lean and
// bookkeeping variants alike).
// reset_pending() above makes every wave pay max-over-lanes Philox passes (1.6 on average for CartPole) with ~3 of 64 lanes
// active in each.  Here the wave's finished (lane, sub-lane) slots — 11.5 on average at 2^20 CartPole lanes — are ranked by
// ballot + mbcnt, handed to the FIRST `total` lanes through a wave-private LDS table, drawn in ONE Philox pass with those
// lanes active, and returned to their owners through LDS as one 16-byte read per finished sub-lane.  The Philox counter is
// the slot's global lane id, exactly as in reset_pending(), so the two forms draw the same bits.  LDS traffic of one wave is
// in order, so the only synchronisation is compiler-level (wavefront-scope fences); no s_barrier.
template <class Env>
struct ResetScratch {
    uint32_t slot[64];              // rank -> owner lane * VEC + sub-lane
    float draw[64][Env::S];         // rank -> the drawn state
};

__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <class Env, int VEC, bool LANE_SEEDS>
__device__ __forceinline__ void reset_pending_wave(uint32_t pending, float (&s)[Env::S][VEC], const StepArgs &a, int64_t i0, int64_t n,
                                                   uint64_t tick, ResetScratch<Env> *sc) {
    constexpr int S = Env::S;
    static_assert(Env::OBS_ALIASES_STATE, "the compacted reset hands back the state only");
    const uint32_t lane = lane_id();
    uint32_t rank[VEC];
    uint32_t total = 0;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const uint64_t m = __ballot((pending >> j) & 1u);
        rank[j] = total + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        total += (uint32_t)__popcll(m);
    }
    if (total == 0) return;                                   // wave-uniform
    const int64_t wave_i0 = i0 - (int64_t)lane * VEC;          // first lane index of this wave
    // The drawing lanes are the wave's ACTIVE lanes.  In the batch's last (partial) wave the threads past the end have left the
    // kernel; the active ones are a prefix 0 .. A-1 (the lane index grows with the thread index), and a round serves A slots.
    const uint32_t A = (uint32_t)__popcll(__ballot(1));
    for (uint32_t base = 0; base < total; base += A) {         // wave-uniform; more finished slots than lanes in a wave: ~never
#pragma unroll
        for (int j = 0; j < VEC; ++j)
            if (((pending >> j) & 1u) && rank[j] - base < A) sc->slot[rank[j] - base] = lane * VEC + (uint32_t)j;
        wave_lds_fence();
        if (lane < total - base) {                             // (an active lane by construction: lane < A whenever it has a slot)
            const uint32_t sl = sc->slot[lane];
            const int64_t gl = wave_i0 + (int64_t)sl;
            uint64_t key = a.seed;
            if constexpr (LANE_SEEDS) {
                if (a.lane_seed && gl < n) key = a.lane_seed[gl];
            }
            const PhiloxWords r = lane_words(key, a.lane_offset + (uint64_t)gl, tick);
            float sj[S];
            Env::reset(sj, r);
#pragma unroll
            for (int k = 0; k < S; ++k) sc->draw[lane][k] = sj[k];
        }
        wave_lds_fence();
        // every lane reads a row for each of its sub-lanes (clamped index; all reads in flight together, ONE wait) and keeps
        // it only where the sub-lane finished: a branch per sub-lane would serialise four LDS round trips
        float got[VEC][S];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const uint32_t r = rank[j] - base;
#pragma unroll
            for (int k = 0; k < S; ++k) got[j][k] = sc->draw[r < A ? r : 0u][k];
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const bool mine = ((pending >> j) & 1u) && rank[j] - base < A;
#pragma unroll
            for (int k = 0; k < S; ++k) s[k][j] = mine ? got[j][k] : s[k][j];
        }
        wave_lds_fence();
    }
}

// ---------------------------------------------------------------------------------------------
// The vector step: ONE launch advances every lane by one env-step.
//   Env       dynamics (envs.hpp)
//   VEC       envs per thread (4 = dwordx4 streams; 1 = fallback for unaligned external buffers)
//   AUTORESET fuse the caller's `if (done) Reset()` (README.md:36-40) as a masked Philox reset
//   EXTRAS    done-list compaction, episode statistics, terminal observations, per-lane seeds, time limit
//   NT        non-temporal mask: 1 state loads, 2 state/obs stores, 4 action load, 8 reward/done stores
// ---------------------------------------------------------------------------------------------
//   GUARD     per-element bounds checks; only the last (partial) workgroup of a launch runs the guarded body
// What one thread reads for its VEC lanes before it can advance them.  (Kept separate from the arithmetic: round 2 tried a
// grid-stride kernel that had the NEXT lanes' loads in flight during the current lanes' arithmetic — slower on every env,
// profiles/pipelined_kernel_probe_r02.txt: loads and stores share gfx9's in-order vmcnt, so waiting for a prefetch also
// waits for the previous lanes' stores.)
template <class Env, int VEC>
struct LaneInputs {
    float s[Env::S][VEC];
    typename Env::Action act[VEC];
    int32_t sbd[VEC];
};

template <class Env, int VEC, bool AUTORESET, int NT, bool GUARD>
__device__ __forceinline__ void load_inputs(const StepArgs &a, const int64_t i0, LaneInputs<Env, VEC> &in) {
    constexpr bool NT_SL = (NT & 1) != 0, NT_A = (NT & 4) != 0;
    const int64_t n = a.n;
#pragma unroll
    for (int k = 0; k < Env::S; ++k)
        load_f32<VEC, NT_SL, GUARD>(state_row_src<Env>(a.state, a.state_stride, a.obs_in, a.obs_stride, k), i0, n, in.s[k]);
    if constexpr (Env::BOX_ACTION) load_f32<VEC, NT_A, GUARD>(static_cast<const float *>(a.action), i0, n, in.act);
    else load_i32<VEC, NT_A, GUARD>(static_cast<const int32_t *>(a.action), i0, n, in.act);
#pragma unroll
    for (int j = 0; j < VEC; ++j) in.sbd[j] = 0;
    if constexpr (!AUTORESET && Env::HAS_SBD) load_i32<VEC, NT_SL, GUARD>(a.sbd, i0, n, in.sbd);
}

template <class Env, int VEC, bool AUTORESET, bool EXTRAS, int NT, bool GUARD, int RESETF = 0, bool PACK = true>
__device__ __forceinline__ void advance_and_store(const StepArgs &a, const int64_t i0, const uint64_t tick, LaneInputs<Env, VEC> &in,
                                                  ResetScratch<Env> *sc = nullptr) {
    constexpr int S = Env::S, O = Env::O;
    constexpr bool NT_SS = (NT & 2) != 0, NT_O = (NT & 8) != 0;
    const int64_t n = a.n;
    float (&s)[S][VEC] = in.s;
    typename Env::Action (&act)[VEC] = in.act;
    int32_t (&sbd)[VEC] = in.sbd;

    constexpr bool NT_SL = (NT & 1) != 0;
    float ep_ret[VEC], fin_ret[VEC];
    int32_t ep_len[VEC], fin_len[VEC];
    bool stats = false;
    if constexpr (EXTRAS) {
        stats = a.ep_ret != nullptr;
        // running return / length: read-modify-write streams like the state, same non-temporal policy
        if (stats) { load_f32<VEC, NT_SL, GUARD>(a.ep_ret, i0, n, ep_ret); load_i32<VEC, NT_SL, GUARD>(a.ep_len, i0, n, ep_len); }
#pragma unroll
        for (int j = 0; j < VEC; ++j) { fin_ret[j] = 0.0f; fin_len[j] = 0; }
    }

    float reward[VEC];
    uint8_t done[VEC];
    bool finished[VEC];
    bool after[VEC];          // sub-lane was stepped although it had already returned done (no auto-reset only)
#pragma unroll
    for (int j = 0; j < VEC; ++j) after[j] = false;
    float o[O][VEC];
    uint32_t pending = 0;     // sub-lanes of this thread that finished and await their reset draw

    float rwv[VEC];
    bool dnv[VEC];
    advance_all<Env, VEC, AUTORESET, GUARD, PACK>(s, act, sbd, rwv, dnv, after, o, i0, n);

#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const float rw = rwv[j];
        uint8_t db = dnv[j] ? 1 : 0;
        if constexpr (EXTRAS) {
            if (stats) {
                ep_ret[j] += rw;
                ep_len[j] += 1;
                if (a.max_episode_steps > 0 && ep_len[j] >= a.max_episode_steps) db |= 2;   // truncated (extension)
            }
        }
        const bool fin = db != 0;
        finished[j] = fin && (!GUARD || i0 + j < n);
        done[j] = db;
        reward[j] = rw;

        if constexpr (EXTRAS) {
            // The finished lanes' records go to the dense per-lane arrays (scattered 4-byte stores, one cache line each) whenever
            // those are handed in — always, unless the caller opted for compact records only (GYMNET_FLAG_COMPACT_RECORDS_ONLY:
            // capi.hip then passes NULL here) — so the dense "last finished episode per lane" view is current after any sequence
            // of launches, read or not.  With a done list (below) they are ALSO written compacted at the lane's
            // position in the list.
            if (fin && a.final_obs && (!GUARD || i0 + j < n)) {
#pragma unroll
                for (int k = 0; k < O; ++k) a.final_obs[k * n + i0 + j] = Env::OBS_ALIASES_STATE ? s[k < S ? k : 0][j] : o[k][j];
            }
            if (stats && fin && (!GUARD || i0 + j < n)) {
                fin_ret[j] = ep_ret[j];
                fin_len[j] = ep_len[j];
                if (a.fin_ret) { a.fin_ret[i0 + j] = ep_ret[j]; a.fin_len[i0 + j] = ep_len[j]; }
                if constexpr (AUTORESET) { ep_ret[j] = 0.0f; ep_len[j] = 0; }
            }
        }

        if constexpr (AUTORESET) pending |= fin ? (1u << j) : 0u;
    }

    if constexpr (!AUTORESET && Env::HAS_SBD) count_after_done<VEC>(a, after);

    // reward / done do not depend on the reset draw: get them on their way before the Philox rounds
    store_f32<VEC, NT_O, GUARD>(a.reward, i0, n, reward);
    store_u8<VEC, NT_O, GUARD>(a.done, i0, n, done);

    if constexpr (EXTRAS) {
        if (a.done_list) {
            // wave64 compaction (before the reset overwrites the terminal state): ballot per sub-lane, one atomic per wave into
            // the wave's shard, order inside the list unspecified.  Everything known about a finished lane is written at ITS
            // POSITION in the list — lane id, and with the corresponding flags its episode return / length and its terminal
            // observation: a wave's ~11 finished lanes write one or two contiguous cache lines per array instead of one
            // scattered line each (compacted (lane, return, length) records, BasePlaySession.cs:58-69).
            const uint32_t lane = lane_id();
            uint32_t off[VEC];
            uint32_t total = 0;
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const uint64_t m = __ballot(finished[j]);
                off[j] = total + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                total += (uint32_t)__popcll(m);
            }
            if (total) {   // wave-uniform
                const int leader = __ffsll((unsigned long long)__ballot(1)) - 1;
                const uint32_t shard = wave_shard();
                uint32_t base = 0;
                if ((int)lane == leader)
                    base = atomicAdd(&a.done_count2[a.cparity * (kShards * kCountStride) + shard * kCountStride], total);
                base = __shfl(base, leader);
                const int64_t seg0 = (int64_t)shard * a.done_cap;
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    if (!finished[j]) continue;
                    const int64_t pos = seg0 + base + off[j];
                    a.done_list[pos] = (int32_t)(i0 + j);
                    if (stats) { a.rec_ret[pos] = fin_ret[j]; a.rec_len[pos] = fin_len[j]; }
                    if (a.rec_obs) {
#pragma unroll
                        for (int k = 0; k < O; ++k)
                            a.rec_obs[((int64_t)shard * O + k) * a.done_cap + base + off[j]] = Env::OBS_ALIASES_STATE ? s[k < S ? k : 0][j] : o[k][j];
                    }
                }
            }
        }
    }

    if constexpr (AUTORESET && RESETF == 1) reset_pending_wave<Env, VEC, EXTRAS>(pending, s, a, i0, n, tick, sc);
    else if constexpr (AUTORESET) reset_pending<Env, VEC, EXTRAS>(pending, s, o, a, i0, n, tick);

#pragma unroll
    for (int k = 0; k < S; ++k)
        if (state_row_own<Env>(k)) store_f32<VEC, NT_SS, GUARD>(a.state_out + k * a.state_stride, i0, n, s[k]);
    if constexpr (!Env::OBS_ALIASES_STATE) {
#pragma unroll
        for (int k = 0; k < O; ++k) store_f32<VEC, NT_SS, GUARD>(a.obs + k * a.obs_stride, i0, n, o[k]);
    }
    if constexpr (!AUTORESET && Env::HAS_SBD) store_i32<VEC, NT_SS, GUARD>(a.sbd, i0, n, sbd);

    if constexpr (EXTRAS) {
        if (stats) { store_f32<VEC, NT_SS, GUARD>(a.ep_ret, i0, n, ep_ret); store_i32<VEC, NT_SS, GUARD>(a.ep_len, i0, n, ep_len); }
    }
}

template <class Env, int VEC, bool AUTORESET, bool EXTRAS, int NT, bool GUARD, int RESETF = 0>
__device__ __forceinline__ void step_body(const StepArgs &a, const int64_t i0, const uint64_t tick, ResetScratch<Env> *sc = nullptr) {
    LaneInputs<Env, VEC> in;
    load_inputs<Env, VEC, AUTORESET, NT, GUARD>(a, i0, in);
    advance_and_store<Env, VEC, AUTORESET, EXTRAS, NT, GUARD, RESETF>(a, i0, tick, in, sc);
}

// ---------------------------------------------------------------------------------------------
// Multi-lane variant for the env with real arithmetic (Acrobot).  The one-shot kernel runs its (two) wave generations
// nearly in lock-step — load burst, ~450 VALU, store burst — so at 2^20 lanes about half of the arithmetic and the head /
// tail bursts are exposed.  Here a thread owns ITEMS lanes (i, i + T, ..., coalesced per item), fully unrolled:
//     issue the loads of ALL its lanes | compute lane 0 | (all loads have landed) store lane 0 | compute lane 1 | store 1 ...
// so lane k's stores drain under lane k+1's arithmetic and only the first lane's loads and the last lane's stores are
// exposed.  The shape is dictated by how the compiler must treat gfx9's single vmcnt: with loads AND stores pending it
// has to assume out-of-order completion and emits vmcnt(0) — a real software pipeline (prefetch lane k+2 while computing
// lane k) therefore stalls on the previous lane's stores every trip, as a loop (profiles/pipelined_kernel_probe_r02.txt)
// and fully unrolled alike.  With every load issued before the first store there is exactly one full wait, placed after
// lane 0's arithmetic where it costs nothing.  Bit-identical to the one-shot kernel (same per-lane code, same counters).
// ---------------------------------------------------------------------------------------------
template <class Env>
struct LaneOutputs { float s[Env::S], o[Env::O], reward; uint8_t done; int32_t sbd; };

template <class Env, bool AUTORESET>
__device__ __forceinline__ void compute_lane(const StepArgs &a, int64_t i, uint64_t tick, LaneInputs<Env, 1> &in, LaneOutputs<Env> &out) {
    constexpr int S = Env::S, O = Env::O;
    float o[O][1], rw[1];
    bool dn[1], after[1] = {false};
    advance_all<Env, 1, AUTORESET, false>(in.s, in.act, in.sbd, rw, dn, after, o, i, a.n);
    if constexpr (!AUTORESET && Env::HAS_SBD) count_after_done<1>(a, after);
    if constexpr (AUTORESET) reset_pending<Env, 1, false>(dn[0] ? 1u : 0u, in.s, o, a, i, a.n, tick);
#pragma unroll
    for (int k = 0; k < S; ++k) out.s[k] = in.s[k][0];
#pragma unroll
    for (int k = 0; k < O; ++k) out.o[k] = o[k][0];
    out.reward = rw[0]; out.done = dn[0] ? 1 : 0; out.sbd = in.sbd[0];
}

template <class Env, bool AUTORESET, int NT>
__device__ __forceinline__ void store_lane(const StepArgs &a, int64_t i, const LaneOutputs<Env> &out) {
    constexpr bool NT_SS = (NT & 2) != 0, NT_O = (NT & 8) != 0;
    auto st = [](float *p, float v, bool nt) { if (nt) __builtin_nontemporal_store(v, p); else *p = v; };
    st(a.reward + i, out.reward, NT_O);
    if constexpr (NT_O) __builtin_nontemporal_store(out.done, a.done + i); else a.done[i] = out.done;
#pragma unroll
    for (int k = 0; k < Env::S; ++k)
        if (state_row_own<Env>(k)) st(a.state_out + k * a.state_stride + i, out.s[k], NT_SS);
    if constexpr (!Env::OBS_ALIASES_STATE) {
#pragma unroll
        for (int k = 0; k < Env::O; ++k) st(a.obs + k * a.obs_stride + i, out.o[k], NT_SS);
    }
    if constexpr (!AUTORESET && Env::HAS_SBD) a.sbd[i] = out.sbd;
}

template <class Env, int ITEMS, bool AUTORESET, int NT>
__global__ __launch_bounds__(256) void step_kernel_pipe(const StepArgs a) {
    const uint64_t tick = a.tick2[a.parity];
    if (blockIdx.x == 0 && threadIdx.x == 0) a.tick2[a.parity ^ 1] = tick + 1;
    const int64_t T = (int64_t)gridDim.x * blockDim.x;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // Loads are UNCONDITIONAL (a lane past the end re-reads the last valid lane; only its stores are suppressed): a branch
    // around a group of loads makes the compiler's waitcnt bookkeeping treat the earlier groups as the most recent ones at
    // the join, and the wait for lane 0 below would then wait for every lane.
    LaneInputs<Env, 1> in[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const int64_t idx = i + k * T;
        load_inputs<Env, 1, AUTORESET, NT, false>(a, idx < a.n ? idx : a.n - 1, in[k]);
    }
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        LaneOutputs<Env> out;
        if (k == 0) {   // lane 0's inputs are needed now (and their first uses must not be hoisted into the load block)
#pragma unroll
            for (int c = 0; c < Env::S; ++c) asm volatile("" : "+v"(in[0].s[c][0]));
            asm volatile("" : "+v"(in[0].act[0]));
        }
        compute_lane<Env, AUTORESET>(a, i + k * T, tick, in[k], out);
        if (k == 0) {
            // touch every remaining lane's inputs AFTER lane 0's results exist (the extra operand ties each touch to them, or
            // the compiler hoists the touches to the top): the one full vmcnt wait of the kernel lands HERE, after lane 0's
            // arithmetic and before the first store, when the loads have long arrived
#pragma unroll
            for (int kk = 1; kk < ITEMS; ++kk) {
#pragma unroll
                for (int c = 0; c < Env::S; ++c) asm volatile("" : "+v"(in[kk].s[c][0]), "+v"(out.s[Env::S - 1]));
                asm volatile("" : "+v"(in[kk].act[0]), "+v"(out.reward));
            }
        }
        if (i + k * T < a.n) store_lane<Env, AUTORESET, NT>(a, i + k * T, out);
    }
}

// Round 4 probe (launch policy vec = 2 together with sequential_lanes = k): the multi-lane kernel over lane PAIRS — a thread owns
// ITEMS pairs of consecutive lanes (pair k at thread index + k * T), 8-byte accesses on every stream, scalar arithmetic lane after
// lane (NOT the packed-FP32 form: PACK = false), all loads first, then advance / store pair after pair.  The shape that took the
// float64 CartPole kernel from 14.4 to 13.1 us.  Whole batches only (n a multiple of 2 * ITEMS * 256); lean variant.  Same
// per-lane code and Philox counters as every other form: bit-identical.  Measured: profiles/acrobot_forms_r04.txt.
template <class Env, int ITEMS, bool AUTORESET, int NT>
__global__ __launch_bounds__(256) void step_kernel_pipe2(const StepArgs a) {
    const uint64_t tick = a.tick2[a.parity];
    if (blockIdx.x == 0 && threadIdx.x == 0) a.tick2[a.parity ^ 1] = tick + 1;
    const int64_t T = (int64_t)gridDim.x * blockDim.x;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    LaneInputs<Env, 2> in[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) load_inputs<Env, 2, AUTORESET, NT, false>(a, (t + k * T) * 2, in[k]);
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        if (k == 0) {   // pair 0's inputs are needed now (their first uses must not be hoisted into the load block)
#pragma unroll
            for (int c = 0; c < Env::S; ++c) asm volatile("" : "+v"(in[0].s[c][0]), "+v"(in[0].s[c][1]));
        }
        advance_and_store<Env, 2, AUTORESET, false, NT, false, 0, false>(a, (t + k * T) * 2, tick, in[k]);
    }
}

// ---------------------------------------------------------------------------------------------
// Producer / consumer form of the multi-lane kernel: the COMPUTING waves never issue a store.  A workgroup
// is 8 computing waves + 1 storing wave and walks TPB tiles of 512 lanes: a computing thread prefetches its lane of tile t + 1
// (the only memory operations it ever has in flight are loads, so its in-order vmcnt means what it says: "tile t has landed"
// while tile t + 1 is still on its way — the distance-1 software pipeline that loads and stores sharing one vmcnt forbids in
// step_kernel_pipe), advances its lane of tile t and hands the results to the storing wave through LDS ([row][512] floats,
// double-buffered, one s_barrier per tile); the storing wave drains tile t - 1 with 16-byte stores while tile t is computed.
// The first request burst is ONE lane per thread instead of ITEMS, so the first arithmetic starts earlier, and no computing
// wave ever stalls on the store path.  Same per-lane code and Philox counters as every other form: bit-identical.
// Needs n % 512 == 0 and 16-byte aligned rows (the launcher falls back to step_kernel_pipe otherwise).
// ---------------------------------------------------------------------------------------------
constexpr int kLdsTileMax = 512;     // lanes per tile of the widest workgroup shape (8 computing waves)

template <class Env>
constexpr int own_state_rows() { int c = 0; for (int k = 0; k < Env::S; ++k) c += state_row_own<Env>(k) ? 1 : 0; return c; }

template <class Env, int TPB, bool AUTORESET, int NT, int CW = 8>
__global__ __launch_bounds__(CW * 64 + 64) void step_kernel_lds(const StepArgs a, const int64_t tiles) {
    constexpr int kLdsComputeWaves = CW, kLdsTile = CW * 64;
    constexpr int S = Env::S, O = Env::O;
    constexpr int NOWN = own_state_rows<Env>();
    constexpr int NROW = NOWN + (Env::OBS_ALIASES_STATE ? 0 : O) + 1;          // own state rows, observation rows, reward
    constexpr bool NT_SS = (NT & 2) != 0, NT_O = (NT & 8) != 0;
    __shared__ float out_buf[2][NROW][kLdsTile];
    __shared__ uint8_t done_buf[2][kLdsTile];
    const uint64_t tick = a.tick2[a.parity];
    if (blockIdx.x == 0 && threadIdx.x == 0) a.tick2[a.parity ^ 1] = tick + 1;
    const int lane = threadIdx.x & 63;
    const bool storer = (threadIdx.x >> 6) == kLdsComputeWaves;
    const int cl = threadIdx.x;                                   // computing thread: its lane inside a tile
    const int64_t t0 = (int64_t)blockIdx.x * TPB;
    const int nt = (int)(tiles - t0 < TPB ? tiles - t0 : TPB);    // workgroup-uniform; >= 1 by the grid size

    // LDS hand-off between the computing waves and the storing wave: LDS traffic only (lgkmcnt), never vmcnt — a
    // __syncthreads() would also wait for the prefetched loads and for the storing wave's global stores
    auto tile_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    LaneInputs<Env, 1> in[TPB + 1];
    if (!storer) load_inputs<Env, 1, AUTORESET, NT, false>(a, t0 * kLdsTile + cl, in[0]);
#pragma unroll
    for (int s = 0; s <= TPB; ++s) {
        if (!storer) {
            if (s < TPB) {
                // Loads are UNCONDITIONAL (past this workgroup's last tile the lane re-reads its current tile): a branch around
                // a group of loads makes the compiler's waitcnt bookkeeping merge the two paths and wait for everything
                const int64_t i = (t0 + (s < nt ? s : nt - 1)) * kLdsTile + cl;
                const int64_t ip = (t0 + (s + 1 < nt ? s + 1 : nt - 1)) * kLdsTile + cl;
                load_inputs<Env, 1, AUTORESET, NT, false>(a, ip, in[s + 1]);
#pragma unroll
                for (int c = 0; c < S; ++c) asm volatile("" : "+v"(in[s].s[c][0]));     // this tile's inputs are needed now ...
                asm volatile("" : "+v"(in[s].act[0]));
                LaneOutputs<Env> out;
                compute_lane<Env, AUTORESET>(a, i, tick, in[s], out);
                // ... and no use of the NEXT tile's inputs may be scheduled before this tile's results exist (the compiler would
                // otherwise hoist e.g. the action's int -> float conversion and wait for the prefetch right after issuing it)
#pragma unroll
                for (int c = 0; c < S; ++c) asm volatile("" : "+v"(in[s + 1].s[c][0]), "+v"(out.s[S - 1]));
                asm volatile("" : "+v"(in[s + 1].act[0]), "+v"(out.reward));
                if (s < nt) {
                    int r = 0;
#pragma unroll
                    for (int k = 0; k < S; ++k)
                        if (state_row_own<Env>(k)) out_buf[s & 1][r++][cl] = out.s[k];
                    if constexpr (!Env::OBS_ALIASES_STATE) {
#pragma unroll
                        for (int k = 0; k < O; ++k) out_buf[s & 1][r++][cl] = out.o[k];
                    }
                    out_buf[s & 1][r][cl] = out.reward;
                    done_buf[s & 1][cl] = out.done;
                }
            }
        } else if (s >= 1 && s - 1 < nt) {
            const int b = (s - 1) & 1;
            const int64_t base = (t0 + s - 1) * kLdsTile;
            auto drain = [&](float *row, int r, bool nt_store) {
#pragma unroll
                for (int j = 0; j < kLdsTile / 256; ++j) {
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(&out_buf[b][r][j * 256 + lane * 4]);
                    f32x4 *dst = reinterpret_cast<f32x4 *>(row + base + j * 256 + lane * 4);
                    if (nt_store) __builtin_nontemporal_store(v, dst); else *dst = v;
                }
                static_assert(kLdsTile % 256 == 0, "the storing wave moves 256 floats per instruction");
            };
            int r = 0;
#pragma unroll
            for (int k = 0; k < S; ++k)
                if (state_row_own<Env>(k)) drain(a.state_out + k * a.state_stride, r++, NT_SS);
            if constexpr (!Env::OBS_ALIASES_STATE) {
#pragma unroll
                for (int k = 0; k < O; ++k) drain(a.obs + k * a.obs_stride, r++, NT_SS);
            }
            drain(a.reward, r, NT_O);
            if (lane < kLdsTile / 16) {
                const i32x4 v = *reinterpret_cast<const i32x4 *>(&done_buf[b][lane * 16]);
                i32x4 *dst = reinterpret_cast<i32x4 *>(a.done + base + lane * 16);
                if constexpr (NT_O) __builtin_nontemporal_store(v, dst); else *dst = v;
            }
        }
        tile_barrier();
    }
}

template <class Env, int VEC, bool AUTORESET, bool EXTRAS, int NT, int RESETF = 0>
__global__ __launch_bounds__(256) void step_kernel(const StepArgs a) {
    const int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * VEC;
    ResetScratch<Env> *sc = nullptr;
    if constexpr (RESETF == 1) {
        __shared__ ResetScratch<Env> scratch[256 / 64];            // one table per wave of the workgroup
        sc = &scratch[threadIdx.x >> 6];
    }
    // engine tick (Philox counter word): double-buffered in device memory so that a replayed
    // hipGraph, whose kernel arguments are frozen, still advances it.
    const uint64_t tick = a.tick2[a.parity];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        a.tick2[a.parity ^ 1] = tick + 1;
    }
    if constexpr (EXTRAS) {
        if (blockIdx.x == 0 && a.done_count2)     // zero the NEXT step launch's half of the shard counters
            for (int sh = threadIdx.x; sh < kShards; sh += blockDim.x) a.done_count2[(a.cparity ^ 1) * (kShards * kCountStride) + sh * kCountStride] = 0u;
    }
    // workgroup-uniform: every workgroup but (at most) the last runs the unguarded body
    if (((int64_t)blockIdx.x + 1) * blockDim.x * VEC <= a.n) {
        step_body<Env, VEC, AUTORESET, EXTRAS, NT, false, RESETF>(a, i0, tick, sc);
    } else {
        if (i0 >= a.n) return;
        step_body<Env, VEC, AUTORESET, EXTRAS, NT, true, RESETF>(a, i0, tick, sc);
    }
}

// ---------------------------------------------------------------------------------------------
// Fused rollout (the example's replay memory batched: ReplayMemory.cs:25-67): T vector steps in
// ONE launch.  Each thread keeps its VEC envs in registers for all T steps, so per env-step only the action is
// read (4 B) and — when recording — obs / reward / done are written (O*4 + 5 B): 25 B instead of 41 B for
// CartPole, and the load-phase / store-phase serialisation of the one-step kernel disappears.  The next step's
// action is loaded before the current step's math.  Results are bit-identical to T one-step launches.
// ---------------------------------------------------------------------------------------------
template <class Env, int VEC, bool AUTORESET, bool GUARD>
__device__ __forceinline__ void rollout_body(const StepArgs &a, const RolloutArgs &ro, const int64_t i0, const uint64_t tick0) {
    constexpr int S = Env::S, O = Env::O;
    using Act = typename Env::Action;
    const int64_t n = a.n;

    float s[S][VEC];
#pragma unroll
    for (int k = 0; k < S; ++k) load_f32<VEC, true, GUARD>(state_row_src<Env>(a.state, a.state_stride, a.obs_in, a.obs_stride, k), i0, n, s[k]);
    int32_t sbd[VEC] = {};
    if constexpr (!AUTORESET && Env::HAS_SBD) load_i32<VEC, true, GUARD>(a.sbd, i0, n, sbd);

    auto load_action = [&](int64_t slice, Act (&dst)[VEC]) {
        const char *base = static_cast<const char *>(a.action) + (size_t)(slice * ro.action_stride) * 4;
        if constexpr (Env::BOX_ACTION) load_f32<VEC, true, GUARD>(reinterpret_cast<const float *>(base), i0, n, dst);
        else load_i32<VEC, true, GUARD>(reinterpret_cast<const int32_t *>(base), i0, n, dst);
    };

    Act act[VEC], act_next[VEC];
    load_action(0, act);
    int64_t slice = 0;
    float reward[VEC];
    uint8_t done[VEC];
    float o[O][VEC];

    for (int64_t t = 0; t < ro.steps; ++t) {
        int64_t nslice = slice + 1;
        if (nslice == ro.ring) nslice = 0;
        if (t + 1 < ro.steps) load_action(nslice, act_next);        // in flight during this step's math
        uint32_t pending = 0;
        bool after[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) after[j] = false;
        // The sub-lanes are advanced by a loop written out HERE, not through advance_all(): the same arithmetic, but with the
        // helper's reward / done arrays in between LLVM turns the reset loop's dynamic sub-lane write-back into 60 compare +
        // select pairs per trip instead of a branch on the sub-lane index (221 vs 97 VALU per trip): 3.57 vs 2.48 us per step at
        // 2^20 CartPole lanes (profiles/forms_probe_r03.txt) — the round-2 regression VERDICT r2 asked about.
        if constexpr (!(Env::PACKED2 && VEC == 2)) {
            auto all_sublanes = [&](auto small_tag) {
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    float sj[S], oj[O], rw;
                    bool dn;
#pragma unroll
                    for (int k = 0; k < S; ++k) sj[k] = s[k][j];
                    advance_sublane<Env, AUTORESET, decltype(small_tag)::value>(sj, act[j], sbd[j], rw, dn, after[j], !GUARD || i0 + j < n, oj);
                    done[j] = dn ? 1 : 0;
                    reward[j] = rw;
                    if constexpr (AUTORESET) pending |= dn ? (1u << j) : 0u;
#pragma unroll
                    for (int k = 0; k < S; ++k) s[k][j] = sj[k];
                    if constexpr (!Env::OBS_ALIASES_STATE) {
#pragma unroll
                        for (int k = 0; k < O; ++k) o[k][j] = oj[k];
                    }
                }
            };
            if constexpr (Env::HAS_SMALL_ANGLE_PATH) {        // wave-uniform choice, bit-identical paths (envs.hpp sincos_tiny)
                if (wave_angles_small<Env, VEC>(s)) all_sublanes(std::true_type{});
                else all_sublanes(std::false_type{});
            } else {
                all_sublanes(std::false_type{});
            }
        } else {
            bool dnv[VEC];
            advance_all<Env, VEC, AUTORESET, GUARD>(s, act, sbd, reward, dnv, after, o, i0, n);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                done[j] = dnv[j] ? 1 : 0;
                if constexpr (AUTORESET) pending |= dnv[j] ? (1u << j) : 0u;
            }
        }
        if constexpr (!AUTORESET && Env::HAS_SBD) count_after_done<VEC>(a, after);
        if (ro.rec_reward) store_f32<VEC, true, GUARD>(ro.rec_reward + t * n, i0, n, reward);
        if (ro.rec_done) store_u8<VEC, true, GUARD>(ro.rec_done + t * n, i0, n, done);
        if constexpr (AUTORESET) reset_pending<Env, VEC, false>(pending, s, o, a, i0, n, tick0 + (uint64_t)t);
        if (ro.rec_obs) {
#pragma unroll
            for (int k = 0; k < O; ++k) {
                if constexpr (Env::OBS_ALIASES_STATE) store_f32<VEC, true, GUARD>(ro.rec_obs + (t * O + k) * n, i0, n, s[k]);
                else store_f32<VEC, true, GUARD>(ro.rec_obs + (t * O + k) * n, i0, n, o[k]);
            }
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) act[j] = act_next[j];
        slice = nslice;
    }

#pragma unroll
    for (int k = 0; k < S; ++k)
        if (state_row_own<Env>(k)) store_f32<VEC, false, GUARD>(a.state_out + k * a.state_stride, i0, n, s[k]);
    if constexpr (!Env::OBS_ALIASES_STATE) {
#pragma unroll
        for (int k = 0; k < O; ++k) store_f32<VEC, false, GUARD>(a.obs + k * a.obs_stride, i0, n, o[k]);
    }
    store_f32<VEC, false, GUARD>(a.reward, i0, n, reward);
    store_u8<VEC, false, GUARD>(a.done, i0, n, done);
    if constexpr (!AUTORESET && Env::HAS_SBD) store_i32<VEC, false, GUARD>(a.sbd, i0, n, sbd);
}

template <class Env, int VEC, bool AUTORESET>
__global__ __launch_bounds__(256) void rollout_kernel(const StepArgs a, const RolloutArgs ro) {
    const int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * VEC;
    const uint64_t tick0 = a.tick2[a.parity];
    if (blockIdx.x == 0 && threadIdx.x == 0) a.tick2[a.parity ^ 1] = tick0 + (uint64_t)ro.steps;
    if (((int64_t)blockIdx.x + 1) * blockDim.x * VEC <= a.n) {     // full workgroup: no bounds checks inside the T-step loop
        rollout_body<Env, VEC, AUTORESET, false>(a, ro, i0, tick0);
    } else {
        if (i0 >= a.n) return;
        rollout_body<Env, VEC, AUTORESET, true>(a, ro, i0, tick0);
    }
}

// ---------------------------------------------------------------------------------------------
// Reset: all lanes, or the lanes selected by a byte mask (the caller's `if (done) Reset()`).
// ---------------------------------------------------------------------------------------------
template <class Env>
__device__ __forceinline__ void reset_lane(const ResetArgs &a, int64_t i, uint64_t tick) {
    constexpr int S = Env::S, O = Env::O;
    const uint64_t key = a.lane_seed ? a.lane_seed[i] : a.seed;
    const PhiloxWords r = lane_words(key, a.lane_offset + (uint64_t)i, tick);
    float s[S];
    Env::reset(s, r);
#pragma unroll
    for (int k = 0; k < S; ++k)
        if (state_row_own<Env>(k)) a.state[k * a.state_stride + i] = s[k];
    if constexpr (!Env::OBS_ALIASES_STATE) {
        float o[O];
        Env::observe_fresh(s, o);
#pragma unroll
        for (int k = 0; k < O; ++k) a.obs[k * a.obs_stride + i] = o[k];
    }
    if (a.sbd) a.sbd[i] = -1;            // CartPoleEnv.cs:64
    if (a.done) a.done[i] = 0;
    if (a.ep_ret) { a.ep_ret[i] = 0.0f; a.ep_len[i] = 0; }
}

// One thread per 4 lanes: the mask is read as one 32-bit word, and a thread whose four lanes are all unselected
// (the common case for `if (done) Reset()`: ~4.5 % of lanes) exits after that single load.  a.mask may alias a.done:
// each lane's flag is read before the same thread clears it.
template <class Env>
__global__ __launch_bounds__(256) void reset_kernel(const ResetArgs a) {
    const int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const uint64_t tick = a.tick2[a.parity];
    if (blockIdx.x == 0 && threadIdx.x == 0) a.tick2[a.parity ^ 1] = tick + 1;
    if (i0 >= a.n) return;
    uint32_t m = 0x01010101u;
    if (a.mask) {
        if (i0 + 4 <= a.n && (reinterpret_cast<uintptr_t>(a.mask) & 3u) == 0) m = *reinterpret_cast<const uint32_t *>(a.mask + i0);
        else { m = 0; for (int j = 0; j < 4; ++j) if (i0 + j < a.n && a.mask[i0 + j]) m |= 1u << (8 * j); }
        if (m == 0) return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (((m >> (8 * j)) & 0xFFu) && i0 + j < a.n) reset_lane<Env>(a, i0 + j, tick);
}

template <class Env>
__global__ __launch_bounds__(256) void observe_kernel(const float *state, int64_t sstride, float *obs, int64_t ostride, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s[Env::S], o[Env::O];
#pragma unroll
    for (int k = 0; k < Env::S; ++k) s[k] = state_row_src<Env>(state, sstride, obs, ostride, k)[i];
    Env::observe(s, o);
#pragma unroll
    for (int k = 0; k < Env::O; ++k) obs[k * ostride + i] = o[k];
}

// SoA [O][stride] -> row-major [n][O] (the NDArray layout at the host boundary).  Reads are coalesced per
// component; each lane then writes its O contiguous floats, so a wave writes 64*O*4 contiguous bytes.
template <int O>
__global__ __launch_bounds__(256) void pack_obs_kernel(const float *__restrict__ obs, int64_t stride,
                                                       float *__restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v[O];
#pragma unroll
    for (int k = 0; k < O; ++k) v[k] = obs[k * stride + i];
    if constexpr (O == 4) {
        *reinterpret_cast<float4 *>(out + i * 4) = make_float4(v[0], v[1], v[2], v[3]);
    } else if constexpr (O % 2 == 0) {
#pragma unroll
        for (int k = 0; k < O; k += 2) *reinterpret_cast<float2 *>(out + i * O + k) = make_float2(v[k], v[k + 1]);
    } else {
#pragma unroll
        for (int k = 0; k < O; ++k) out[i * O + k] = v[k];
    }
}

// Gathers the kShards segments of one step's done list — and of the records written beside it — into compact arrays, and /
// or applies the records to the dense per-lane arrays (the "last finished episode of every lane" view).  One workgroup per
// shard; every workgroup recomputes the (tiny) exclusive scan of the 256 shard counts in LDS, then copies its segment coalesced.
__global__ __launch_bounds__(256) void compact_done_kernel(const CompactArgs a) {
    __shared__ uint32_t scan[kShards];
    const int t = threadIdx.x;
    const uint32_t mine = a.counts[t * kCountStride];
    scan[t] = mine;
    __syncthreads();
    for (int d = 1; d < kShards; d <<= 1) {            // Hillis-Steele inclusive scan, 8 rounds
        const uint32_t v = t >= d ? scan[t - d] : 0u;
        __syncthreads();
        scan[t] += v;
        __syncthreads();
    }
    const int shard = blockIdx.x;
    const uint32_t cnt = a.counts[shard * kCountStride];
    const uint32_t start = scan[shard] - cnt;
    if (shard == 0 && t == 0 && a.out_count) *a.out_count = scan[kShards - 1];
    const int64_t seg0 = (int64_t)shard * a.cap;
    const int O = a.obs_dim;
    for (uint32_t k = t; k < cnt; k += 256) {
        const int32_t lane = a.list[seg0 + k];
        const uint32_t dst = start + k;
        const bool fits = (int64_t)dst < a.out_capacity;
        if (a.out_list && fits) a.out_list[dst] = lane;
        if (a.rec_ret) {
            const float r = a.rec_ret[seg0 + k];
            const int32_t l = a.rec_len[seg0 + k];
            if (a.out_ret && fits) a.out_ret[dst] = r;
            if (a.out_len && fits) a.out_len[dst] = l;
            if (a.dense_ret) { a.dense_ret[lane] = r; a.dense_len[lane] = l; }
        }
        if (a.rec_obs) {
            for (int c = 0; c < O; ++c) {
                const float v = a.rec_obs[((int64_t)shard * O + c) * a.cap + k];
                if (a.out_obs && fits) a.out_obs[(int64_t)dst * O + c] = v;
                if (a.dense_obs) a.dense_obs[(int64_t)c * a.n + lane] = v;
            }
        }
    }
}

template <int O>
__global__ __launch_bounds__(256) void export_small_kernel(const float *__restrict__ obs, int64_t stride, const float *__restrict__ reward,
                                                           const uint8_t *__restrict__ done, float *out_obs, float *out_reward,
                                                           uint8_t *out_done, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
#pragma unroll
    for (int k = 0; k < O; ++k) out_obs[i * O + k] = obs[k * stride + i];
    if (out_reward) out_reward[i] = reward[i];
    if (out_done) out_done[i] = done[i];
}

// The host boundary without staging: observations (SoA -> row-major [n][O]), rewards and done flags written STRAIGHT into
// page-locked, device-mapped host memory (gymnet_vecenv_host_buffers) — the stores themselves are the PCIe transfer, no
// device-side pack buffer, no memcpy calls.  A thread owns 4 consecutive lanes: it transposes their 4 x O observation words in
// registers and writes them as O 16-byte stores to 16 * O contiguous bytes, so a wave writes one contiguous 1 KiB * O block
// (full PCIe write payloads for any O, including the 3- and 6-wide observations).
template <int O>
__global__ __launch_bounds__(256) void export_host_kernel(const float *__restrict__ obs, int64_t stride, const float *__restrict__ reward,
                                                          const uint8_t *__restrict__ done, float *__restrict__ out_obs,
                                                          float *__restrict__ out_reward, uint8_t *__restrict__ out_done, int64_t n,
                                                          int vec_ok) {
    const int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i0 >= n) return;
    if (vec_ok && i0 + 4 <= n) {
        if (out_obs) {
            float row[4 * O];
#pragma unroll
            for (int k = 0; k < O; ++k) {
                const f32x4 t = *reinterpret_cast<const f32x4 *>(obs + k * stride + i0);
                row[0 * O + k] = t.x; row[1 * O + k] = t.y; row[2 * O + k] = t.z; row[3 * O + k] = t.w;
            }
            f32x4 *dst = reinterpret_cast<f32x4 *>(out_obs + i0 * O);
#pragma unroll
            for (int q = 0; q < O; ++q) { f32x4 t; t.x = row[4 * q]; t.y = row[4 * q + 1]; t.z = row[4 * q + 2]; t.w = row[4 * q + 3]; dst[q] = t; }
        }
        if (out_reward) *reinterpret_cast<f32x4 *>(out_reward + i0) = *reinterpret_cast<const f32x4 *>(reward + i0);
        if (out_done) *reinterpret_cast<uint32_t *>(out_done + i0) = *reinterpret_cast<const uint32_t *>(done + i0);
    } else {
        for (int64_t i = i0; i < n && i < i0 + 4; ++i) {
            if (out_obs) for (int k = 0; k < O; ++k) out_obs[i * O + k] = obs[k * stride + i];
            if (out_reward) out_reward[i] = reward[i];
            if (out_done) out_done[i] = done[i];
        }
    }
}

__global__ __launch_bounds__(256) void fill_i32_kernel(int32_t *p, int32_t v, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// Discrete.Contains(int) (Discrete.cs:38-40) over a batch: counts actions outside [0, nvals)
__global__ __launch_bounds__(256) void validate_discrete_kernel(const int32_t *__restrict__ a, int64_t n, int32_t nvals,
                                                                uint32_t *bad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool invalid = i < n && (a[i] < 0 || a[i] >= nvals);
    const uint64_t m = __ballot(invalid);
    if (m && lane_id() == (uint32_t)(__ffsll((unsigned long long)m) - 1)) atomicAdd(bad, (uint32_t)__popcll(m));
}

// Discrete.Sample() (Discrete.cs:17-28, no mask): start + randint(0, n)
__global__ __launch_bounds__(256) void sample_discrete_kernel(int32_t *__restrict__ out, int64_t n, int32_t nvals,
                                                              int32_t start, uint64_t seed, uint64_t lane_offset,
                                                              uint64_t tick) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const PhiloxWords r = stream_words(kStreamAction, seed, lane_offset + (uint64_t)i, tick);
    out[i] = start + (int32_t)__umulhi(r.w[0], (uint32_t)nvals);
}

// Discrete.Sample(mask) (Discrete.cs:18-26): valid = nonzero(mask == 1); any -> start + valid[choice(len(valid))], none -> start.
// choice(k) = hi32(w0 * k) with the same Philox word the unmasked draw uses.  One row of `nvals` mask bytes per lane
// (mask_stride = nvals) or one shared row (mask_stride = 0).
__global__ __launch_bounds__(256) void sample_discrete_masked_kernel(int32_t *__restrict__ out, int64_t n, int32_t nvals, int32_t start,
                                                                     const uint8_t *__restrict__ mask, int64_t mask_stride,
                                                                     uint64_t seed, uint64_t lane_offset, uint64_t tick) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t *m = mask + i * mask_stride;
    int32_t valid = 0;
    for (int32_t k = 0; k < nvals; ++k) valid += m[k] == 1 ? 1 : 0;
    int32_t pick = 0;
    if (valid > 0) {
        const PhiloxWords r = stream_words(kStreamAction, seed, lane_offset + (uint64_t)i, tick);
        int32_t want = (int32_t)__umulhi(r.w[0], (uint32_t)valid);     // index into the list of valid actions
        for (int32_t k = 0; k < nvals; ++k) {
            if (m[k] == 1) { if (want == 0) { pick = k; break; } --want; }
        }
    }
    out[i] = start + pick;
}

// The caller's epsilon-greedy composer (examples/.../PlaySessions/TrainingPlaySession.cs:46-52), batched:
//   if (Random.NextDouble() <= epsilon) action = ActionSpace.Sample(); else action = policy action
// Lane i uses the ACTION stream of Philox(seed, (lane_offset + i, tick)): word 0 is the sampled action (identical to sample_discrete_kernel),
// word 1 the 24-bit uniform that is compared with epsilon.
__global__ __launch_bounds__(256) void compose_discrete_kernel(const int32_t *__restrict__ policy, int32_t *__restrict__ out,
                                                               int64_t n, int32_t nvals, float epsilon, uint64_t seed,
                                                               uint64_t lane_offset, uint64_t tick) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const PhiloxWords r = stream_words(kStreamAction, seed, lane_offset + (uint64_t)i, tick);
    const bool explore = u01_24(r.w[1]) <= epsilon;
    out[i] = explore ? (int32_t)__umulhi(r.w[0], (uint32_t)nvals) : policy[i];
}

// Box.Sample() (Box.cs:69-90): the reference's four regimes, selected by which bounds are finite
__global__ __launch_bounds__(256) void sample_box_kernel(float *__restrict__ out, int64_t n, float low, float high,
                                                         uint64_t seed, uint64_t lane_offset, uint64_t tick) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const PhiloxWords r = stream_words(kStreamAction, seed, lane_offset + (uint64_t)i, tick);
    const bool blo = low > -INFINITY, bhi = high < INFINITY;     // Box.CheckBounded (Box.cs:53-58)
    const float u = u01_24(r.w[0]);
    float v;
    if (blo && bhi) {
        v = low + (high - low) * u;                               // Box.cs:85 uniform(low, high)
    } else if (blo) {
        v = -logf(1.0f - u) + low;                                // Box.cs:83 exponential(1) + low
    } else if (bhi) {
        v = -logf(1.0f - u) + high;                               // Box.cs:84 exponential(1) + high (sic)
    } else {
        const float u1 = (float)((r.w[0] >> 8) + 1u) * (1.0f / 16777216.0f);   // (0, 1]
        const float u2 = u01_24(r.w[1]);
        v = 0.5f + sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);    // Box.cs:82 normal(0.5, 1) (sic)
    }
    out[i] = v;
}

// Box.Sample() for a Box whose Low / High are ARRAYS (Box.cs:25-51): the regime is chosen PER ELEMENT from that element's own
// bounds (Box.cs:74-85: unbounded / low-bounded / high-bounded / bounded masks), `dim` elements per lane, output row-major
// [count][dim].  Element e of lane i draws from Philox(key = action-stream key + e * odd constant, counter = (lane, tick)):
// element 0 uses exactly the words of the scalar sampler above, so a (1,)-shaped Box samples the same values either way.
__global__ __launch_bounds__(256) void sample_box_elementwise_kernel(float *__restrict__ out, int64_t n, int32_t dim,
                                                                     const float *__restrict__ low, const float *__restrict__ high,
                                                                     uint64_t seed, uint64_t lane_offset, uint64_t tick) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // one thread per ELEMENT: coalesced row-major stores
    if (idx >= n * dim) return;
    const int64_t i = idx / dim;
    const int32_t e = (int32_t)(idx - i * dim);
    const PhiloxWords r = stream_words(kStreamAction, seed + (uint64_t)e * 0xD1B54A32D192ED03ull, lane_offset + (uint64_t)i, tick);
    const float lo = low[e], hi = high[e];
    const bool blo = lo > -INFINITY, bhi = hi < INFINITY;           // Box.CheckBounded (Box.cs:53-58)
    const float u = u01_24(r.w[0]);
    float v;
    if (blo && bhi) {
        v = lo + (hi - lo) * u;                                     // Box.cs:85 uniform(low, high)
    } else if (blo) {
        v = -logf(1.0f - u) + lo;                                   // Box.cs:83 exponential(1) + low
    } else if (bhi) {
        v = -logf(1.0f - u) + hi;                                   // Box.cs:84 exponential(1) + high (sic)
    } else {
        const float u1 = (float)((r.w[0] >> 8) + 1u) * (1.0f / 16777216.0f);   // (0, 1]
        const float u2 = u01_24(r.w[1]);
        v = 0.5f + sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);    // Box.cs:82 normal(0.5, 1) (sic)
    }
    out[idx] = v;
}

// Direct (full-mesh) all-gather of observations, push form: this member's slice [D][N/G] is stored into
// the same offset of every peer's replica buffer.  blockIdx.y selects the peer, so all peers' links carry traffic
// concurrently (xGMI is point-to-point: 7 links x ~153 GB/s, one per peer); each lane moves 16 bytes per trip.  Plain stores:
// the bytes have to leave this GPU anyway, and the kernel boundary is the release the peers' next kernels acquire against.
__global__ __launch_bounds__(256) void push_obs_kernel(const PushArgs a) {
    float *__restrict__ dst = a.dst[blockIdx.y];
    const float *__restrict__ src = a.src;
    const int64_t nvec = a.count >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const bool vec_ok = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15u) == 0;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (vec_ok) {
        for (int64_t v = i; v < nvec; v += stride)
            reinterpret_cast<f32x4 *>(dst)[v] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(src) + v);
        for (int64_t k = (nvec << 2) + i; k < a.count; k += stride) dst[k] = src[k];
    } else {
        for (int64_t k = i; k < a.count; k += stride) dst[k] = src[k];
    }
}

// ---------------------------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------------------------
static inline unsigned grid_for(int64_t items, int block) { return (unsigned)((items + block - 1) / block); }

// Which instantiation a step launch resolves to: ONE function decides, the launcher dispatches on it and
// describe_step_kernel() prints it (gymnet_vecenv_kernel_name: tests and bench.py name the kernel they ran from the library,
// not from a copy of this policy).
struct StepVariant {
    int lds_tiles;      // > 1: step_kernel_lds<Env, lds_tiles, AUTORESET, 15> (producer / consumer form of the multi-lane kernel)
    int pipe_items;     // > 1: step_kernel_pipe<Env, pipe_items, AUTORESET, 15> (or step_kernel_pipe2 with pipe_pairs); else step_kernel
    bool pipe_pairs = false;   // the multi-lane kernel over lane PAIRS (step_kernel_pipe2)
    int vec, nt;        // step_kernel<Env, vec, AUTORESET, EXTRAS, nt, resetf>
    int resetf;
};

static LaunchCfg normalized(LaunchCfg cfg) {
    if (cfg.vec != 4 && cfg.vec != 2) cfg.vec = 1;
    if (cfg.block != 64 && cfg.block != 128) cfg.block = 256;
    if (cfg.nt != 12 && cfg.nt != 15) cfg.nt = 0;
    return cfg;
}

template <class Env>
static StepVariant resolve_variant(bool autoreset, bool extras, const LaunchCfg &cfg, int64_t n) {
    StepVariant v{};
    v.lds_tiles = 1; v.pipe_items = 1; v.vec = 1; v.nt = cfg.nt; v.resetf = 0;
    if constexpr (Env::PIPELINED) {        // multi-lane kernel, cfg.items lanes per thread (2..5)
        if (cfg.items > 1 && cfg.items <= 5 && !extras && cfg.vec == 1) {
            v.nt = 15;
            if (cfg.lds_pipe) v.lds_tiles = cfg.items; else v.pipe_items = cfg.items;
            return v;
        }
        // lane pairs (probe form): 2..4 pairs per thread, whole batches only — otherwise the ordinary forms below
        if (cfg.items > 1 && cfg.items <= 4 && !extras && cfg.vec == 2 && !cfg.lds_pipe && n > 0 && n % (2 * (int64_t)cfg.items * 256) == 0) {
            v.nt = 15; v.vec = 2; v.pipe_items = cfg.items; v.pipe_pairs = true;
            return v;
        }
    }
    // the wide form of an env: four lanes per thread on dwordx4 streams, or — for the env with a two-lane packed-FP32 form
    // (Acrobot) — two lanes per thread on dwordx2 streams
    if (cfg.vec > 1) v.vec = Env::PACKED2 ? 2 : 4;
    // wave-compacted fused reset: lean dwordx4 variant of an env whose observation IS its state
    if (Env::OBS_ALIASES_STATE && !Env::PACKED2 && cfg.reset_form == 1 && v.vec == 4 && autoreset) v.resetf = 1;
    return v;
}

template <class Env>
static hipError_t launch_step_env(bool autoreset, bool extras, const StepArgs &a, LaunchCfg cfg, hipStream_t st) {
    const StepVariant v = resolve_variant<Env>(autoreset, extras, cfg, a.n);
    if constexpr (Env::PIPELINED) {
        if (v.pipe_pairs) {
            const dim3 qgrid((unsigned)(a.n / (2 * (int64_t)v.pipe_items * 256))), qblk(256);
#define GYMNET_PIPE2(I)                                                                                                 \
    case I:                                                                                                             \
        if (autoreset) hipLaunchKernelGGL((step_kernel_pipe2<Env, I, true, 15>), qgrid, qblk, 0, st, a);                 \
        else hipLaunchKernelGGL((step_kernel_pipe2<Env, I, false, 15>), qgrid, qblk, 0, st, a);                          \
        break;
            switch (v.pipe_items) {
                GYMNET_PIPE2(2) GYMNET_PIPE2(3) GYMNET_PIPE2(4)
                default: return hipErrorInvalidValue;
            }
#undef GYMNET_PIPE2
            return hipGetLastError();
        }
        if (v.lds_tiles > 1) {
            const int64_t tiles = a.n / kLdsTileMax;
            const dim3 lgrid(grid_for(tiles, v.lds_tiles)), lblk(kLdsTileMax + 64);
#define GYMNET_LDS(I)                                                                                                   \
    case I:                                                                                                             \
        if (autoreset) hipLaunchKernelGGL((step_kernel_lds<Env, I, true, 15>), lgrid, lblk, 0, st, a, tiles);            \
        else hipLaunchKernelGGL((step_kernel_lds<Env, I, false, 15>), lgrid, lblk, 0, st, a, tiles);                     \
        break;
            switch (v.lds_tiles) {
                GYMNET_LDS(2) GYMNET_LDS(3) GYMNET_LDS(4) GYMNET_LDS(5)
                default: return hipErrorInvalidValue;
            }
#undef GYMNET_LDS
            return hipGetLastError();
        }
        if (v.pipe_items > 1) {
            const int64_t per_block = 256 * (int64_t)v.pipe_items;
            const dim3 pgrid(grid_for(a.n > 0 ? (a.n + per_block - 1) / per_block : 1, 1)), pblk(256);
#define GYMNET_PIPE(I)                                                                                                  \
    case I:                                                                                                             \
        if (autoreset) hipLaunchKernelGGL((step_kernel_pipe<Env, I, true, 15>), pgrid, pblk, 0, st, a);                  \
        else hipLaunchKernelGGL((step_kernel_pipe<Env, I, false, 15>), pgrid, pblk, 0, st, a);                           \
        break;
            switch (v.pipe_items) {
                GYMNET_PIPE(2) GYMNET_PIPE(3) GYMNET_PIPE(4) GYMNET_PIPE(5)
                default: return hipErrorInvalidValue;
            }
#undef GYMNET_PIPE
            return hipGetLastError();
        }
    }
    constexpr int WIDE = Env::PACKED2 ? 2 : 4;
    const bool wide = v.vec > 1;
    const int64_t threads = (a.n + v.vec - 1) / v.vec;
    const dim3 grid(grid_for(threads > 0 ? threads : 1, cfg.block)), blk(cfg.block);
#define GYMNET_LAUNCH(V, AR, EX, NTM) hipLaunchKernelGGL((step_kernel<Env, V, AR, EX, NTM>), grid, blk, (size_t)cfg.lds_bytes, st, a)
#define GYMNET_LAUNCH_NT(V, AR, EX)                                   \
    do {                                                              \
        if (v.nt == 15) GYMNET_LAUNCH(V, AR, EX, 15);                 \
        else if (v.nt == 12) GYMNET_LAUNCH(V, AR, EX, 12);            \
        else GYMNET_LAUNCH(V, AR, EX, 0);                             \
    } while (0)
    if constexpr (Env::OBS_ALIASES_STATE && !Env::PACKED2) {
        if (v.resetf == 1) {
#define GYMNET_LAUNCH_RF(EX)                                                                                                          \
    do {                                                                                                                              \
        if (v.nt == 15) hipLaunchKernelGGL((step_kernel<Env, 4, true, EX, 15, 1>), grid, blk, (size_t)cfg.lds_bytes, st, a);           \
        else if (v.nt == 12) hipLaunchKernelGGL((step_kernel<Env, 4, true, EX, 12, 1>), grid, blk, (size_t)cfg.lds_bytes, st, a);      \
        else hipLaunchKernelGGL((step_kernel<Env, 4, true, EX, 0, 1>), grid, blk, (size_t)cfg.lds_bytes, st, a);                       \
    } while (0)
            if (extras) GYMNET_LAUNCH_RF(true); else GYMNET_LAUNCH_RF(false);
#undef GYMNET_LAUNCH_RF
            return hipGetLastError();
        }
    }
    if (extras) {   // bookkeeping variants follow the same stream policy (their own arrays stay cacheable)
        if (wide) { if (autoreset) GYMNET_LAUNCH_NT(WIDE, true, true); else GYMNET_LAUNCH_NT(WIDE, false, true); }
        else      { if (autoreset) GYMNET_LAUNCH_NT(1, true, true); else GYMNET_LAUNCH_NT(1, false, true); }
    } else if (wide) {
        if (autoreset) GYMNET_LAUNCH_NT(WIDE, true, false); else GYMNET_LAUNCH_NT(WIDE, false, false);
    } else {
        if (autoreset) GYMNET_LAUNCH_NT(1, true, false); else GYMNET_LAUNCH_NT(1, false, false);
    }
#undef GYMNET_LAUNCH_NT
#undef GYMNET_LAUNCH
    return hipGetLastError();
}

hipError_t launch_step(int env_id, bool autoreset, bool extras, const StepArgs &a, LaunchCfg cfg, hipStream_t st) {
    cfg = normalized(cfg);
    switch (env_id) {
        case 0: return launch_step_env<CartPole>(autoreset, extras, a, cfg, st);
        case 1: return launch_step_env<Pendulum>(autoreset, extras, a, cfg, st);
        case 2: return launch_step_env<MountainCar>(autoreset, extras, a, cfg, st);
        case 3: return launch_step_env<Acrobot>(autoreset, extras, a, cfg, st);
        default: return hipErrorInvalidValue;
    }
}

int describe_step_kernel(int env_id, bool autoreset, bool extras, LaunchCfg cfg, int64_t n, char *buf, size_t cap) {
    cfg = normalized(cfg);
    StepVariant v;
    const char *env;
    switch (env_id) {
        case 0: v = resolve_variant<CartPole>(autoreset, extras, cfg, n); env = "CartPole"; break;
        case 1: v = resolve_variant<Pendulum>(autoreset, extras, cfg, n); env = "Pendulum"; break;
        case 2: v = resolve_variant<MountainCar>(autoreset, extras, cfg, n); env = "MountainCar"; break;
        case 3: v = resolve_variant<Acrobot>(autoreset, extras, cfg, n); env = "Acrobot"; break;
        default: return -1;
    }
    const char *ar = autoreset ? "true" : "false";
    if (v.pipe_pairs) return std::snprintf(buf, cap, "step_kernel_pipe2<%s,%d,%s,15>", env, v.pipe_items, ar);
    if (v.lds_tiles > 1) return std::snprintf(buf, cap, "step_kernel_lds<%s,%d,%s,15>", env, v.lds_tiles, ar);
    if (v.pipe_items > 1) return std::snprintf(buf, cap, "step_kernel_pipe<%s,%d,%s,15>", env, v.pipe_items, ar);
    return std::snprintf(buf, cap, "step_kernel<%s,%d,%s,%s,%d,%d>", env, v.vec, ar, extras ? "true" : "false", v.nt, v.resetf);
}

template <class Env>
static hipError_t launch_rollout_env(bool autoreset, const StepArgs &a, const RolloutArgs &r, LaunchCfg cfg, hipStream_t st) {
    constexpr int WIDE = Env::PACKED2 ? 2 : 4;
    const bool wide = cfg.vec > 1;
    const int64_t threads = (a.n + (wide ? WIDE : 1) - 1) / (wide ? WIDE : 1);
    const dim3 grid(grid_for(threads > 0 ? threads : 1, 256)), blk(256);
    if (wide) {
        if (autoreset) hipLaunchKernelGGL((rollout_kernel<Env, WIDE, true>), grid, blk, 0, st, a, r);
        else hipLaunchKernelGGL((rollout_kernel<Env, WIDE, false>), grid, blk, 0, st, a, r);
    } else {
        if (autoreset) hipLaunchKernelGGL((rollout_kernel<Env, 1, true>), grid, blk, 0, st, a, r);
        else hipLaunchKernelGGL((rollout_kernel<Env, 1, false>), grid, blk, 0, st, a, r);
    }
    return hipGetLastError();
}

hipError_t launch_rollout_fused(int env_id, bool autoreset, const StepArgs &a, const RolloutArgs &r, LaunchCfg cfg, hipStream_t st) {
    if (cfg.vec != 4 && cfg.vec != 2) cfg.vec = 1;
    switch (env_id) {
        case 0: return launch_rollout_env<CartPole>(autoreset, a, r, cfg, st);
        case 1: return launch_rollout_env<Pendulum>(autoreset, a, r, cfg, st);
        case 2: return launch_rollout_env<MountainCar>(autoreset, a, r, cfg, st);
        case 3: return launch_rollout_env<Acrobot>(autoreset, a, r, cfg, st);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_reset(int env_id, const ResetArgs &a, hipStream_t st) {
    const dim3 grid(grid_for(a.n > 0 ? (a.n + 3) / 4 : 1, 256)), blk(256);
    switch (env_id) {
        case 0: hipLaunchKernelGGL(reset_kernel<CartPole>, grid, blk, 0, st, a); break;
        case 1: hipLaunchKernelGGL(reset_kernel<Pendulum>, grid, blk, 0, st, a); break;
        case 2: hipLaunchKernelGGL(reset_kernel<MountainCar>, grid, blk, 0, st, a); break;
        case 3: hipLaunchKernelGGL(reset_kernel<Acrobot>, grid, blk, 0, st, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_observe(int env_id, const float *state, int64_t sstride, float *obs, int64_t ostride, int64_t n,
                          hipStream_t st) {
    if (n <= 0) return hipSuccess;
    const dim3 grid(grid_for(n, 256)), blk(256);
    switch (env_id) {
        case 1: hipLaunchKernelGGL(observe_kernel<Pendulum>, grid, blk, 0, st, state, sstride, obs, ostride, n); break;
        case 3: hipLaunchKernelGGL(observe_kernel<Acrobot>, grid, blk, 0, st, state, sstride, obs, ostride, n); break;
        default: return hipSuccess;   // aliasing envs: nothing to recompute
    }
    return hipGetLastError();
}

hipError_t launch_pack_obs(int obs_dim, const float *obs, int64_t stride, float *out, int64_t n, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    const dim3 grid(grid_for(n, 256)), blk(256);
    switch (obs_dim) {
        case 2: hipLaunchKernelGGL(pack_obs_kernel<2>, grid, blk, 0, st, obs, stride, out, n); break;
        case 3: hipLaunchKernelGGL(pack_obs_kernel<3>, grid, blk, 0, st, obs, stride, out, n); break;
        case 4: hipLaunchKernelGGL(pack_obs_kernel<4>, grid, blk, 0, st, obs, stride, out, n); break;
        case 6: hipLaunchKernelGGL(pack_obs_kernel<6>, grid, blk, 0, st, obs, stride, out, n); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_export_small(int obs_dim, const float *obs, int64_t stride, const float *reward, const uint8_t *done,
                               float *out_obs, float *out_reward, uint8_t *out_done, int64_t n, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    const dim3 grid(grid_for(n, 256)), blk(256);
    switch (obs_dim) {
        case 2: hipLaunchKernelGGL(export_small_kernel<2>, grid, blk, 0, st, obs, stride, reward, done, out_obs, out_reward, out_done, n); break;
        case 3: hipLaunchKernelGGL(export_small_kernel<3>, grid, blk, 0, st, obs, stride, reward, done, out_obs, out_reward, out_done, n); break;
        case 4: hipLaunchKernelGGL(export_small_kernel<4>, grid, blk, 0, st, obs, stride, reward, done, out_obs, out_reward, out_done, n); break;
        case 6: hipLaunchKernelGGL(export_small_kernel<6>, grid, blk, 0, st, obs, stride, reward, done, out_obs, out_reward, out_done, n); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_export_host(int obs_dim, const float *obs, int64_t stride, const float *reward, const uint8_t *done,
                              float *out_obs, float *out_reward, uint8_t *out_done, int64_t n, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    auto al = [](const void *p, uintptr_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; };
    const int vec_ok = al(obs, 16) && stride % 4 == 0 && al(reward, 16) && al(done, 4) && (!out_obs || al(out_obs, 16)) &&
                       (!out_reward || al(out_reward, 16)) && (!out_done || al(out_done, 4));
    const dim3 grid(grid_for((n + 3) / 4, 256)), blk(256);
    switch (obs_dim) {
        case 2: hipLaunchKernelGGL(export_host_kernel<2>, grid, blk, 0, st, obs, stride, reward, done, out_obs, out_reward, out_done, n, vec_ok); break;
        case 3: hipLaunchKernelGGL(export_host_kernel<3>, grid, blk, 0, st, obs, stride, reward, done, out_obs, out_reward, out_done, n, vec_ok); break;
        case 4: hipLaunchKernelGGL(export_host_kernel<4>, grid, blk, 0, st, obs, stride, reward, done, out_obs, out_reward, out_done, n, vec_ok); break;
        case 6: hipLaunchKernelGGL(export_host_kernel<6>, grid, blk, 0, st, obs, stride, reward, done, out_obs, out_reward, out_done, n, vec_ok); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_fill_i32(int32_t *p, int32_t v, int64_t n, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(fill_i32_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, p, v, n);
    return hipGetLastError();
}

hipError_t launch_compact_done(const CompactArgs &a, hipStream_t st) {
    hipLaunchKernelGGL(compact_done_kernel, dim3(kShards), dim3(256), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_validate_discrete(const int32_t *a, int64_t n, int32_t nvals, uint32_t *bad, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(validate_discrete_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, a, n, nvals, bad);
    return hipGetLastError();
}

hipError_t launch_sample_discrete(int32_t *out, int64_t n, int32_t nvals, int32_t start, uint64_t seed,
                                  uint64_t lane_offset, uint64_t tick, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(sample_discrete_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, out, n, nvals, start, seed,
                       lane_offset, tick);
    return hipGetLastError();
}

hipError_t launch_sample_discrete_masked(int32_t *out, int64_t n, int32_t nvals, int32_t start, const uint8_t *mask,
                                         int64_t mask_stride, uint64_t seed, uint64_t lane_offset, uint64_t tick, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(sample_discrete_masked_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, out, n, nvals, start, mask,
                       mask_stride, seed, lane_offset, tick);
    return hipGetLastError();
}

hipError_t launch_push_obs(const PushArgs &a, hipStream_t st) {
    if (a.count <= 0 || a.npeers <= 0) return hipSuccess;
    // enough workgroups per peer to keep a link busy, few enough that G-1 peers do not oversubscribe the chip
    int64_t per_peer = (a.count / 4 + 255) / 256;
    if (per_peer > 256) per_peer = 256;
    if (per_peer < 1) per_peer = 1;
    hipLaunchKernelGGL(push_obs_kernel, dim3((unsigned)per_peer, (unsigned)a.npeers), dim3(256), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_sample_box_elementwise(float *out, int64_t n, int32_t dim, const float *low, const float *high, uint64_t seed,
                                         uint64_t lane_offset, uint64_t tick, hipStream_t st) {
    if (n <= 0 || dim <= 0) return hipSuccess;
    hipLaunchKernelGGL(sample_box_elementwise_kernel, dim3(grid_for(n * dim, 256)), dim3(256), 0, st, out, n, dim, low, high, seed,
                       lane_offset, tick);
    return hipGetLastError();
}

hipError_t launch_compose_discrete(const int32_t *policy, int32_t *out, int64_t n, int32_t nvals, float epsilon, uint64_t seed,
                                   uint64_t lane_offset, uint64_t tick, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(compose_discrete_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, policy, out, n, nvals, epsilon, seed,
                       lane_offset, tick);
    return hipGetLastError();
}

hipError_t launch_sample_box(float *out, int64_t n, float low, float high, uint64_t seed, uint64_t lane_offset,
                             uint64_t tick, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(sample_box_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, out, n, low, high, seed,
                       lane_offset, tick);
    return hipGetLastError();
}

}  // namespace gymnet
