// step_kernels.hpp — the env-dependent HIP kernels of the batched classic-control engine (vector step, fused T-step rollout,
// reset, observation rebuild) as templates over an Env (envs.hpp, cartpole64.hpp), written for gfx950 (CDNA4, wave64), and their
// launchers.  Each env is instantiated in a translation unit of its own (env_*.hip), so the build runs them side by side.
//
// ONE skeleton for both state scalars: `typename Env::Real` is float for the structure-of-arrays hot path and
// double for GYMNET_FLAG_F64 (CartPole64: the reference's own arithmetic, CartPoleEnv.cs:141-166,185).  Everything around the
// per-lane arithmetic — lane I/O, fused auto-reset, steps_beyond_done, episode bookkeeping, done-list compaction, terminal
// observations, double buffering, external observation buffers, the multi-lane forms, the fused rollout — is written once and
// is therefore available in both.
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
#pragma once
#include "kernels.hpp"

#include <cstdio>
#include <type_traits>

#include "lanes.hpp"
#include "philox.hpp"

namespace gymnet {

__device__ __forceinline__ float abs_real(float x) { return fabsf(x); }
__device__ __forceinline__ double abs_real(double x) { return __builtin_fabs(x); }

// The reset draw of one lane: Philox4x32-10 keyed by (seed | per-lane seed), counter = (GLOBAL lane id, engine tick).  float32 envs
// take the four words of ONE call; CartPole64 makes its own two calls (53-bit uniforms, cartpole64.hpp).
// UNIFORM_KEY: the key is the handle's one seed (no per-lane keys in this kernel variant) — philox.hpp on what that buys.
template <class Env, bool UNIFORM_KEY = false>
__device__ __forceinline__ void draw_reset(typename Env::Real (&s)[Env::S], uint64_t key, uint64_t lane, uint64_t tick) {
    if constexpr (Env::RESET_TAKES_KEY) Env::reset(s, key, lane, tick);
    else Env::reset(s, lane_words<UNIFORM_KEY>(key, lane, tick));
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
__device__ __forceinline__ const typename Env::Real *state_row_src(const typename Env::Real *state, int64_t state_stride,
                                                                    const typename Env::Real *obs_in, int64_t obs_stride, int k) {
    if constexpr (Env::OBS_ALIASES_STATE) return state + k * state_stride;
    else return Env::OBS_ROW_OF_STATE[k] < 0 ? state + k * state_stride : obs_in + Env::OBS_ROW_OF_STATE[k] * obs_stride;
}

// shard of the calling wave for the sharded counters / done list (StepArgs)
__device__ __forceinline__ uint32_t wave_shard() {
    return (uint32_t)((((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6) & (kShards - 1));
}

// one 64-bit atomic per WAVE (and only if the wave has something to report) into the wave's shard
template <int VEC, class Args>
__device__ __forceinline__ void count_after_done(const Args &a, const bool (&after)[VEC]) {
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
__device__ __forceinline__ void advance_sublane(typename Env::Real (&sj)[Env::S], typename Env::Action act, int32_t &sbd, float &rw,
                                                bool &dn, bool &after, bool in_range, typename Env::Real (&oj)[Env::O]) {
    if constexpr (Env::HAS_SMALL_ANGLE_PATH) Env::template step<SMALL_ANGLE, AUTORESET>(sj, act, rw, dn);
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
__device__ __forceinline__ bool wave_angles_small(const typename Env::Real (&s)[Env::S][VEC]) {
    if constexpr (!Env::HAS_SMALL_ANGLE_PATH) return false;
    else {
        bool small = true;
#pragma unroll
        for (int j = 0; j < VEC; ++j) small = small && (abs_real(s[Env::ANGLE_ROW][j]) <= Env::SMALL_ANGLE_BOUND);
        return __ballot(!small) == 0;
    }
}

// One env-step of ALL VEC sub-lanes of a thread.  Generic: sub-lane after sub-lane.  Envs that provide a two-lane packed
// form (Acrobot: both envs of a thread ride the v_pk_*_f32 instructions, envs.hpp) take it when VEC == 2; per element the
// arithmetic is the same IEEE sequence, so the results are bit-identical to the sub-lane loop.
template <class Env, int VEC, bool AUTORESET, bool GUARD, bool PACK = true>
__device__ __forceinline__ void advance_all(typename Env::Real (&s)[Env::S][VEC], typename Env::Action (&act)[VEC], int32_t (&sbd)[VEC],
                                            float (&rw)[VEC], bool (&dn)[VEC], bool (&after)[VEC], typename Env::Real (&o)[Env::O][VEC],
                                            int64_t i0, int64_t n) {
    constexpr int S = Env::S, O = Env::O;
    using Real = typename Env::Real;
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
                Real sj[S], oj[O];
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
__device__ __forceinline__ void reset_pending(uint32_t pending, typename Env::Real (&s)[Env::S][VEC], typename Env::Real (&o)[Env::O][VEC],
                                              const StepArgsT<typename Env::Real> &a, int64_t i0, int64_t n, uint64_t tick) {
    constexpr int S = Env::S, O = Env::O;
    using Real = typename Env::Real;
    while (pending) {
        const int j = __ffs(pending) - 1;
        pending &= pending - 1;
        // (a bookkeeping kernel serves per-lane keys only when VecEnv.Seed(int[]) installed some: the handle's ONE seed otherwise — a
        // kernel-uniform choice, and the uniform-key form of the call keeps its round keys out of the vector registers: philox.hpp)
        Real sj[S];
        bool drawn = false;
        if constexpr (LANE_SEEDS) {
            if (a.lane_seed) {
                draw_reset<Env, false>(sj, i0 + j < n ? a.lane_seed[i0 + j] : a.seed, a.lane_offset + (uint64_t)(i0 + j), tick);
                drawn = true;
            }
        }
        if (!drawn) draw_reset<Env, true>(sj, a.seed, a.lane_offset + (uint64_t)(i0 + j), tick);
        Real oj[O];
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
// Envs whose reset is TWO Philox calls (float64 CartPole: eight words for four 53-bit uniforms, cartpole64.hpp) spread one reset
// over two lanes in the compacted forms: lane 2r makes call 0, lane 2r + 1 call 1, a DPP exchange joins the halves and the even
// lane converts — a Philox pass of the wave then serves 32 resets instead of costing two passes for 64.
template <class Env>
constexpr bool has_split_reset() { return Env::RESET_TAKES_KEY && Env::OBS_ALIASES_STATE; }

// the words of the odd neighbour (quad_perm [1, 0, 3, 2]): what lane 2r needs from lane 2r + 1
__device__ __forceinline__ PhiloxWords neighbour_words(const PhiloxWords &w) {
    PhiloxWords o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o.w[k] = (uint32_t)__builtin_amdgcn_mov_dpp((int)w.w[k], 0xB1, 0xF, 0xF, true);
    return o;
}

// Round 6, tools/issue_rate_probe.hip (profiles/issue_rate_r06.txt): on gfx950 a VALU instruction executed under an exec mask with 16 or
// fewer ACTIVE lanes — any 16: the first, every fourth, lanes 16..31 — holds the SIMD ~21 cycles instead of ~4.4 (17 lanes: 4.4).  The
// compacted reset's draw ran with the wave's ~11 finished slots as its only active lanes.  With this constant every active lane of the
// wave makes a draw (the ones without a slot repeat slot 0's and park it in a row nobody reads): same words for the slots, full mask.
#ifdef GYMNET_PROBE_RESET_MASKED       // probe builds only: the round 3-5 form, for the A/B
constexpr bool kResetFullExec = false;
#else
constexpr bool kResetFullExec = true;
#endif

template <class Env>
struct ResetScratch {
    uint32_t slot[64];              // rank -> owner lane * VEC + sub-lane
    typename Env::Real draw[64][Env::S];   // rank -> the drawn state
};

__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <class Env, int VEC, bool LANE_SEEDS>
__device__ __forceinline__ void reset_pending_wave(uint32_t pending, typename Env::Real (&s)[Env::S][VEC], const StepArgsT<typename Env::Real> &a,
                                                   int64_t i0, int64_t n, uint64_t tick, ResetScratch<Env> *sc) {
    constexpr int S = Env::S;
    using Real = typename Env::Real;
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
    constexpr bool SPLIT = has_split_reset<Env>();             // two lanes per reset: a round serves A / 2 slots
    const uint32_t active = (uint32_t)__popcll(__ballot(1));
    const uint32_t A = SPLIT ? active >> 1 : active;
    if constexpr (SPLIT) {
        if (A == 0) {                                          // a last wave of ONE thread: it draws for itself
            Real o_unused[Env::O][VEC];
            reset_pending<Env, VEC, LANE_SEEDS>(pending, s, o_unused, a, i0, n, tick);
            return;
        }
    }
    for (uint32_t base = 0; base < total; base += A) {         // wave-uniform; more finished slots than lanes in a wave: ~never
#pragma unroll
        for (int j = 0; j < VEC; ++j)
            if (((pending >> j) & 1u) && rank[j] - base < A) sc->slot[rank[j] - base] = lane * VEC + (uint32_t)j;
        wave_lds_fence();
        if constexpr (SPLIT) {
            const uint32_t r = lane >> 1, call = lane & 1u;
            const bool draws = r < A && r < total - base;      // (lanes 2r and 2r + 1 are both active: 2r + 1 < 2A <= active)
            PhiloxWords w{};
            if (kResetFullExec || draws) {
                const uint32_t sl = sc->slot[draws ? r : 0u];
                const int64_t gl = wave_i0 + (int64_t)sl;
                uint64_t key = a.seed;
                if constexpr (LANE_SEEDS) {
                    if (a.lane_seed && gl < n) key = a.lane_seed[gl];
                }
                w = lane_words(Env::reset_call_key(key, call), a.lane_offset + (uint64_t)gl, tick);
            }
            const PhiloxWords other = neighbour_words(w);
            if ((kResetFullExec || draws) && call == 0) {
                Real sj[S];
                Env::reset_from_words(sj, w, other);
#pragma unroll
                for (int k = 0; k < S; ++k) sc->draw[r][k] = sj[k];
            }
        } else if (kResetFullExec || lane < total - base) {    // (a lane with a slot is active by construction: lane < A)
            // kResetFullExec: EVERY active lane draws — the lanes without a slot redo slot 0's draw and park it in their own row, which
            // nobody reads (see the constant's comment: an instruction under a mask of <= 16 lanes can cost five times a full one)
            const uint32_t sl = sc->slot[lane < total - base ? lane : 0u];
            const int64_t gl = wave_i0 + (int64_t)sl;
            Real sj[S];
            bool drawn = false;
            if constexpr (LANE_SEEDS) {
                if (a.lane_seed) {                             // kernel-uniform: per-lane keys are installed
                    draw_reset<Env, false>(sj, gl < n ? a.lane_seed[gl] : a.seed, a.lane_offset + (uint64_t)gl, tick);
                    drawn = true;
                }
            }
            if (!drawn) draw_reset<Env, true>(sj, a.seed, a.lane_offset + (uint64_t)gl, tick);
#pragma unroll
            for (int k = 0; k < S; ++k) sc->draw[lane][k] = sj[k];
        }
        wave_lds_fence();
        // every lane reads a row for each of its sub-lanes (clamped index; all reads in flight together, ONE wait) and keeps
        // it only where the sub-lane finished: a branch per sub-lane would serialise four LDS round trips
        Real got[VEC][S];
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

// wave64 compaction of the sub-lanes flagged in `finished`: rank of every flagged sub-lane inside the wave (ballot + mbcnt),
// returns how many the wave holds
template <int VEC>
__device__ __forceinline__ uint32_t rank_finished(const bool (&finished)[VEC], uint32_t (&off)[VEC]) {
    uint32_t total = 0;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const uint64_t m = __ballot(finished[j]);
        off[j] = total + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        total += (uint32_t)__popcll(m);
    }
    return total;
}

// The done list of ONE vector step (before the reset overwrites the terminal state): ballot per sub-lane, one atomic per wave into
// the wave's shard, order inside the list unspecified.  Everything known about a finished lane is written at ITS
// POSITION in the list — lane id, and with the corresponding flags its episode return / length and its terminal
// observation: a wave's ~11 finished lanes write one or two contiguous cache lines per array instead of one
// scattered line each (compacted (lane, return, length) records, BasePlaySession.cs:58-69).
template <class Env, int VEC>
__device__ __forceinline__ void append_done_records(const StepArgsT<typename Env::Real> &a, const bool (&finished)[VEC], int64_t i0,
                                                    const typename Env::Real (&s)[Env::S][VEC], const typename Env::Real (&o)[Env::O][VEC],
                                                    const float (&fin_ret)[VEC], const int32_t (&fin_len)[VEC], bool stats) {
    constexpr int S = Env::S, O = Env::O;
    const uint32_t lane = lane_id();
    uint32_t off[VEC];
    const uint32_t total = rank_finished<VEC>(finished, off);
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
    typename Env::Real s[Env::S][VEC];
    typename Env::Action act[VEC];
    int32_t sbd[VEC];
};

// The inputs of the batch's ragged end (the GUARD bodies of every kernel form): UNCONDITIONAL element loads, a lane past the end
// re-reads the last valid lane (only its stores are suppressed).  The bounds-checked load_row<GUARD> branches per access, and a branch around
// a load makes the compiler wait for it at the join: a thread's twenty loads then cost twenty memory latencies in a row (measured:
// 2^20 + 2 float64 lanes 19.6 us per step with branching loads in the last workgroups against 11.1 for the whole batch).
template <bool NT, class T>
__device__ __forceinline__ T load_elem(const T *p) { if constexpr (NT) return __builtin_nontemporal_load(p); else return *p; }

// one row's VEC elements, index clamped to the last valid lane
template <class T, int VEC, bool NT>
__device__ __forceinline__ void load_row_clamped(const T *__restrict__ p, int64_t i0, int64_t n, T (&v)[VEC]) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) v[j] = load_elem<NT>(p + (i0 + j < n ? i0 + j : n - 1));
}

template <class Env, int VEC, bool AUTORESET, int NT>
__device__ __forceinline__ void load_inputs_clamped(const StepArgsT<typename Env::Real> &a, const int64_t i0, LaneInputs<Env, VEC> &in) {
    constexpr bool NT_SL = (NT & 1) != 0, NT_A = (NT & 4) != 0;
#pragma unroll
    for (int k = 0; k < Env::S; ++k)
        load_row_clamped<typename Env::Real, VEC, NT_SL>(state_row_src<Env>(a.state, a.state_stride, a.obs_in, a.obs_stride, k), i0, a.n, in.s[k]);
    if constexpr (Env::BOX_ACTION) load_row_clamped<float, VEC, NT_A>(static_cast<const float *>(a.action), i0, a.n, in.act);
    else load_row_clamped<int32_t, VEC, NT_A>(static_cast<const int32_t *>(a.action), i0, a.n, in.act);
#pragma unroll
    for (int j = 0; j < VEC; ++j) in.sbd[j] = 0;
    if constexpr (!AUTORESET && Env::HAS_SBD) load_row_clamped<int32_t, VEC, NT_SL>(a.sbd, i0, a.n, in.sbd);
}

template <class Env, int VEC, bool AUTORESET, int NT, bool GUARD>
__device__ __forceinline__ void load_inputs(const StepArgsT<typename Env::Real> &a, const int64_t i0, LaneInputs<Env, VEC> &in) {
    constexpr bool NT_SL = (NT & 1) != 0, NT_A = (NT & 4) != 0;
    if constexpr (GUARD) { load_inputs_clamped<Env, VEC, AUTORESET, NT>(a, i0, in); return; }
    const int64_t n = a.n;
#pragma unroll
    for (int k = 0; k < Env::S; ++k)
        load_row<typename Env::Real, VEC, NT_SL, GUARD>(state_row_src<Env>(a.state, a.state_stride, a.obs_in, a.obs_stride, k), i0, n, in.s[k]);
    if constexpr (Env::BOX_ACTION) load_f32<VEC, NT_A, GUARD>(static_cast<const float *>(a.action), i0, n, in.act);
    else load_i32<VEC, NT_A, GUARD>(static_cast<const int32_t *>(a.action), i0, n, in.act);
#pragma unroll
    for (int j = 0; j < VEC; ++j) in.sbd[j] = 0;
    if constexpr (!AUTORESET && Env::HAS_SBD) load_i32<VEC, NT_SL, GUARD>(a.sbd, i0, n, in.sbd);
}

template <class Env, int VEC, bool AUTORESET, bool EXTRAS, int NT, bool GUARD, int RESETF = 0, bool PACK = true>
__device__ __forceinline__ void advance_and_store(const StepArgsT<typename Env::Real> &a, const int64_t i0, const uint64_t tick,
                                                  LaneInputs<Env, VEC> &in, ResetScratch<Env> *sc = nullptr) {
    constexpr int S = Env::S, O = Env::O;
    using Real = typename Env::Real;
    constexpr bool NT_SS = (NT & 2) != 0, NT_O = (NT & 8) != 0;
    const int64_t n = a.n;
    Real (&s)[S][VEC] = in.s;
    typename Env::Action (&act)[VEC] = in.act;
    int32_t (&sbd)[VEC] = in.sbd;

    constexpr bool NT_SL = (NT & 1) != 0;
    float ep_ret[VEC], fin_ret[VEC];
    int32_t ep_len[VEC], fin_len[VEC];
    bool stats = false;
    if constexpr (EXTRAS) {
        stats = a.ep_ret != nullptr;
        // running return / length: read-modify-write streams like the state, same non-temporal policy
        if (stats) {
            if constexpr (GUARD) { load_row_clamped<float, VEC, NT_SL>(a.ep_ret, i0, n, ep_ret); load_row_clamped<int32_t, VEC, NT_SL>(a.ep_len, i0, n, ep_len); }
            else { load_f32<VEC, NT_SL, false>(a.ep_ret, i0, n, ep_ret); load_i32<VEC, NT_SL, false>(a.ep_len, i0, n, ep_len); }
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) { fin_ret[j] = 0.0f; fin_len[j] = 0; }
    }

    float reward[VEC];
    uint8_t done[VEC];
    bool finished[VEC];
    bool after[VEC];          // sub-lane was stepped although it had already returned done (no auto-reset only)
#pragma unroll
    for (int j = 0; j < VEC; ++j) after[j] = false;
    Real o[O][VEC];
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
        if (a.done_list) append_done_records<Env, VEC>(a, finished, i0, s, o, fin_ret, fin_len, stats);
    }

    if constexpr (AUTORESET && RESETF == 1) reset_pending_wave<Env, VEC, EXTRAS>(pending, s, a, i0, n, tick, sc);
    else if constexpr (AUTORESET) reset_pending<Env, VEC, EXTRAS>(pending, s, o, a, i0, n, tick);

#pragma unroll
    for (int k = 0; k < S; ++k)
        if (state_row_own<Env>(k)) store_row<Real, VEC, NT_SS, GUARD>(a.state_out + k * a.state_stride, i0, n, s[k]);
    if constexpr (!Env::OBS_ALIASES_STATE) {
#pragma unroll
        for (int k = 0; k < O; ++k) store_row<Real, VEC, NT_SS, GUARD>(a.obs + k * a.obs_stride, i0, n, o[k]);
    }
    if constexpr (!AUTORESET && Env::HAS_SBD) store_i32<VEC, NT_SS, GUARD>(a.sbd, i0, n, sbd);

    if constexpr (EXTRAS) {
        if (stats) { store_f32<VEC, NT_SS, GUARD>(a.ep_ret, i0, n, ep_ret); store_i32<VEC, NT_SS, GUARD>(a.ep_len, i0, n, ep_len); }
    }
}

template <class Env, int VEC, bool AUTORESET, bool EXTRAS, int NT, bool GUARD, int RESETF = 0>
__device__ __forceinline__ void step_body(const StepArgsT<typename Env::Real> &a, const int64_t i0, const uint64_t tick, ResetScratch<Env> *sc = nullptr) {
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
struct LaneOutputs { typename Env::Real s[Env::S], o[Env::O]; float reward; uint8_t done; int32_t sbd; };

template <class Env, bool AUTORESET>
__device__ __forceinline__ void compute_lane(const StepArgsT<typename Env::Real> &a, int64_t i, uint64_t tick, LaneInputs<Env, 1> &in,
                                             LaneOutputs<Env> &out) {
    constexpr int S = Env::S, O = Env::O;
    typename Env::Real o[O][1];
    float rw[1];
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
__device__ __forceinline__ void store_lane(const StepArgsT<typename Env::Real> &a, int64_t i, const LaneOutputs<Env> &out) {
    constexpr bool NT_SS = (NT & 2) != 0, NT_O = (NT & 8) != 0;
    auto st = [](auto *p, auto v, bool nt) { if (nt) __builtin_nontemporal_store(v, p); else *p = v; };
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
__global__ __launch_bounds__(256) void step_kernel_pipe(const StepArgsT<typename Env::Real> a) {
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
// Round 5, measured and removed again (profiles/f64_forms_r05.txt): lane QUADS per item for the float64 kernel (14.8-16.7 us against
// 13.0), the wave-compacted reset per item (no better than the drain loop), and a reset deferred to ONE compacted pass per wave for
// all of a thread's pairs with the drawing lanes storing the fresh states (correct, and 8 us slower: a state line written in two
// pieces at two times costs far more than the Philox passes it saves).
//
// Round 5, second pass — the fused reset of this kernel, drawn ONCE per thread-group of pairs (envs whose reset is two Philox calls:
// float64 CartPole).  tools/skeleton_floor.hip prices the kernel's parts on one box: data movement alone 11.7 us, with the physics
// and a constant reset 12.1, with the real reset 14.0 — the time above the skeleton was the Philox passes, not the binary64
// arithmetic.  The per-thread drain loop costs a wave ~1.9 trips per pair, two calls per trip, ~3 of 64 lanes active: 15 call-passes
// per wave and launch, each 20 quarter-rate v_mad_u64_u32.  reset_group_deferred() ranks the finished sub-lanes of ALL the thread's
// pairs (ballot + mbcnt), hands them to the wave's lanes through LDS TWO LANES PER RESET (lane 2r draws call 0, lane 2r + 1 call 1:
// one call-pass serves 32 resets, and a wave of 512 lanes holds ~23), joins the two halves with one DPP exchange, converts on the
// even lane and hands the states back BEFORE the pairs' state rows are stored — whole rows, written once (the earlier deferred form
// that let the drawing lanes store the fresh states wrote rows in two pieces and lost 8 us).  Same counters, words and conversions:
// bit-identical (tools/deferred_reset_probe.hip, tests).  14.5 -> 12.5 us on the probe's box (profiles/f64_deferred_reset_r05.txt).
template <class Env>
struct DeferScratch {
    uint32_t slot[32];                        // rank -> owner lane * 8 + (pair * 2 + sub-lane)
    typename Env::Real draw[32][Env::S];      // rank -> the drawn state
};

// pending: bit (pair * 2 + sub-lane) of this thread's 2 * PAIRS sub-lanes; s(pair) returns the pair's state registers
template <class Env, int PAIRS, class StateOf>
__device__ __forceinline__ void reset_group_deferred(uint32_t pending, StateOf s, const StepArgsT<typename Env::Real> &a, int64_t t_wave0,
                                                     int64_t T, int first_pair, uint64_t tick, DeferScratch<Env> *sc) {
    static_assert(Env::RESET_CALLS == 2 && 2 * PAIRS <= 8, "two lanes per reset; the slot word keeps 3 bits for the sub-lane position");
    constexpr int Q = 2 * PAIRS, S = Env::S;
    using Real = typename Env::Real;
    const uint32_t lane = lane_id();
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
        const uint32_t r = lane >> 1, call = lane & 1u;
        const bool draws = r < total - base;
        PhiloxWords w{};
        if (kResetFullExec || draws) {
            const uint32_t sl = sc->slot[draws ? r : 0u];
            const uint32_t owner = sl >> 3, q = sl & 7u;
            const int64_t gl = ((t_wave0 + owner) + (int64_t)(first_pair + (int)(q >> 1)) * T) * 2 + (q & 1u);
            w = lane_words(Env::reset_call_key(a.seed, call), a.lane_offset + (uint64_t)gl, tick);
        }
        // the even lane takes its odd neighbour's words (call 1) and converts
        const PhiloxWords other = neighbour_words(w);
        if ((kResetFullExec || draws) && call == 0) {
            Real sj[S];
            Env::reset_from_words(sj, w, other);
#pragma unroll
            for (int k = 0; k < S; ++k) sc->draw[r][k] = sj[k];
        }
        wave_lds_fence();
        // every lane reads a row for each of its sub-lane positions (clamped index: all reads in flight together, one wait)
        Real got[Q][S];
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const uint32_t rr = rank[q] - base;
#pragma unroll
            for (int k = 0; k < S; ++k) got[q][k] = sc->draw[rr < 32u ? rr : 0u][k];
        }
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const bool mine = ((pending >> q) & 1u) && rank[q] - base < 32u;
            Real (&sq)[S][2] = s(q / 2);
#pragma unroll
            for (int k = 0; k < S; ++k) sq[k][q % 2] = mine ? got[q][k] : sq[k][q % 2];
        }
        wave_lds_fence();
    }
}

// the thread's ITEMS pairs with ONE deferred reset: all loads | advance pair after pair (reward / done leave at once) | one
// wave-compacted draw for all of them | the state rows.  GUARD: the batch's ragged end (a lane past the end re-reads the last valid
// lane, is advanced like any other, never pends a reset and stores nothing) — only the batch's last workgroups run it.
template <class Env, int ITEMS, int NT, bool GUARD>
__device__ __forceinline__ void pipe2_split_body(const StepArgsT<typename Env::Real> &a, int64_t t, int64_t T, uint64_t tick, DeferScratch<Env> *sc) {
    using Real = typename Env::Real;
    constexpr bool NT_SS = (NT & 2) != 0, NT_O = (NT & 8) != 0;
    LaneInputs<Env, 2> in[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        load_inputs<Env, 2, true, NT, GUARD>(a, (t + k * T) * 2, in[k]);
    }
    uint32_t pending = 0;
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        if (k == 0) {   // pair 0's inputs are needed now (their first uses must not be hoisted into the load block)
#pragma unroll
            for (int c = 0; c < Env::S; ++c) asm volatile("" : "+v"(in[0].s[c][0]), "+v"(in[0].s[c][1]));
        }
        const int64_t i0 = (t + k * T) * 2;
        Real o[Env::O][2];
        float rw[2];
        bool dn[2], after[2] = {false, false};
        advance_all<Env, 2, true, GUARD, false>(in[k].s, in[k].act, in[k].sbd, rw, dn, after, o, i0, a.n);
        const uint8_t db[2] = {(uint8_t)(dn[0] ? 1 : 0), (uint8_t)(dn[1] ? 1 : 0)};
        store_f32<2, NT_O, GUARD>(a.reward, i0, a.n, rw);           // reward / done do not wait for the reset draw
        store_u8<2, NT_O, GUARD>(a.done, i0, a.n, db);
        const bool p0 = dn[0] && (!GUARD || i0 < a.n), p1 = dn[1] && (!GUARD || i0 + 1 < a.n);
        pending |= ((p0 ? 1u : 0u) | (p1 ? 2u : 0u)) << (2 * k);
    }
    reset_group_deferred<Env, ITEMS>(pending, [&](int pair) -> Real (&)[Env::S][2] { return in[pair].s; }, a, t - (int64_t)lane_id(), T, 0, tick, sc);
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
#pragma unroll
        for (int row = 0; row < Env::S; ++row)
            store_row<Real, 2, NT_SS, GUARD>(a.state_out + row * a.state_stride, (t + k * T) * 2, a.n, in[k].s[row]);
    }
}

// Round 6 — batches beyond ONE resident generation.  The four-pair float64 kernel holds 185 VGPRs: two waves per SIMD, 2048 waves of
// 512 lanes = 2^20 lanes resident at once, and there it runs load burst | arithmetic | store burst as one generation (0.84 of 8 TB/s).
// Launched over 2^21 lanes it ran as hardware-scheduled generations, every new workgroup waiting for four wave slots (one per SIMD of
// a CU) to drain: 29.6 us, worse than the one-shot kernel's 27.7 (profiles/f64_sizes_r05.txt).  Two remedies were measured
// (profiles/f64_sizes_r06.txt):
//   * a grid of one resident generation whose threads LOOP over the generations: slower still (35.2 us at 2^21 lanes) — a wave's
//     load, arithmetic and store phases are serial and two waves per SIMD cannot cover each other's memory latency; the one-generation
//     launch is fast because the whole CHIP moves through the phases together, not because of the grid's shape.  Removed.
//   * one launch PER generation over consecutive slices of the batch (launch_step_env below: pipe2_chunks) — each slice is exactly the
//     2^20-lane launch, and consecutive launches on a stream overlap their ramp with the predecessor's drain the way consecutive steps
//     of a rollout do.  Kept where it measures faster than the one-shot kernel.
template <class Env, int ITEMS, bool AUTORESET, int NT>
// (No occupancy hint: the float64 four-pair kernel holds 185 VGPRs = two waves per SIMD; capped at 168 for three it spills 48 bytes and
// runs at 14.0 instead of 11.1 us per 2^20-lane step — profiles/occupancy_hints_r05.txt.)
__global__ __launch_bounds__(256) void step_kernel_pipe2(const StepArgsT<typename Env::Real> a) {
    const uint64_t tick = a.tick2[a.parity];
    if (blockIdx.x == 0 && threadIdx.x == 0) a.tick2[a.parity ^ 1] = tick + 1;
    const int64_t T = (int64_t)gridDim.x * blockDim.x;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if constexpr (AUTORESET && has_split_reset<Env>()) {
        __shared__ DeferScratch<Env> scratch[256 / 64];            // one table per wave of the workgroup
        DeferScratch<Env> *sc = &scratch[threadIdx.x >> 6];
        // ANY batch size: the grid is ceil(n / (2 * ITEMS * block)) workgroups; a workgroup whose last pair of its last item lies
        // inside the batch runs the unguarded body (workgroup-uniform), the batch's last few run the guarded one
        if ((((int64_t)blockIdx.x + 1) * blockDim.x + (int64_t)(ITEMS - 1) * T) * 2 <= a.n) pipe2_split_body<Env, ITEMS, NT, false>(a, t, T, tick, sc);
        else pipe2_split_body<Env, ITEMS, NT, true>(a, t, T, tick, sc);
        return;
    }
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

// The slices a multi-pair step is launched in: `chunks` consecutive slices of `lanes` lanes (the last one shorter), each ONE resident
// generation of the kernel.  A workgroup is four waves, one per SIMD of a CU, so the chip holds (SIMDs / 4) x (waves per SIMD the
// kernel's registers allow) workgroups at once: two for the four-pair float64 kernel (185 VGPRs), three for fewer pairs
// (tests/test_kernel_resources.py asserts both).  Only the form with the deferred reset is sliced (any batch size, lean, auto-reset).
struct Pipe2Chunks { int64_t lanes; int chunks; };
static inline Pipe2Chunks pipe2_chunks(int64_t n, int items, int block, bool sliced, int simds) {
    const int64_t per_block = 2 * (int64_t)items * block;
    const int64_t resident = (int64_t)(simds > 0 ? simds : 1024) / 4 * (items >= 4 ? 2 : 3) * (256 / block);
    const int64_t cap = resident * per_block;
    if (!sliced || n <= cap) return Pipe2Chunks{n, 1};
    return Pipe2Chunks{cap, (int)((n + cap - 1) / cap)};
}

// a StepArgs for the slice of `count` lanes that starts at lane `first` (lean variant: the bookkeeping arrays are not in use)
template <class R>
static StepArgsT<R> slice_of(const StepArgsT<R> &a, int64_t first, int64_t count) {
    StepArgsT<R> s = a;
    s.state += first; s.state_out += first;
    if (s.obs_in) s.obs_in += first;
    if (s.obs) s.obs += first;
    s.action = static_cast<const char *>(a.action) + (size_t)first * 4;
    s.reward += first; s.done += first;
    if (s.sbd) s.sbd += first;
    if (s.lane_seed) s.lane_seed += first;
    s.lane_offset += (uint64_t)first;
    s.n = count;
    return s;
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
    static_assert(std::is_same<typename Env::Real, float>::value, "the producer / consumer form stages float rows through LDS");
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
__global__ __launch_bounds__(256) void step_kernel(const StepArgsT<typename Env::Real> a) {
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
// ActionSpace.Sample() of one lane for one step, drawn in the kernel from the lane's word A of the action stream (philox.hpp, action
// stream v2): the value sample_discrete_kernel / sample_box_kernel (kernels.hip) write for the same (seed, global lane, tick); word B
// is the epsilon-greedy coin (compose_discrete_kernel).
template <class Env>
__device__ __forceinline__ typename Env::Action sampled_action(uint32_t word_a) {
    if constexpr (Env::BOX_ACTION) return Env::ACTION_LOW + (Env::ACTION_HIGH - Env::ACTION_LOW) * u01_24(word_a);     // Box.cs:85
    else return (int32_t)__umulhi(word_a, (uint32_t)Env::ACTION_N);                                                    // Discrete.cs:27
}

// Words A (and, if asked, B) of a thread's VEC consecutive lanes gl0 .. gl0 + VEC - 1.  `one_group` (kernel-uniform: the handle's
// lane offset is a multiple of VEC, so gl0 is) = the lanes lie in ONE group of four and share its Philox call: a dwordx4 thread pays
// one call per step for its four actions where v1 paid four.  Otherwise (a shard that starts inside a group) every lane makes its
// group's call and keeps its own word — the same words, four times the arithmetic.
template <int VEC>
__device__ __forceinline__ void thread_action_words(uint64_t seed, uint64_t gl0, uint64_t tick, bool want_aux, bool one_group,
                                                    uint32_t (&wa)[VEC], uint32_t (&wb)[VEC]) {
    static_assert(VEC == 1 || VEC == 2 || VEC == 4, "a thread's lanes must fit one group of four");
    if (one_group) {
        const PhiloxWords r = action_group_words<true>(seed, gl0 >> 2, tick);       // (the action seed is a kernel argument: uniform)
        PhiloxWords c{};
        if (want_aux) c = aux_group_words<true>(seed, gl0 >> 2, tick);
        const uint32_t base = (uint32_t)gl0 & 3u;
        if constexpr (VEC == 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { wa[j] = r.w[j]; wb[j] = c.w[j]; }
        } else if constexpr (VEC == 2) {
#pragma unroll
            for (int j = 0; j < 2; ++j) { wa[j] = base ? r.w[2 + j] : r.w[j]; wb[j] = base ? c.w[2 + j] : c.w[j]; }
        } else {
            wa[0] = word_of(r, base); wb[0] = word_of(c, base);
        }
    } else {
        // one copy of the call in the instruction stream (a rolled loop; the results go to their registers through selects on the
        // loop index, not through a dynamic index): the fast path above is the one that runs, this one only has to stay small
#pragma unroll
        for (int j = 0; j < VEC; ++j) { wa[j] = 0u; wb[j] = 0u; }
#pragma unroll 1
        for (int j = 0; j < VEC; ++j) {
            const uint32_t a = action_word<true>(seed, gl0 + (uint64_t)j, tick);
            const uint32_t b = want_aux ? aux_word<true>(seed, gl0 + (uint64_t)j, tick) : 0u;
#pragma unroll
            for (int k = 0; k < VEC; ++k) { wa[k] = k == j ? a : wa[k]; wb[k] = k == j ? b : wb[k]; }
        }
    }
}

//   EXTRAS  bookkeeping handle (EPISODE_STATS / DONE_LIST / FINAL_OBS / per-lane seeds): running return / length in registers,
//           truncation, dense last-finished-episode views, per-rollout compact episode records, the done list of the LAST step
//   SAMPLE  the actions are drawn in the kernel (RolloutArgs::action_source 1 or 2) instead of read from the ring
//   RESETF  1 = the wave-compacted reset (reset_pending_wave) per step instead of the per-thread drain loop
// Episode records of a rollout are STAGED per wave in LDS (kStageRecords: ~22 steps' worth at CartPole's rate) and flushed with one atomic
// and full-line stores; appending each step's ~11 records directly cost one atomic round trip per wave and step on the critical path.
// (Round 6: 256 records per wave instead of 64 — a wave flushes every ~22 steps instead of every ~5; each flush is an atomic round trip
// the wave waits for.  4 KiB of LDS per wave, 16 KiB per workgroup.)
constexpr uint32_t kStageRecords = 256;
struct EpisodeStage { int32_t t[kStageRecords], lane[kStageRecords]; float ret[kStageRecords]; int32_t len[kStageRecords]; };

//   RECORDS 1 / 2: the rollout keeps compact episode records (RolloutArgs::ep_*; bookkeeping handles) — 1 (the default): what a shard's segment
//           cannot hold spills to the shared overflow segment, nothing is lost below the caller's capacity; 2 (GYMNET_RECORDS_NO_OVERFLOW):
//           no spill path — its mere presence costs the kernel 8 % (a call, or a cold block, inside a loop whose registers are capped
//           at 128), and an evenly finishing batch never takes it.  A variant of its own, so that the
//           rollouts that keep none do not carry the staging code's registers (128-VGPR budget, above)
template <class Env, int VEC, bool AUTORESET, bool GUARD, bool EXTRAS, bool SAMPLE, int RESETF = 0, int RECORDS = 0>
__device__ __forceinline__ void rollout_body(const StepArgsT<typename Env::Real> &a, const RolloutArgsT<typename Env::Real> &ro,
                                             const int64_t i0, const uint64_t tick0, ResetScratch<Env> *sc = nullptr, EpisodeStage *stage = nullptr) {
    constexpr int S = Env::S, O = Env::O;
    using Act = typename Env::Action;
    using Real = typename Env::Real;
    const int64_t n = a.n;

    Real s[S][VEC];
#pragma unroll
    for (int k = 0; k < S; ++k) load_row<Real, VEC, true, GUARD>(state_row_src<Env>(a.state, a.state_stride, a.obs_in, a.obs_stride, k), i0, n, s[k]);
    int32_t sbd[VEC] = {};
    if constexpr (!AUTORESET && Env::HAS_SBD) load_i32<VEC, true, GUARD>(a.sbd, i0, n, sbd);

    auto load_action = [&](int64_t slice, Act (&dst)[VEC]) {
        const char *base = static_cast<const char *>(a.action) + (size_t)(slice * ro.action_stride) * 4;
        if constexpr (Env::BOX_ACTION) load_f32<VEC, true, GUARD>(reinterpret_cast<const float *>(base), i0, n, dst);
        else load_i32<VEC, true, GUARD>(reinterpret_cast<const int32_t *>(base), i0, n, dst);
    };

    Act act[VEC], act_next[VEC];
    // the ring is read unless every action is sampled; epsilon-greedy (action_source 2) reads it as the POLICY's actions
    bool use_ring = true;                                          // kernel-uniform
    const bool one_group = (a.lane_offset % (uint64_t)VEC) == 0;   // kernel-uniform: the thread's lanes share one action-stream call
    const uint32_t explore_at_or_below = coin_threshold(ro.epsilon);   // kernel-uniform (philox.hpp): the coin is one integer compare
    if constexpr (SAMPLE) {
        use_ring = ro.action_source == 2;
#pragma unroll
        for (int j = 0; j < VEC; ++j) { act[j] = Act(0); act_next[j] = Act(0); }
        if (use_ring) load_action(0, act);
    } else {
        load_action(0, act);
    }
    int64_t slice = 0;
    float reward[VEC];
    uint8_t done[VEC];
    Real o[O][VEC];

    // bookkeeping state (EXTRAS): the running episode return / length of the thread's lanes stay in registers for the whole rollout
    float ep_ret[VEC], fin_ret[VEC];
    int32_t ep_len[VEC], fin_len[VEC];
    bool stats = false;
    if constexpr (EXTRAS) {
        stats = a.ep_ret != nullptr;
        if (stats) { load_f32<VEC, true, GUARD>(a.ep_ret, i0, n, ep_ret); load_i32<VEC, true, GUARD>(a.ep_len, i0, n, ep_len); }
    }

    uint32_t staged = 0;               // records waiting in the wave's LDS stage (wave-uniform)
    auto flush_stage = [&]() {
        if constexpr (EXTRAS && RECORDS) {
            if (staged == 0) return;                                   // wave-uniform
            wave_lds_fence();
            const uint64_t act_mask = __ballot(1);
            const int leader = __ffsll((unsigned long long)act_mask) - 1;
            const uint32_t A = (uint32_t)__popcll(act_mask);           // the active lanes are a prefix (a partial last wave)
            const uint32_t shard = wave_shard();
            uint32_t base = 0;
            if ((int)lane_id() == leader) base = atomicAdd(&ro.ep_count[shard * kCountStride], staged);
            base = __shfl(base, leader);
            for (uint32_t b0 = 0; b0 < staged; b0 += A) {
                const uint32_t q = b0 + lane_id();
                if (q < staged && (int64_t)(base + q) < ro.ep_cap) {
                    const int64_t pos = (int64_t)shard * ro.ep_cap + base + q;
                    ro.ep_t[pos] = stage->t[q];
                    ro.ep_lane[pos] = stage->lane[q];
                    if (ro.ep_ret) { ro.ep_ret[pos] = stage->ret[q]; ro.ep_len[pos] = stage->len[q]; }
                }
            }
            // what the shard's segment cannot hold goes to the shared overflow segment: one more atomic, only for a shard that is full
            if constexpr (RECORDS == 1) {
                // "segment kShards" of the same arrays, counter ep_count[kShards * kCountStride]; a wave-uniform branch nobody takes in an evenly
                // finishing batch (measured as this block and as a noinline call: the same 8 %, and the call adds a stack frame)
                if (__builtin_expect((int64_t)base + (int64_t)staged > ro.ep_cap, 0)) {
                    const uint32_t first_ov = (int64_t)base < ro.ep_cap ? (uint32_t)(ro.ep_cap - (int64_t)base) : 0u;
                    uint32_t ovbase = 0;
                    if ((int)lane_id() == leader) ovbase = atomicAdd(&ro.ep_count[kShards * kCountStride], staged - first_ov);
                    ovbase = __shfl(ovbase, leader);
                    for (uint32_t b0 = 0; b0 < staged; b0 += A) {
                        const uint32_t q = b0 + lane_id();
                        if (q >= first_ov && q < staged && (int64_t)(ovbase + (q - first_ov)) < ro.ov_cap) {   // beyond ep_capacity in total: counted, not kept
                            const int64_t pos = (int64_t)kShards * ro.ep_cap + ovbase + (q - first_ov);
                            ro.ep_t[pos] = stage->t[q];
                            ro.ep_lane[pos] = stage->lane[q];
                            if (ro.ep_ret) { ro.ep_ret[pos] = stage->ret[q]; ro.ep_len[pos] = stage->len[q]; }
                        }
                    }
                }
            }
            wave_lds_fence();
            staged = 0;
        }
    };

    for (int64_t t = 0; t < ro.steps; ++t) {
        int64_t nslice = slice + 1;
        if (nslice == ro.ring) nslice = 0;
        if constexpr (SAMPLE) {
            if (use_ring && t + 1 < ro.steps) load_action(nslice, act_next);
            uint32_t wa[VEC], wb[VEC];
            thread_action_words<VEC>(ro.action_seed, a.lane_offset + (uint64_t)i0, ro.action_tick0 + (uint64_t)t, use_ring, one_group, wa, wb);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const Act drawn = sampled_action<Env>(wa[j]);
                if constexpr (Env::BOX_ACTION) act[j] = drawn;                   // (epsilon-greedy is defined for Discrete spaces)
                else act[j] = (use_ring && !(wb[j] <= explore_at_or_below)) ? act[j] : drawn;   // u01_24(word B) <= epsilon: TrainingPlaySession.cs:46-52
            }
        } else {
            if (t + 1 < ro.steps) load_action(nslice, act_next);        // in flight during this step's math
        }
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
                    Real sj[S], oj[O];
                    float rw;
                    bool dn;
#pragma unroll
                    for (int k = 0; k < S; ++k) sj[k] = s[k][j];
                    advance_sublane<Env, AUTORESET, decltype(small_tag)::value>(sj, act[j], sbd[j], rw, dn, after[j], !GUARD || i0 + j < n, oj);
                    done[j] = dn ? 1 : 0;
                    reward[j] = rw;
                    if constexpr (AUTORESET && !EXTRAS) pending |= dn ? (1u << j) : 0u;
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
                if constexpr (AUTORESET && !EXTRAS) pending |= dnv[j] ? (1u << j) : 0u;
            }
        }
        if constexpr (EXTRAS) {
            // the step kernel's bookkeeping (advance_and_store), per step of the rollout: truncation, the dense "last finished
            // episode per lane" views, and the compact records of this step's finished lanes
            bool finished[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const bool in_range = !GUARD || i0 + j < n;
                fin_ret[j] = 0.0f; fin_len[j] = 0;
                if (stats) {
                    ep_ret[j] += reward[j];
                    ep_len[j] += 1;
                    if (a.max_episode_steps > 0 && ep_len[j] >= a.max_episode_steps) done[j] |= 2;   // truncated (extension)
                }
                const bool fin = done[j] != 0;
                finished[j] = fin && in_range;
                if (fin && a.final_obs && in_range) {
#pragma unroll
                    for (int k = 0; k < O; ++k) a.final_obs[k * n + i0 + j] = Env::OBS_ALIASES_STATE ? s[k < S ? k : 0][j] : o[k][j];
                }
                if (stats && fin && in_range) {
                    fin_ret[j] = ep_ret[j];
                    fin_len[j] = ep_len[j];
                    if (a.fin_ret) { a.fin_ret[i0 + j] = ep_ret[j]; a.fin_len[i0 + j] = ep_len[j]; }
                    if constexpr (AUTORESET) { ep_ret[j] = 0.0f; ep_len[j] = 0; }
                }
                if constexpr (AUTORESET) pending |= fin ? (1u << j) : 0u;
            }
            if constexpr (RECORDS) {
                // (t, lane, return, length) of every episode that ended in this step, compacted per wave (ballot + mbcnt) into the wave's
                // LDS stage; a full stage is flushed with ONE atomic into the wave's shard (4096 waves appending to one counter would
                // serialise: StepArgs::done_list) and contiguous stores.  A step with more finished lanes than the stage holds (every
                // lane hitting the time limit in the same step) flushes and appends directly.
                uint32_t off[VEC];
                const uint32_t total = rank_finished<VEC>(finished, off);
                if (total && staged + total > kStageRecords) flush_stage();   // wave-uniform
                if (total > kStageRecords) {
                    const int leader = __ffsll((unsigned long long)__ballot(1)) - 1;
                    const uint32_t shard = wave_shard();
                    uint32_t base = 0;
                    if ((int)lane_id() == leader) base = atomicAdd(&ro.ep_count[shard * kCountStride], total);
                    base = __shfl(base, leader);
#pragma unroll
                    for (int j = 0; j < VEC; ++j) {
                        if (!finished[j] || (int64_t)(base + off[j]) >= ro.ep_cap) continue;       // beyond the capacity: counted, not kept
                        const int64_t pos = (int64_t)shard * ro.ep_cap + base + off[j];
                        ro.ep_t[pos] = (int32_t)t;
                        ro.ep_lane[pos] = (int32_t)(i0 + j);
                        if (ro.ep_ret) { ro.ep_ret[pos] = fin_ret[j]; ro.ep_len[pos] = fin_len[j]; }
                    }
                } else if (total) {
#pragma unroll
                    for (int j = 0; j < VEC; ++j) {
                        if (!finished[j]) continue;
                        const uint32_t q = staged + off[j];
                        stage->t[q] = (int32_t)t; stage->lane[q] = (int32_t)(i0 + j); stage->ret[q] = fin_ret[j]; stage->len[q] = fin_len[j];
                    }
                    staged += total;
                }
            }
            // the done list "of the most recent step" (gymnet_vecenv_done_lanes / _done_records) describes the rollout's LAST step
            if (a.done_list && t + 1 == ro.steps) append_done_records<Env, VEC>(a, finished, i0, s, o, fin_ret, fin_len, stats);
        }
        if constexpr (!AUTORESET && Env::HAS_SBD) count_after_done<VEC>(a, after);
        if (ro.rec_reward) store_f32<VEC, true, GUARD>(ro.rec_reward + t * n, i0, n, reward);
        if (ro.rec_done) store_u8<VEC, true, GUARD>(ro.rec_done + t * n, i0, n, done);
        if constexpr (EXTRAS || SAMPLE) {
            if (ro.rec_action) {
                if constexpr (Env::BOX_ACTION) store_f32<VEC, true, GUARD>(static_cast<float *>(ro.rec_action) + t * n, i0, n, act);
                else store_i32<VEC, true, GUARD>(static_cast<int32_t *>(ro.rec_action) + t * n, i0, n, act);
            }
        }
        if constexpr (AUTORESET && RESETF == 1) reset_pending_wave<Env, VEC, EXTRAS>(pending, s, a, i0, n, tick0 + (uint64_t)t, sc);
        else if constexpr (AUTORESET) reset_pending<Env, VEC, EXTRAS>(pending, s, o, a, i0, n, tick0 + (uint64_t)t);
        if (ro.rec_obs) {
#pragma unroll
            for (int k = 0; k < O; ++k) {
                if constexpr (Env::OBS_ALIASES_STATE) store_row<Real, VEC, true, GUARD>(ro.rec_obs + (t * O + k) * n, i0, n, s[k]);
                else store_row<Real, VEC, true, GUARD>(ro.rec_obs + (t * O + k) * n, i0, n, o[k]);
            }
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) act[j] = act_next[j];
        slice = nslice;
    }
    flush_stage();

#pragma unroll
    for (int k = 0; k < S; ++k)
        if (state_row_own<Env>(k)) store_row<Real, VEC, false, GUARD>(a.state_out + k * a.state_stride, i0, n, s[k]);
    if constexpr (!Env::OBS_ALIASES_STATE) {
#pragma unroll
        for (int k = 0; k < O; ++k) store_row<Real, VEC, false, GUARD>(a.obs + k * a.obs_stride, i0, n, o[k]);
    }
    store_f32<VEC, false, GUARD>(a.reward, i0, n, reward);
    store_u8<VEC, false, GUARD>(a.done, i0, n, done);
    if constexpr (!AUTORESET && Env::HAS_SBD) store_i32<VEC, false, GUARD>(a.sbd, i0, n, sbd);
    if constexpr (EXTRAS) {
        if (stats) { store_f32<VEC, false, GUARD>(a.ep_ret, i0, n, ep_ret); store_i32<VEC, false, GUARD>(a.ep_len, i0, n, ep_len); }
    }
}

// Occupancy: a float32 rollout at 2^20 lanes is 4096 waves of four lanes per thread — FOUR per SIMD.  The bookkeeping variants
// allocate 131-152 VGPRs by themselves (three waves per SIMD: a quarter of the launch runs as a second generation); capped at 128
// (four blocks of four waves per CU) they fit in one — 4.1 -> 3.2 us per vector step with episode statistics, 7.3 -> 6.3 with sampled
// actions and episode records, at the price of 36-92 bytes of scratch (profiles/occupancy_hints_r05.txt).  The float64 variants lose under
// the same cap (bookkeeping rollout with the compacted reset 6.9 -> 7.7 us) and are left alone.
template <class Env, int VEC, bool AUTORESET, bool EXTRAS = false, bool SAMPLE = false, int RESETF = 0, int RECORDS = 0>
__global__ __launch_bounds__(256, sizeof(typename Env::Real) == 4 ? 4 : 1) void rollout_kernel(const StepArgsT<typename Env::Real> a, const RolloutArgsT<typename Env::Real> ro) {
    const int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * VEC;
    ResetScratch<Env> *sc = nullptr;
    if constexpr (RESETF == 1) {
        __shared__ ResetScratch<Env> scratch[256 / 64];            // one table per wave of the workgroup
        sc = &scratch[threadIdx.x >> 6];
    }
    EpisodeStage *stage = nullptr;
    if constexpr (EXTRAS && RECORDS) {
        __shared__ EpisodeStage stages[256 / 64];                  // one per wave of the workgroup
        stage = &stages[threadIdx.x >> 6];
    }
    const uint64_t tick0 = a.tick2[a.parity];
    if (blockIdx.x == 0 && threadIdx.x == 0) a.tick2[a.parity ^ 1] = tick0 + (uint64_t)ro.steps;
    if constexpr (EXTRAS) {
        if (blockIdx.x == 0 && a.done_count2)     // zero the NEXT step launch's half of the shard counters (as step_kernel does)
            for (int sh = threadIdx.x; sh < kShards; sh += blockDim.x) a.done_count2[(a.cparity ^ 1) * (kShards * kCountStride) + sh * kCountStride] = 0u;
    }
    if (((int64_t)blockIdx.x + 1) * blockDim.x * VEC <= a.n) {     // full workgroup: no bounds checks inside the T-step loop
        rollout_body<Env, VEC, AUTORESET, false, EXTRAS, SAMPLE, RESETF, RECORDS>(a, ro, i0, tick0, sc, stage);
    } else {
        if (i0 >= a.n) return;
        rollout_body<Env, VEC, AUTORESET, true, EXTRAS, SAMPLE, RESETF, RECORDS>(a, ro, i0, tick0, sc, stage);
    }
}

// ---------------------------------------------------------------------------------------------
// Reset: all lanes, or the lanes selected by a byte mask (the caller's `if (done) Reset()`).
// ---------------------------------------------------------------------------------------------
template <class Env>
__device__ __forceinline__ void reset_lane(const ResetArgsT<typename Env::Real> &a, int64_t i, uint64_t tick) {
    constexpr int S = Env::S, O = Env::O;
    const uint64_t key = a.lane_seed ? a.lane_seed[i] : a.seed;
    typename Env::Real s[S];
    draw_reset<Env>(s, key, a.lane_offset + (uint64_t)i, tick);
#pragma unroll
    for (int k = 0; k < S; ++k)
        if (state_row_own<Env>(k)) a.state[k * a.state_stride + i] = s[k];
    if constexpr (!Env::OBS_ALIASES_STATE) {
        typename Env::Real o[O];
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
__global__ __launch_bounds__(256) void reset_kernel(const ResetArgsT<typename Env::Real> a) {
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
__global__ __launch_bounds__(256) void observe_kernel(const typename Env::Real *state, int64_t sstride, typename Env::Real *obs, int64_t ostride, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    typename Env::Real s[Env::S], o[Env::O];
#pragma unroll
    for (int k = 0; k < Env::S; ++k) s[k] = state_row_src<Env>(state, sstride, obs, ostride, k)[i];
    Env::observe(s, o);
#pragma unroll
    for (int k = 0; k < Env::O; ++k) obs[k * ostride + i] = o[k];
}

// ---------------------------------------------------------------------------------------------
// Resident single-wave kernel (kernels.hpp: Mailbox).  One wave; lane i serves env i.  Commands arrive through the mailbox in
// coherent host memory: every poll is an ACQUIRE load of cmd_seq at system scope, so the loads of the command and of the actions
// that follow cannot be satisfied from anything older; results are published by a system-scope RELEASE fence executed by the
// whole wave followed by the store of done_seq.  The device-side state lives in the handle's ordinary arrays (so every other
// entry point sees it after the kernel has left); the engine tick is kept in a register and written back on the way out.
// ---------------------------------------------------------------------------------------------
template <class Env, bool AUTORESET, bool EXTRAS>
__global__ __launch_bounds__(64) void resident_kernel(const StepArgsT<typename Env::Real> a, const ResetArgsT<typename Env::Real> ra, Mailbox *mb,
                                                      const uint64_t idle_polls) {
    using Real = typename Env::Real;
    constexpr int O = Env::O;
    const int64_t lane = threadIdx.x;
    const bool mine = lane < a.n;
    Real *mb_obs = reinterpret_cast<Real *>(reinterpret_cast<char *>(mb) + kMailboxObsOffset);
    uint64_t tick = a.tick2[a.parity];
    uint64_t last = __hip_atomic_load(&mb->done_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    uint64_t idle = 0;
    for (;;) {
        const uint64_t seq = __hip_atomic_load(&mb->cmd_seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (seq == last) {                                     // wave-uniform (every lane read the same word)
            if (++idle > idle_polls) break;
            __builtin_amdgcn_s_sleep(2);
            continue;
        }
        idle = 0;
        const uint32_t cmd = __hip_atomic_load(&mb->cmd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (cmd == kMailboxExit) { last = seq; break; }
        if (mine) {
            if (cmd == kMailboxStep) step_body<Env, 1, AUTORESET, EXTRAS, 0, true, 0>(a, lane, tick);
            else if (cmd == kMailboxResetAll || (cmd == kMailboxResetDone && a.done[lane])) reset_lane<Env>(ra, lane, tick);
        }
        tick += 1;
        if (mine) {                                            // results: this lane's observation row, reward, done flag
            const Real *src = Env::OBS_ALIASES_STATE ? a.state_out : a.obs;
            const int64_t stride = Env::OBS_ALIASES_STATE ? a.state_stride : a.obs_stride;
#pragma unroll
            for (int k = 0; k < O; ++k) mb_obs[lane * O + k] = src[k * stride + lane];
            mb->reward[lane] = a.reward[lane];
            mb->done[lane] = a.done[lane];
        }
        if (lane == 0) mb->tick = tick;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");          // system scope, the whole wave: every lane's results are out
        if (lane == 0) __hip_atomic_store(&mb->done_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        last = seq;
    }
    if (lane == 0) {
        a.tick2[0] = tick; a.tick2[1] = tick;                  // both halves: whichever the next launch reads
        mb->tick = tick;
        __hip_atomic_store(&mb->done_seq, last, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&mb->exited, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

template <class Env>
static hipError_t launch_resident_env(bool autoreset, bool extras, const StepArgsT<typename Env::Real> &a, const ResetArgsT<typename Env::Real> &r,
                                      Mailbox *mb, uint64_t idle_polls, hipStream_t st) {
    if (a.n > kMailboxLanes) return hipErrorInvalidValue;
    const dim3 grid(1), blk(64);
    if (autoreset) { if (extras) hipLaunchKernelGGL((resident_kernel<Env, true, true>), grid, blk, 0, st, a, r, mb, idle_polls);
                     else hipLaunchKernelGGL((resident_kernel<Env, true, false>), grid, blk, 0, st, a, r, mb, idle_polls); }
    else           { if (extras) hipLaunchKernelGGL((resident_kernel<Env, false, true>), grid, blk, 0, st, a, r, mb, idle_polls);
                     else hipLaunchKernelGGL((resident_kernel<Env, false, false>), grid, blk, 0, st, a, r, mb, idle_polls); }
    return hipGetLastError();
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

// lanes per thread of an env's WIDE form: 16-byte accesses on the state rows — four floats, two doubles — or, for the env with a
// two-lane packed-FP32 form (Acrobot), two lanes per thread on dwordx2 streams
template <class Env>
constexpr int wide_of() { return (sizeof(typename Env::Real) == 8 || Env::PACKED2) ? 2 : 4; }

template <class Env>
static StepVariant resolve_variant(bool autoreset, bool extras, const LaunchCfg &cfg, int64_t n) {
    StepVariant v{};
    v.lds_tiles = 1; v.pipe_items = 1; v.vec = 1; v.nt = cfg.nt; v.resetf = 0;
    if constexpr (Env::PIPE_LANES) {       // multi-lane kernel, cfg.items lanes per thread (2..5)
        if (cfg.items > 1 && cfg.items <= 5 && !extras && cfg.vec == 1) {
            v.nt = 15;
            if (cfg.lds_pipe) v.lds_tiles = cfg.items; else v.pipe_items = cfg.items;
            return v;
        }
    }
    if constexpr (Env::PIPE_PAIRS) {
        // lane pairs: 2..4 pairs per thread, whole batches only — otherwise the ordinary forms below
        // (the form with the deferred reset — envs whose reset is two Philox calls, auto-reset — takes any batch size)
        const bool any_n = has_split_reset<Env>() && autoreset;
        if (cfg.items > 1 && cfg.items <= 4 && !extras && cfg.vec == 2 && !cfg.lds_pipe && n > 0 && (any_n || n % (2 * (int64_t)cfg.items * 256) == 0)) {
            // every stream non-temporal, except that the four-pair form with the deferred reset also exists with the other two masks
            // (the stream policy of a batch beyond one resident generation is a measured choice: capi.hip default_policy)
            v.nt = (any_n && cfg.items == 4) ? cfg.nt : 15; v.vec = 2; v.pipe_items = cfg.items; v.pipe_pairs = true;
            return v;
        }
    }
    if (cfg.vec > 1) v.vec = wide_of<Env>();
    // wave-compacted fused reset: wide variant of an env whose observation IS its state
    if (Env::OBS_ALIASES_STATE && !Env::PACKED2 && cfg.reset_form == 1 && v.vec == wide_of<Env>() && autoreset) v.resetf = 1;
    return v;
}

// the one-shot kernel with V lanes per thread: dispatch on (AUTORESET, EXTRAS, non-temporal mask, reset form)
template <class Env, int V>
static hipError_t launch_one_shot(const StepVariant &v, bool autoreset, bool extras, const LaunchCfg &cfg, const StepArgsT<typename Env::Real> &a,
                                  hipStream_t st) {
    const int64_t threads = (a.n + V - 1) / V;
    const dim3 grid(grid_for(threads > 0 ? threads : 1, cfg.block)), blk(cfg.block);
#define GYMNET_LAUNCH(AR, EX, NTM, RF) hipLaunchKernelGGL((step_kernel<Env, V, AR, EX, NTM, RF>), grid, blk, (size_t)cfg.lds_bytes, st, a)
#define GYMNET_LAUNCH_NT(AR, EX, RF)                                  \
    do {                                                              \
        if (v.nt == 15) GYMNET_LAUNCH(AR, EX, 15, RF);                \
        else if (v.nt == 12) GYMNET_LAUNCH(AR, EX, 12, RF);           \
        else GYMNET_LAUNCH(AR, EX, 0, RF);                            \
    } while (0)
    if constexpr (Env::OBS_ALIASES_STATE && !Env::PACKED2 && V > 1) {
        if (v.resetf == 1) {      // (resolve_variant: only with auto-reset)
            if (extras) GYMNET_LAUNCH_NT(true, true, 1); else GYMNET_LAUNCH_NT(true, false, 1);
            return hipGetLastError();
        }
    }
    if (extras) {   // bookkeeping variants follow the same stream policy (their own arrays stay cacheable)
        if (autoreset) GYMNET_LAUNCH_NT(true, true, 0); else GYMNET_LAUNCH_NT(false, true, 0);
    } else {
        if (autoreset) GYMNET_LAUNCH_NT(true, false, 0); else GYMNET_LAUNCH_NT(false, false, 0);
    }
#undef GYMNET_LAUNCH_NT
#undef GYMNET_LAUNCH
    return hipGetLastError();
}

template <class Env>
static hipError_t launch_step_env(bool autoreset, bool extras, const StepArgsT<typename Env::Real> &a, LaunchCfg cfg, hipStream_t st) {
    cfg = normalized(cfg);
    const StepVariant v = resolve_variant<Env>(autoreset, extras, cfg, a.n);
    if constexpr (Env::PIPE_PAIRS) {
        if (v.pipe_pairs) {
            // whole groups of 2 * items * 256 lanes (resolve_variant: any workgroup size divides the batch) — or, for the form with
            // the deferred reset, any batch: the last workgroups run the guarded body
            // one launch per resident generation (pipe2_chunks): every slice reads the same tick word and writes the same successor
            const Pipe2Chunks ch = pipe2_chunks(a.n, v.pipe_items, cfg.block, has_split_reset<Env>() && autoreset, cfg.simds);
            const int64_t per_block = 2 * (int64_t)v.pipe_items * cfg.block;
            for (int c = 0; c < ch.chunks; ++c) {
                const int64_t first = (int64_t)c * ch.lanes, count = (a.n - first < ch.lanes) ? a.n - first : ch.lanes;
                const StepArgsT<typename Env::Real> sa = ch.chunks == 1 ? a : slice_of(a, first, count);
                const dim3 qgrid((unsigned)((count + per_block - 1) / per_block)), qblk(cfg.block);
#define GYMNET_PIPE2(I)                                                                                                 \
    case I:                                                                                                             \
        if (autoreset) hipLaunchKernelGGL((step_kernel_pipe2<Env, I, true, 15>), qgrid, qblk, 0, st, sa);                \
        else hipLaunchKernelGGL((step_kernel_pipe2<Env, I, false, 15>), qgrid, qblk, 0, st, sa);                         \
        break;
                if constexpr (has_split_reset<Env>()) {
                    if (autoreset && v.pipe_items == 4 && v.nt != 15) {
                        if (v.nt == 12) hipLaunchKernelGGL((step_kernel_pipe2<Env, 4, true, 12>), qgrid, qblk, 0, st, sa);
                        else hipLaunchKernelGGL((step_kernel_pipe2<Env, 4, true, 0>), qgrid, qblk, 0, st, sa);
                        continue;
                    }
                }
                switch (v.pipe_items) {
                    GYMNET_PIPE2(2) GYMNET_PIPE2(3) GYMNET_PIPE2(4)
                    default: return hipErrorInvalidValue;
                }
#undef GYMNET_PIPE2
            }
            return hipGetLastError();
        }
    }
    if constexpr (Env::PIPE_LANES) {
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
    if (v.vec == 1) return launch_one_shot<Env, 1>(v, autoreset, extras, cfg, a, st);
    return launch_one_shot<Env, wide_of<Env>()>(v, autoreset, extras, cfg, a, st);
}

template <class Env>
static int describe_step_env(bool autoreset, bool extras, LaunchCfg cfg, int64_t n, char *buf, size_t cap) {
    cfg = normalized(cfg);
    const StepVariant v = resolve_variant<Env>(autoreset, extras, cfg, n);
    const char *env = Env::NAME;
    const char *ar = autoreset ? "true" : "false";
    if (v.pipe_pairs) {
        // " x G": the step is G launches of this kernel over consecutive slices of the batch (pipe2_chunks)
        const Pipe2Chunks ch = pipe2_chunks(n, v.pipe_items, cfg.block, has_split_reset<Env>() && autoreset, cfg.simds);
        if (ch.chunks > 1) return std::snprintf(buf, cap, "step_kernel_pipe2<%s,%d,%s,%d> x %d", env, v.pipe_items, ar, v.nt, ch.chunks);
        return std::snprintf(buf, cap, "step_kernel_pipe2<%s,%d,%s,%d>", env, v.pipe_items, ar, v.nt);
    }
    if (v.lds_tiles > 1) return std::snprintf(buf, cap, "step_kernel_lds<%s,%d,%s,15>", env, v.lds_tiles, ar);
    if (v.pipe_items > 1) return std::snprintf(buf, cap, "step_kernel_pipe<%s,%d,%s,15>", env, v.pipe_items, ar);
    return std::snprintf(buf, cap, "step_kernel<%s,%d,%s,%s,%d,%d>", env, v.vec, ar, extras ? "true" : "false", v.nt, v.resetf);
}

// lanes per thread on 16-byte rows the NEXT step launch resolves to, and (multi-lane forms) lanes or lane pairs per thread
template <class Env>
static void resolved_shape_env(bool autoreset, bool extras, LaunchCfg cfg, int64_t n, int *vec, int *sequential) {
    cfg = normalized(cfg);
    const StepVariant v = resolve_variant<Env>(autoreset, extras, cfg, n);
    *vec = v.vec;
    *sequential = v.lds_tiles > 1 ? v.lds_tiles : v.pipe_items;
}

// Lanes per thread of the fused rollout's FAT form — beyond the widest step-kernel form: four doubles per thread for the float64 env
// whose observation is the state.  0 = the env has none.  (Measured and NOT kept for float32: eight floats per thread save 5 of 102
// VALU per env-step and halve the waves — 2.6 -> 3.3 us per vector step, profiles/pmc_rollout_r05.txt.)
template <class Env>
constexpr int rollout_fat_lanes() { return (sizeof(typename Env::Real) == 8 && Env::OBS_ALIASES_STATE && !Env::PACKED2) ? 4 : 0; }

template <class Env>
static hipError_t launch_rollout_env(bool autoreset, bool extras, const StepArgsT<typename Env::Real> &a, const RolloutArgsT<typename Env::Real> &r,
                                     LaunchCfg cfg, hipStream_t st) {
    constexpr int WIDE = wide_of<Env>();
    const bool wide = cfg.vec == 4 || cfg.vec == 2;
    const bool sample = r.action_source != 0;
    const int64_t threads = (a.n + (wide ? WIDE : 1) - 1) / (wide ? WIDE : 1);
    const dim3 grid(grid_for(threads > 0 ? threads : 1, 256)), blk(256);
    const bool records = extras && r.ep_lane != nullptr;
#define GYMNET_ROLL(V, AR, RF)                                                                                                     \
    do {                                                                                                                          \
        if (records && r.records_no_overflow) { if (sample) hipLaunchKernelGGL((rollout_kernel<Env, V, AR, true, true, RF, 2>), grid, blk, 0, st, a, r); \
                       else hipLaunchKernelGGL((rollout_kernel<Env, V, AR, true, false, RF, 2>), grid, blk, 0, st, a, r); }        \
        else if (records) { if (sample) hipLaunchKernelGGL((rollout_kernel<Env, V, AR, true, true, RF, 1>), grid, blk, 0, st, a, r); \
                       else hipLaunchKernelGGL((rollout_kernel<Env, V, AR, true, false, RF, 1>), grid, blk, 0, st, a, r); }        \
        else if (extras) { if (sample) hipLaunchKernelGGL((rollout_kernel<Env, V, AR, true, true, RF>), grid, blk, 0, st, a, r);   \
                           else hipLaunchKernelGGL((rollout_kernel<Env, V, AR, true, false, RF>), grid, blk, 0, st, a, r); }       \
        else        { if (sample) hipLaunchKernelGGL((rollout_kernel<Env, V, AR, false, true, RF>), grid, blk, 0, st, a, r);       \
                      else hipLaunchKernelGGL((rollout_kernel<Env, V, AR, false, false, RF>), grid, blk, 0, st, a, r); }           \
    } while (0)
    if constexpr (rollout_fat_lanes<Env>() > 0) {
        // The FAT form (cfg.vec == rollout_fat_lanes: capi selects it when every stream is aligned for it): 8 floats / 4 doubles per
        // thread.  The state lives in registers for all T steps, so the lane count per thread only decides how many lanes share one
        // wave's per-step overheads — the wave-compacted reset pass, its LDS hand-off, the loop — and how many independent chains a
        // thread interleaves; the rollouts are instruction-issue bound (SQ counters, profiles/pmc_rollout_r05.txt: float64 187 -> 145
        // VALU per env-step).
        constexpr int FAT = rollout_fat_lanes<Env>();
        if (cfg.vec == FAT) {
            const int64_t threads4 = (a.n + FAT - 1) / FAT;
            const dim3 grid4(grid_for(threads4 > 0 ? threads4 : 1, 256));
#define GYMNET_ROLL4(AR, RF)                                                                                                        \
    do {                                                                                                                            \
        if (records && r.records_no_overflow) { if (sample) hipLaunchKernelGGL((rollout_kernel<Env, FAT, AR, true, true, RF, 2>), grid4, blk, 0, st, a, r);  \
                       else hipLaunchKernelGGL((rollout_kernel<Env, FAT, AR, true, false, RF, 2>), grid4, blk, 0, st, a, r); }         \
        else if (records) { if (sample) hipLaunchKernelGGL((rollout_kernel<Env, FAT, AR, true, true, RF, 1>), grid4, blk, 0, st, a, r);  \
                       else hipLaunchKernelGGL((rollout_kernel<Env, FAT, AR, true, false, RF, 1>), grid4, blk, 0, st, a, r); }         \
        else if (extras) { if (sample) hipLaunchKernelGGL((rollout_kernel<Env, FAT, AR, true, true, RF>), grid4, blk, 0, st, a, r);    \
                           else hipLaunchKernelGGL((rollout_kernel<Env, FAT, AR, true, false, RF>), grid4, blk, 0, st, a, r); }        \
        else        { if (sample) hipLaunchKernelGGL((rollout_kernel<Env, FAT, AR, false, true, RF>), grid4, blk, 0, st, a, r);        \
                      else hipLaunchKernelGGL((rollout_kernel<Env, FAT, AR, false, false, RF>), grid4, blk, 0, st, a, r); }            \
    } while (0)
            if (autoreset) { if (cfg.reset_form == 1) GYMNET_ROLL4(true, 1); else GYMNET_ROLL4(true, 0); }
            else GYMNET_ROLL4(false, 0);
#undef GYMNET_ROLL4
            return hipGetLastError();
        }
    }
    if (wide) {
        if (autoreset) {
            // the wave-compacted reset per step (cfg.reset_form = 1; envs whose observation IS the state, wide lanes)
            if constexpr (Env::OBS_ALIASES_STATE && !Env::PACKED2 && WIDE > 1) {
                if (cfg.reset_form == 1) { GYMNET_ROLL(WIDE, true, 1); return hipGetLastError(); }
            }
            GYMNET_ROLL(WIDE, true, 0);
        } else GYMNET_ROLL(WIDE, false, 0);
    } else {
        if (autoreset) GYMNET_ROLL(1, true, 0); else GYMNET_ROLL(1, false, 0);
    }
#undef GYMNET_ROLL
    return hipGetLastError();
}

template <class Env>
static hipError_t launch_reset_env(const ResetArgsT<typename Env::Real> &a, hipStream_t st) {
    const dim3 grid(grid_for(a.n > 0 ? (a.n + 3) / 4 : 1, 256)), blk(256);
    hipLaunchKernelGGL(reset_kernel<Env>, grid, blk, 0, st, a);
    return hipGetLastError();
}

template <class Env>
static hipError_t launch_observe_env(const typename Env::Real *state, int64_t sstride, typename Env::Real *obs, int64_t ostride, int64_t n,
                                     hipStream_t st) {
    if (n <= 0) return hipSuccess;
    if constexpr (Env::OBS_ALIASES_STATE) return hipSuccess;   // nothing to recompute
    else {
        hipLaunchKernelGGL(observe_kernel<Env>, dim3(grid_for(n, 256)), dim3(256), 0, st, state, sstride, obs, ostride, n);
        return hipGetLastError();
    }
}

}  // namespace gymnet

// One translation unit per env (env_*.hip) instantiates the kernels above and exports them under the env's tag; kernels.hip
// dispatches on (env_id, state scalar) — see GYMNET_DECLARE_ENV in kernels.hpp.
#define GYMNET_DEFINE_ENV(tag, Env)                                                                                                   \
    namespace gymnet {                                                                                                                \
    hipError_t launch_step_##tag(bool autoreset, bool extras, const StepArgsT<Env::Real> &a, LaunchCfg cfg, hipStream_t st) {         \
        return launch_step_env<Env>(autoreset, extras, a, cfg, st);                                                                   \
    }                                                                                                                                 \
    int describe_step_##tag(bool autoreset, bool extras, LaunchCfg cfg, int64_t n, char *buf, size_t cap) {                           \
        return describe_step_env<Env>(autoreset, extras, cfg, n, buf, cap);                                                           \
    }                                                                                                                                 \
    void resolved_shape_##tag(bool autoreset, bool extras, LaunchCfg cfg, int64_t n, int *vec, int *sequential) {                     \
        resolved_shape_env<Env>(autoreset, extras, cfg, n, vec, sequential);                                                          \
    }                                                                                                                                 \
    hipError_t launch_rollout_##tag(bool autoreset, bool extras, const StepArgsT<Env::Real> &a, const RolloutArgsT<Env::Real> &r,    \
                                    LaunchCfg cfg, hipStream_t st) {                                                                  \
        return launch_rollout_env<Env>(autoreset, extras, a, r, cfg, st);                                                             \
    }                                                                                                                                 \
    hipError_t launch_reset_##tag(const ResetArgsT<Env::Real> &a, hipStream_t st) { return launch_reset_env<Env>(a, st); }            \
    hipError_t launch_resident_##tag(bool autoreset, bool extras, const StepArgsT<Env::Real> &a, const ResetArgsT<Env::Real> &r,      \
                                     Mailbox *mb, uint64_t idle_polls, hipStream_t st) {                                              \
        return launch_resident_env<Env>(autoreset, extras, a, r, mb, idle_polls, st);                                                 \
    }                                                                                                                                 \
    hipError_t launch_observe_##tag(const Env::Real *state, int64_t sstride, Env::Real *obs, int64_t ostride, int64_t n,              \
                                    hipStream_t st) {                                                                                 \
        return launch_observe_env<Env>(state, sstride, obs, ostride, n, st);                                                          \
    }                                                                                                                                 \
    }
