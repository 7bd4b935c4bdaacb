// philox.hpp — Philox4x32-10 counter-based RNG (Salmon et al., SC'11), device + host.
//
// north_star replaces the reference's un-vendored NumSharp RNG (CartPoleEnv.cs:49,65,196-198) with a
// stateless per-lane generator: counter = (global lane lo, hi, tick lo, hi), key = (seed lo, hi).
// No RNG state lives in HBM; a reset costs ten rounds of 2 mul_hi + 2 mul_lo.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gymnet {

struct PhiloxWords { uint32_t w[4]; };

// One round = two 32x32->64 products.  Written as 64-bit multiplies so that gfx950 emits ONE v_mad_u64_u32 per
// product instead of a v_mul_hi_u32 + v_mul_lo_u32 pair: the multiplies dominate a Philox pass.  (Measured, tools/issue_rate_probe.hip:
// v_mad_u64_u32 holds the SIMD ~5 cycles per wave — NOT the 16 of a quarter-rate instruction, as rounds 1-5 assumed.)
//   UNIFORM_KEY (device code, round 6): the key is the same for every lane of the kernel (a kernel argument: the handle's seed, the
//   action seed).  The compiler then folds the ten round keys k + r * W into TWENTY scalar registers per keyed stream and keeps them
//   live across the kernel's main loop; with a reset stream, an action stream and an aux stream that is sixty SGPRs, the rollout
//   kernels ran out (106 of 106) and spilled scalars into VGPR lanes — hundreds of v_readlane_b32, VALU instructions all, in the loop
//   (the sampled-action rollout: 432).  An empty asm statement per round makes each round key the product of two s_add_u32 on the
//   spot (scalar ALU, issued beside the vector work), two SGPRs per stream.  Only valid for a uniform key: the "s" constraint would
//   read one lane's value of a per-lane key.  Same integers either way.
template <bool UNIFORM_KEY = false>
__host__ __device__ __forceinline__ PhiloxWords philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                              uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
#if defined(__HIP_DEVICE_COMPILE__)
        // a ^ b ^ c as ONE v_bitop3_b32 (truth table 0x96): gfx950 has no v_xor3_b32 and LLVM emits two v_xor_b32 per word — 60
        // VALU per call instead of 40
        const uint32_t n0 = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c1, k0, 0x96);
        const uint32_t n2 = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c3, k1, 0x96);
#else
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
#endif
        c0 = n0; c1 = (uint32_t)p1; c2 = n2; c3 = (uint32_t)p0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (UNIFORM_KEY) asm volatile("" : "+s"(k0), "+s"(k1));
#endif
    }
    PhiloxWords o; o.w[0] = c0; o.w[1] = c1; o.w[2] = c2; o.w[3] = c3;
    return o;
}

template <bool UNIFORM_KEY = false>
__host__ __device__ __forceinline__ PhiloxWords lane_words(uint64_t seed, uint64_t lane, uint64_t tick) {
    return philox4x32_10<UNIFORM_KEY>((uint32_t)lane, (uint32_t)(lane >> 32), (uint32_t)tick, (uint32_t)(tick >> 32),
                                      (uint32_t)seed, (uint32_t)(seed >> 32));
}

// Independent streams by purpose: reset draws use the caller's key unchanged (stream 0); space sampling and the
// epsilon-greedy composer XOR a constant into it, so ActionSpace.Sample() with the env's own (seed, tick) can never replay the
// words that produced a reset state.
//
// ACTION STREAM v2 (ABI 6).  A sampled action consumes ONE 32-bit word, a Philox4x32-10 call yields four: the call is
// therefore shared by the four consecutive GLOBAL lanes of a group,
//     word A of global lane L at tick t = word (L & 3) of Philox(key = seed ^ kStreamAction, counter = (L >> 2, t))
//     word B of global lane L at tick t = word (L & 3) of Philox(key = seed ^ kStreamAux,    counter = (L >> 2, t))
// A is the ActionSpace.Sample() word (Discrete.cs:27 randint, Box.cs:85 uniform, the first uniform of the other Box regimes); B is
// the second word of the consumers that need one (the epsilon-greedy coin of TrainingPlaySession.cs:46, the second uniform of
// Box.cs:82's normal) and is only drawn by them.  A thread that owns four aligned lanes (the dwordx4 forms) pays one call per step
// instead of four; v1 drew a whole call per lane (counter (L, t)) and used words 0 and 1 of it.
constexpr uint64_t kStreamReset = 0ull, kStreamAction = 0x9E3779B97F4A7C15ull, kStreamAux = 0xD6E8FEB86659FD93ull;
template <bool UNIFORM_KEY = false>
__host__ __device__ __forceinline__ PhiloxWords stream_words(uint64_t stream, uint64_t seed, uint64_t lane, uint64_t tick) {
    return lane_words<UNIFORM_KEY>(seed ^ stream, lane, tick);
}
// the four A (or B) words of the group of global lanes 4 * group .. 4 * group + 3
template <bool UNIFORM_KEY = false>
__host__ __device__ __forceinline__ PhiloxWords action_group_words(uint64_t seed, uint64_t group, uint64_t tick) {
    return stream_words<UNIFORM_KEY>(kStreamAction, seed, group, tick);
}
template <bool UNIFORM_KEY = false>
__host__ __device__ __forceinline__ PhiloxWords aux_group_words(uint64_t seed, uint64_t group, uint64_t tick) {
    return stream_words<UNIFORM_KEY>(kStreamAux, seed, group, tick);
}
// word (L & 3) of a group's call, for the forms that serve one lane at a time
__host__ __device__ __forceinline__ uint32_t word_of(const PhiloxWords &r, uint32_t k) {
    return k == 0 ? r.w[0] : k == 1 ? r.w[1] : k == 2 ? r.w[2] : r.w[3];
}
template <bool UNIFORM_KEY = false>
__host__ __device__ __forceinline__ uint32_t action_word(uint64_t seed, uint64_t lane, uint64_t tick) {
    return word_of(action_group_words<UNIFORM_KEY>(seed, lane >> 2, tick), (uint32_t)lane & 3u);
}
template <bool UNIFORM_KEY = false>
__host__ __device__ __forceinline__ uint32_t aux_word(uint64_t seed, uint64_t lane, uint64_t tick) {
    return word_of(aux_group_words<UNIFORM_KEY>(seed, lane >> 2, tick), (uint32_t)lane & 3u);
}

// 24-bit uniform in [0,1): exactly representable in binary32
__host__ __device__ __forceinline__ float u01_24(uint32_t r) { return (float)(r >> 8) * (1.0f / 16777216.0f); }

// The epsilon-greedy coin as ONE integer compare.  u01_24(w) <= epsilon, for 0 <= epsilon, holds exactly when (w >> 8) <= floor(epsilon * 2^24)
// (the scaling by 2^24 is exact and the left side is an integer), i.e. when w <= coin_threshold(epsilon): the shift, the conversion and
// the multiply of u01_24 leave the per-lane path, the threshold is computed once per launch.  Callers validate 0 <= epsilon <= 1.
__host__ __device__ __forceinline__ uint32_t coin_threshold(float epsilon) {
    const float t = __builtin_floorf(epsilon * 16777216.0f);
    if (!(t >= 0.0f)) return 0u;                              // (never for a validated epsilon; a NaN explores like epsilon = 0)
    return t >= 16777215.0f ? 0xFFFFFFFFu : (((uint32_t)t << 8) | 0xFFu);
}

}  // namespace gymnet
