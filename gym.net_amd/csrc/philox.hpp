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
// product (quarter-rate) instead of a v_mul_hi_u32 + v_mul_lo_u32 pair: the multiplies dominate a Philox pass.
__host__ __device__ __forceinline__ PhiloxWords philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                              uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c0 = n0; c1 = (uint32_t)p1; c2 = n2; c3 = (uint32_t)p0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    PhiloxWords o; o.w[0] = c0; o.w[1] = c1; o.w[2] = c2; o.w[3] = c3;
    return o;
}

__host__ __device__ __forceinline__ PhiloxWords lane_words(uint64_t seed, uint64_t lane, uint64_t tick) {
    return philox4x32_10((uint32_t)lane, (uint32_t)(lane >> 32), (uint32_t)tick, (uint32_t)(tick >> 32),
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
__host__ __device__ __forceinline__ PhiloxWords stream_words(uint64_t stream, uint64_t seed, uint64_t lane, uint64_t tick) {
    return lane_words(seed ^ stream, lane, tick);
}
// the four A (or B) words of the group of global lanes 4 * group .. 4 * group + 3
__host__ __device__ __forceinline__ PhiloxWords action_group_words(uint64_t seed, uint64_t group, uint64_t tick) {
    return stream_words(kStreamAction, seed, group, tick);
}
__host__ __device__ __forceinline__ PhiloxWords aux_group_words(uint64_t seed, uint64_t group, uint64_t tick) {
    return stream_words(kStreamAux, seed, group, tick);
}
// word (L & 3) of a group's call, for the forms that serve one lane at a time
__host__ __device__ __forceinline__ uint32_t word_of(const PhiloxWords &r, uint32_t k) {
    return k == 0 ? r.w[0] : k == 1 ? r.w[1] : k == 2 ? r.w[2] : r.w[3];
}
__host__ __device__ __forceinline__ uint32_t action_word(uint64_t seed, uint64_t lane, uint64_t tick) {
    return word_of(action_group_words(seed, lane >> 2, tick), (uint32_t)lane & 3u);
}
__host__ __device__ __forceinline__ uint32_t aux_word(uint64_t seed, uint64_t lane, uint64_t tick) {
    return word_of(aux_group_words(seed, lane >> 2, tick), (uint32_t)lane & 3u);
}

// 24-bit uniform in [0,1): exactly representable in binary32
__host__ __device__ __forceinline__ float u01_24(uint32_t r) { return (float)(r >> 8) * (1.0f / 16777216.0f); }

}  // namespace gymnet
