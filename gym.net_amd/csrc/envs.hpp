// envs.hpp — per-lane dynamics of the classic-control environments, binary32, one env per GPU lane.
//
// This translation unit is compiled with -ffp-contract=off: every operation rounds to binary32 on
// its own, in the association order written here, so the arithmetic is reproducible and can be
// bounded tightly against the reference's binary64 arithmetic (<= 1e-5 abs per step, north_star).
//
// CartPole follows src/Gym.Environments/Envs/Classic/CartPoleEnv.cs (paths relative to the Gym.NET
// tree): constants :24-36 (the C# `const float` bit patterns), Step :137-186, Reset :63-67.
// Pendulum / MountainCar / Acrobot do not exist in the reference (README.md:69-76, unchecked roadmap
// items); they follow the upstream openai/gym classic_control algorithms.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "philox.hpp"

namespace gymnet {

// ---------------------------------------------------------------------------------------------
// sin/cos — the reference calls Math.Sin / Math.Cos (CartPoleEnv.cs:147-148).  The step kernel at 2^20
// lanes is balanced between its memory time and its ALU chain, so every instruction
// counts: OCML's sincosf costs ~50 VALU per lane on its fast path.  This is a 3-constant Cody-Waite
// reduction to [-pi/4, pi/4] plus the classic degree-7 / degree-8 minimax polynomials: 22 VALU, built
// only from IEEE mul / fma / round-to-nearest-even, so the CPU restatement the tests check against
// reproduces it BIT FOR BIT.  Measured against float64 sin/cos: |abs error| <= 9.3e-8 for every
// |x| <= 1e5, <= 1.5 ulp for |x| <= 10.  sin(-0) returns +0.  |x| > 65536 (or inf) falls back to OCML.
// ---------------------------------------------------------------------------------------------

// ---------------------------------------------------------------------------------------------
// One source for one env per lane (float) and TWO envs per lane (f2 = two floats in a VGPR pair, arithmetic on
// v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32).  Each element of a packed operation is the same IEEE operation as its
// scalar form: the two-lane code is bit-identical to running the scalar code twice (tests + the full-size checksum).
// Not packed on this ISA (done per element): rint, float -> int conversion, integer / bit operations, selects.
// MEASURED, and the reason the two-lane form is opt-in (GYMNET_VEC=2) rather than the default: it halves nothing that
// matters.  The Acrobot step's arithmetic alone (tools/acrobot_alu_probe.hip, 8 waves per SIMD, no memory traffic) takes
// 7.4 us per 2^20 env-steps as 454 scalar instructions (2.4 SIMD-cycles each in this mix) and 8.2 us as 287 packed + 287
// scalar instructions per PAIR: a v_pk_*_f32 occupies the SIMD about as long as the two scalar instructions it replaces
// (~5 cycles, tools/valu_probe.hip), and the per-element leftovers are not free.  docs/ledger.md §4a.
// ---------------------------------------------------------------------------------------------
typedef float f2 __attribute__((ext_vector_type(2)));
namespace vm {
__device__ __forceinline__ float splat(float, float v) { return v; }
__device__ __forceinline__ f2 splat(f2, float v) { return f2{v, v}; }
__device__ __forceinline__ float fma(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ f2 fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
template <class T> __device__ __forceinline__ T fmak(float k, T b, T c) { return fma(splat(b, k), b, c); }   // k*b + c
template <class T> __device__ __forceinline__ T fmac(T a, T b, float k) { return fma(a, b, splat(a, k)); }   // a*b + k
}  // namespace vm

// quadrant fix-up of one element: sin x = {s, c, -s, -c}[q], cos x = {c, -s, -c, s}[q], q = n mod 4 (see sincos_f32)
__device__ __forceinline__ void sincos_quadrant(float s, float c, float n, float &s_out, float &c_out) {
    const uint32_t q = (uint32_t)(int)n;
    const uint32_t q30 = q << 30, q31 = q << 31;
    const bool odd = (int32_t)q31 < 0;
    const float ss = odd ? c : s, cc = odd ? s : c;
    s_out = __uint_as_float(__builtin_amdgcn_bitop3_b32(__float_as_uint(ss), q30, 0x80000000u, 0x78));
    c_out = __uint_as_float(__builtin_amdgcn_bitop3_b32(__float_as_uint(cc), q30 ^ q31, 0x80000000u, 0x78));
}

// sincos_f32<true> for T = float or f2: the identical operation sequence per element (reduction and both polynomials on
// the packed instructions for f2; rint and the quadrant logic per element).
template <class T>
__device__ __forceinline__ void sincos_bounded(T x, T &s_out, T &c_out) {
    T n;
    const T xs = x * vm::splat(x, 0.636619772367581343f);
    if constexpr (sizeof(T) == sizeof(float)) n = rintf(xs);
    else n = T{rintf(xs.x), rintf(xs.y)};
    T r = vm::fmak(-1.5703125f, n, x);
    r = vm::fmak(-4.837512969970703125e-4f, n, r);
    r = vm::fmak(-7.54978995489188216e-8f, n, r);
    const T z = r * r;
    T ps = vm::fmac(vm::splat(z, -1.9515295891e-4f), z, 8.3321608736e-3f);
    ps = vm::fmac(ps, z, -1.6666654611e-1f);
    const T s = vm::fma(r * z, ps, r);
    T pc = vm::fmac(vm::splat(z, 2.443315711809948e-5f), z, -1.388731625493765e-3f);
    pc = vm::fmac(pc, z, 4.166664568298827e-2f);
    const T c = vm::fma(z * z, pc, vm::fmac(vm::splat(z, -0.5f), z, 1.0f));
    if constexpr (sizeof(T) == sizeof(float)) {
        sincos_quadrant(s, c, n, s_out, c_out);
    } else {
        float s0, c0, s1, c1;
        sincos_quadrant(s.x, c.x, n.x, s0, c0);
        sincos_quadrant(s.y, c.y, n.y, s1, c1);
        s_out = T{s0, s1};
        c_out = T{c0, c1};
    }
}

// sin/cos for SMALL arguments, |x| < 24 (|n| <= 15): Acrobot's wrapped angles and RK4 stage angles (|x| < 12 for every state
// the dynamics can produce).  Two instructions shorter than sincos_bounded, per call: a TWO-constant Cody-Waite reduction —
// with n at most 4 bits wide, C1 = pi/2 cut to 20 bits makes n*C1 exact, and the one remaining constant carries the next 24
// bits (|r - (x - n pi/2)| < 2^-40) — and the cosine polynomial in Horner form (4 fma instead of 3 fma + 2 mul).  Same
// minimax coefficients; |error| <= 1.2e-7 against float64 over |x| <= 16 (tests/test_oracle.py).  Its results differ from
// sincos_bounded's in the last bit for some arguments, so the CPU twin has its own restatement (ref_sincos_f32_small).
template <class T>
__device__ __forceinline__ void sincos_small(T x, T &s_out, T &c_out) {
    T n;
    const T xs = x * vm::splat(x, 0.636619772367581343f);
    if constexpr (sizeof(T) == sizeof(float)) n = rintf(xs);
    else n = T{rintf(xs.x), rintf(xs.y)};
    T r = vm::fmak(-0x1.921fap+0f, n, x);                 // pi/2 to 20 bits: n * C1 is exact for |n| <= 15
    r = vm::fmak(-0x1.54442ep-20f, n, r);                 // pi/2 - C1
    const T z = r * r;
    T ps = vm::fmac(vm::splat(z, -1.9515295891e-4f), z, 8.3321608736e-3f);
    ps = vm::fmac(ps, z, -1.6666654611e-1f);
    const T s = vm::fma(r * z, ps, r);
    T pc = vm::fmac(vm::splat(z, 2.443315711809948e-5f), z, -1.388731625493765e-3f);
    pc = vm::fmac(pc, z, 4.166664568298827e-2f);
    pc = vm::fmac(pc, z, -0.5f);
    const T c = vm::fmac(pc, z, 1.0f);
    if constexpr (sizeof(T) == sizeof(float)) {
        sincos_quadrant(s, c, n, s_out, c_out);
    } else {
        float s0, c0, s1, c1;
        sincos_quadrant(s.x, c.x, n.x, s0, c0);
        sincos_quadrant(s.y, c.y, n.y, s1, c1);
        s_out = T{s0, s1};
        c_out = T{c0, c1};
    }
}

// sin / cos for |x| <= 0.785 (< pi/4), where the range reduction of sincos_bounded is the identity: n = rint(x * 2/pi) = 0
// (|x * 2/pi| < 0.4998), each fma(-C, 0, r) returns r bit for bit (also for +-0), and the quadrant fix-up with q = 0 changes
// nothing.  So evaluating the two polynomials on x itself gives EXACTLY the bits of sincos_f32(x) — 14 instructions shorter
// (multiply, rint, three fma, float->int, eight quadrant instructions).  CartPole's pole angle is below 0.21 rad until the
// episode ends, so its step takes this path whenever every lane of the wave qualifies (kernels.hip: advance_all).
constexpr float kSmallAngle = 0.785f;
__device__ __forceinline__ void sincos_tiny(float x, float &s_out, float &c_out) {
    const float z = x * x;
    float ps = fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
    ps = fmaf(ps, z, -1.6666654611e-1f);
    s_out = fmaf(x * z, ps, x);
    float pc = fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    pc = fmaf(pc, z, 4.166664568298827e-2f);
    c_out = fmaf(z * z, pc, fmaf(-0.5f, z, 1.0f));
}

// BOUNDED = true drops the OCML fallback: for callers whose argument is bounded by construction (Acrobot's wrapped
// angles and RK4 stage angles, |x| < 16).  Same bits as the full version for every |x| <= 65536; beyond that the result
// is unspecified (finite garbage or NaN, never a hang).  It exists because the ten inlined Payne-Hanek fallbacks made the
// ALU-bound Acrobot kernel 2.5x longer than its hot path.
template <bool BOUNDED = false>
__device__ __forceinline__ void sincos_f32(float x, float &s_out, float &c_out) {
    if constexpr (!BOUNDED) {
        if (__builtin_expect(fabsf(x) > 65536.0f, 0)) {   // never taken by a sane rollout; keeps huge / infinite angles defined
            sincosf(x, &s_out, &c_out);
            return;
        }
    }
    sincos_bounded<float>(x, s_out, c_out);
}
__device__ __forceinline__ float sin_f32(float x) { float s, c; sincos_f32(x, s, c); return s; }
__device__ __forceinline__ float cos_f32(float x) { float s, c; sincos_f32(x, s, c); return c; }

// x / C for a compile-time constant C, as two IEEE operations instead of the 11-instruction division
// sequence: q = fma(x, zh, x * zl) with zh = RN(1/C), zl = RN(1/C - zh) (Brisebarre, Muller, Raina:
// "Accelerating correctly rounded floating-point division when the divisor is known in advance").
// For C = total_mass = 0x1.19999ap+0 the result equals IEEE x / C for EVERY float with |x| >= 2^-104
// (and for +-0, +-inf, NaN): checked exhaustively over all 2^23 significands per binade by the CPU tests.
// Below 2^-104 (4.9e-32) the x*zl term goes subnormal and the last bit can differ.
template <typename Dummy = void>
struct DivByTotalMass {
    static constexpr float C = 0.1f + 1.0f;
    static constexpr float ZH = 1.0f / C;
    static constexpr float ZL = (float)(1.0 / (double)C - (double)ZH);
    __device__ __forceinline__ static float apply(float x) { return fmaf(x, ZH, x * ZL); }
};

// ---------------------------------------------------------------------------------------------
// CartPole-v1  (CartPoleEnv.cs)
// ---------------------------------------------------------------------------------------------
struct CartPole {
    static constexpr int S = 4;              // x, x_dot, theta, theta_dot       (:141-144)
    static constexpr int O = 4;              // observation == state             (:166,185)
    static constexpr bool OBS_ALIASES_STATE = true;
    static constexpr bool HAS_SBD = true;    // steps_beyond_done state machine  (:41,168-183)
    static constexpr bool BOX_ACTION = false;
    static constexpr bool PACKED2 = false;   // no two-lane packed-FP32 form (the kernel is memory-bound)
    static constexpr bool PIPE_LANES = false, PIPE_PAIRS = false;   // no multi-lane kernel forms (ditto)
    using Action = int32_t;                  // Discrete(2)                      (:47)
    using Real = float;                      // state scalar of this engine (CartPole64 in cartpole64.hpp: the reference's float64)
    static constexpr bool RESET_TAKES_KEY = false;                  // reset() takes the four words of one Philox call
    static constexpr const char *NAME = "CartPole";
    static constexpr int32_t ACTION_N = 2;                          // Discrete(2).Sample(): start 0 + randint(0, 2)  (Discrete.cs:17-28)

    // :24-36 — the float32 values of the C# consts (total_mass, polemass_length const-folded in float)
    static constexpr float gravity = 9.8f;
    static constexpr float masspole = 0.1f;
    static constexpr float total_mass = 0.1f + 1.0f;
    static constexpr float length = 0.5f;
    static constexpr float polemass_length = 0.1f * 0.5f;
    static constexpr float force_mag = 10.0f;
    static constexpr float tau = 0.02f;
    static constexpr float theta_threshold = 0.20943951606750488f;   // (float)(12*2*Math.PI/360) = 0x1.aceeap-3
    static constexpr float x_threshold = 2.4f;

    __device__ __forceinline__ static float div_tm(float x) { return DivByTotalMass<>::apply(x); }

    // The state component whose magnitude decides whether the step may use sincos_tiny (SMALL_ANGLE), and the bound
    static constexpr bool HAS_SMALL_ANGLE_PATH = true;
    static constexpr int ANGLE_ROW = 2;
    static constexpr float SMALL_ANGLE_BOUND = kSmallAngle;

    // :146-167.  Any action != 1 pushes left (validity is only Debug.Assert'ed, :139).
    // SMALL_ANGLE: the caller guarantees |theta| <= kSmallAngle; the results are bit-identical either way (sincos_tiny).
    // (AUTORESET: accepted for the interface it shares with CartPole64, where it drops a guard; nothing depends on it here)
    template <bool SMALL_ANGLE = false, bool AUTORESET = false>
    __device__ __forceinline__ static void step(float (&s)[S], Action a, float &reward, bool &done) {
        const float x = s[0], x_dot = s[1], theta = s[2], theta_dot = s[3];
        const float force = (a == 1) ? force_mag : -force_mag;                                   // :146
        float sintheta, costheta;
        if constexpr (SMALL_ANGLE) sincos_tiny(theta, sintheta, costheta);
        else sincos_f32(theta, sintheta, costheta);                                              // :147-148
        // `/ total_mass` below is IEEE division by a constant, evaluated as DivByTotalMass (bit-identical)
        const float temp = div_tm(force + polemass_length * theta_dot * theta_dot * sintheta);   // :149
        const float thetaacc = (gravity * sintheta - costheta * temp)
                               / (length * (4.0f / 3.0f - div_tm(masspole * costheta * costheta))); // :150
        const float xacc = temp - div_tm(polemass_length * thetaacc * costheta);                // :151
        // explicit Euler (:32,153-158): positions advance with the OLD velocities
        const float nx = x + tau * x_dot;
        const float nx_dot = x_dot + tau * xacc;
        const float ntheta = theta + tau * theta_dot;
        const float ntheta_dot = theta_dot + tau * thetaacc;
        s[0] = nx; s[1] = nx_dot; s[2] = ntheta; s[3] = ntheta_dot;                              // :166
        // :167 — the INTEGER output.  The reference forms x + tau * x_dot and theta + tau * theta_dot in binary64 (:154,156:
        // double state, tau widened from its float const) and compares THOSE with the widened float thresholds.  The flag is
        // therefore derived from the same two binary64 sums, not from the binary32 nx / ntheta stored above: for every
        // float32-representable input it is the reference's flag exactly (tau * x_dot is a 48-bit product, exact in binary64,
        // so fma(tau, x_dot, x) IS the reference's x + tau * x_dot; |v| > thr is the four strict comparisons, false for NaN).
        // Comparing the rounded binary32 values instead flips the flag for inputs within ~2 float32 ulps of a threshold
        // (22 of the 200 such vectors in tests/golden/cartpole_reference_text.npz).  4 conversions + 2 v_fma_f64 + 2 v_cmp.
        const double vx = __builtin_fma((double)tau, (double)x_dot, (double)x);
        const double vtheta = __builtin_fma((double)tau, (double)theta_dot, (double)theta);
        done = __builtin_fabs(vx) > (double)x_threshold || __builtin_fabs(vtheta) > (double)theta_threshold;
        reward = 1.0f;   // the steps_beyond_done rule (:168-183) is applied by the kernel, which owns sbd
    }

    // :63-67 — state = uniform(-0.05, 0.05, 4) as low + (high-low)*u
    __device__ __forceinline__ static void reset(float (&s)[S], const PhiloxWords &r) {
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k] = -0.05f + 0.1f * u01_24(r.w[k]);
    }

    __device__ __forceinline__ static void observe(const float (&s)[S], float (&o)[O]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = s[k];
    }
    // observation of a state reset() has just drawn (bounded by construction: an env may use cheaper trigonometry here)
    __device__ __forceinline__ static void observe_fresh(const float (&s)[S], float (&o)[O]) { observe(s, o); }
    // dynamics + observation of the new state in one call (Acrobot shares its trigonometry between the two)
    __device__ __forceinline__ static void step_observe(float (&s)[S], Action a, float &reward, bool &done, float (&o)[O]) {
        step(s, a, reward, done);
        observe(s, o);
    }
};

// ---------------------------------------------------------------------------------------------
// Pendulum-v1  (upstream gym; semi-implicit Euler; never terminates)
// ---------------------------------------------------------------------------------------------
struct Pendulum {
    static constexpr int S = 2;              // theta, theta_dot
    static constexpr int O = 3;              // cos, sin, theta_dot
    static constexpr bool OBS_ALIASES_STATE = false;
    static constexpr int OBS_ROW_OF_STATE[S] = {-1, 2};   // theta_dot IS obs[2]: stored once, in the observation array
    static constexpr bool HAS_SBD = false;
    static constexpr bool BOX_ACTION = true;
    static constexpr bool PACKED2 = false;
    static constexpr bool PIPE_LANES = false, PIPE_PAIRS = false;
    static constexpr bool HAS_SMALL_ANGLE_PATH = false;
    using Action = float;                    // Box(-2, 2, (1,))
    using Real = float;
    static constexpr bool RESET_TAKES_KEY = false;
    static constexpr const char *NAME = "Pendulum";
    static constexpr float ACTION_LOW = -2.0f, ACTION_HIGH = 2.0f;  // Box(-2, 2, (1,)).Sample(): the bounded regime, uniform(low, high)  (Box.cs:85)
    static constexpr float PI = 3.14159265358979323846f;

    // fmodf(a, m) for the ONE modulus the env uses (m = 2 pi), in ~11 instructions instead of OCML's iterative reduction, and
    // EXACT like fmod itself (so the CPU restatement keeps calling libm's fmodf and the bits agree): q = trunc(|a| * RN(1/m)) is
    // the true truncated quotient up to +-1 for |a| < 2^22 m (relative error of the product 2^-23, so absolute error < 1/2);
    // r = fma(-q, m, |a|) is exact for the true q (the remainder of an IEEE fmod is representable) and, for a q off by one,
    // lands below 0 or at / above m — rounding cannot move it across either bound, m being representable — so one compare pair
    // repairs q and a second fma gives the exact remainder.  Larger, infinite or NaN arguments take fmodf.
    __device__ __forceinline__ static float fmod_2pi(float a) {
        constexpr float m = 2.0f * PI, inv_m = 1.0f / m;
        const float ax = fabsf(a);
        if (__builtin_expect(!(ax < 4194304.0f * m), 0)) return fmodf(a, m);
        float q = truncf(ax * inv_m);
        float r = fmaf(-q, m, ax);
        q = r < 0.0f ? q - 1.0f : (r >= m ? q + 1.0f : q);
        r = fmaf(-q, m, ax);
        return copysignf(r, a);
    }

    __device__ __forceinline__ static float floored_mod_2pi(float a) {
        float r = fmod_2pi(a);
        if (r < 0.0f) r += 2.0f * PI;
        return r;
    }

    __device__ __forceinline__ static void step(float (&s)[S], Action a, float &reward, bool &done) {
        const float max_speed = 8.0f, max_torque = 2.0f, dt = 0.05f;
        const float th = s[0], thdot = s[1];
        const float u = a < -max_torque ? -max_torque : (a > max_torque ? max_torque : a);
        const float nrm = floored_mod_2pi(th + PI) - PI;
        const float costs = nrm * nrm + 0.1f * (thdot * thdot) + 0.001f * (u * u);
        float newthdot = thdot + (15.0f * sin_f32(th) + 3.0f * u) * dt;     // 3g/(2l) = 15, 3/(m l^2) = 3
        newthdot = newthdot < -max_speed ? -max_speed : (newthdot > max_speed ? max_speed : newthdot);
        const float newth = th + newthdot * dt;
        s[0] = newth; s[1] = newthdot;
        reward = -costs;
        done = false;
    }

    __device__ __forceinline__ static void reset(float (&s)[S], const PhiloxWords &r) {
        s[0] = -PI + (2.0f * PI) * u01_24(r.w[0]);
        s[1] = -1.0f + 2.0f * u01_24(r.w[1]);
    }

    __device__ __forceinline__ static void observe(const float (&s)[S], float (&o)[O]) {
        float sn, cs;
        sincos_f32(s[0], sn, cs);
        o[0] = cs; o[1] = sn; o[2] = s[1];
    }
    // observation of a state reset() has just drawn (bounded by construction: an env may use cheaper trigonometry here)
    __device__ __forceinline__ static void observe_fresh(const float (&s)[S], float (&o)[O]) { observe(s, o); }
    __device__ __forceinline__ static void step_observe(float (&s)[S], Action a, float &reward, bool &done, float (&o)[O]) {
        step(s, a, reward, done);
        observe(s, o);
    }
};

// ---------------------------------------------------------------------------------------------
// MountainCar-v0  (upstream gym)
// ---------------------------------------------------------------------------------------------
struct MountainCar {
    static constexpr int S = 2;              // position, velocity
    static constexpr int O = 2;
    static constexpr bool OBS_ALIASES_STATE = true;
    static constexpr bool HAS_SBD = false;
    static constexpr bool BOX_ACTION = false;
    static constexpr bool PACKED2 = false;
    static constexpr bool PIPE_LANES = false, PIPE_PAIRS = false;
    static constexpr bool HAS_SMALL_ANGLE_PATH = false;
    using Action = int32_t;                  // Discrete(3)
    using Real = float;
    static constexpr bool RESET_TAKES_KEY = false;
    static constexpr const char *NAME = "MountainCar";
    static constexpr int32_t ACTION_N = 3;

    __device__ __forceinline__ static void step(float (&s)[S], Action a, float &reward, bool &done) {
        float p = s[0], v = s[1];
        v += (float)(a - 1) * 0.001f + cos_f32(3.0f * p) * (-0.0025f);
        v = v < -0.07f ? -0.07f : (v > 0.07f ? 0.07f : v);
        p += v;
        p = p < -1.2f ? -1.2f : (p > 0.6f ? 0.6f : p);
        if (p == -1.2f && v < 0.0f) v = 0.0f;
        s[0] = p; s[1] = v;
        reward = -1.0f;
        done = p >= 0.5f && v >= 0.0f;
    }

    __device__ __forceinline__ static void reset(float (&s)[S], const PhiloxWords &r) {
        s[0] = -0.6f + 0.2f * u01_24(r.w[0]);
        s[1] = 0.0f;
    }

    __device__ __forceinline__ static void observe(const float (&s)[S], float (&o)[O]) { o[0] = s[0]; o[1] = s[1]; }
    // observation of a state reset() has just drawn (bounded by construction: an env may use cheaper trigonometry here)
    __device__ __forceinline__ static void observe_fresh(const float (&s)[S], float (&o)[O]) { observe(s, o); }
    __device__ __forceinline__ static void step_observe(float (&s)[S], Action a, float &reward, bool &done, float (&o)[O]) {
        step(s, a, reward, done);
        observe(s, o);
    }
};

// ---------------------------------------------------------------------------------------------
// Acrobot-v1  (upstream gym; "book" dynamics; one classical RK4 step of dt = 0.2)
// ---------------------------------------------------------------------------------------------
struct Acrobot {
    static constexpr int S = 4;              // theta1, theta2, dtheta1, dtheta2
    static constexpr int O = 6;              // cos1, sin1, cos2, sin2, dtheta1, dtheta2
    static constexpr bool OBS_ALIASES_STATE = false;
    static constexpr int OBS_ROW_OF_STATE[S] = {-1, -1, 4, 5};   // dtheta1, dtheta2 ARE obs[4], obs[5]: stored once, in the observation array
    static constexpr bool HAS_SBD = false;
    static constexpr bool BOX_ACTION = false;
    static constexpr bool PACKED2 = true;    // step_observe_x2: two envs per thread on v_pk_*_f32
    static constexpr bool PIPE_LANES = true;  // step_kernel_pipe / _lds: ITEMS lanes per thread, loads / arithmetic / stores overlapped
    static constexpr bool PIPE_PAIRS = true;  // step_kernel_pipe2: the same over lane pairs
    static constexpr bool HAS_SMALL_ANGLE_PATH = false;
    using Action = int32_t;                  // Discrete(3): torque = a - 1
    using Real = float;
    static constexpr bool RESET_TAKES_KEY = false;
    static constexpr const char *NAME = "Acrobot";
    static constexpr int32_t ACTION_N = 3;
    static constexpr float PI = 3.14159265358979323846f;

    // m1 = m2 = l1 = I1 = I2 = 1, lc1 = lc2 = 0.5, g = 9.8 folded into the literals.
    // Acrobot is the one kernel of the four with real arithmetic (RK4 = 4 x dsdt): ~7.4 us of VALU issue per 2^20 lanes
    // against ~11.6 us of memory time for its 65 B per lane, in a launch of two lock-step wave generations where about half of
    // the arithmetic ends up exposed.  The lever that works is the instruction count per env-step (682 -> 454 in round 2), so
    // dsdt and the RK4 combination are written for it:
    //  - upstream's cos(th1 + th2 - pi/2) and cos(th1 - pi/2) are sin(th1 + th2) = s1*c2 + c1*s2 and sin(th1);
    //  - numerator and denominator of ddth2 are multiplied through by d1, so ONE reciprocal R = 1/(d1*det) serves both
    //    accelerations (1/det = R*d1, 1/d1 = R*det) instead of two IEEE divisions (11 instructions each), and that
    //    reciprocal is recip_p() below: 6 fma for the narrow range d1*det lives in;
    //  - every a*b+c is an explicit fmaf (one instruction, one rounding), products are factored
    //    (B^2 s2/2 + A B s2 = s2 B (B/2 + A));
    //  - sincos without the OCML fallback (arguments are bounded here).
    // ~35 instructions + two sincos per stage instead of ~54 + two.  Mathematically identical to upstream; in float32 it
    // differs from the literal transcription by rounding only; the float64 oracle keeps upstream's literal formula and the
    // float32 "kernel semantics" twin of the CPU test infrastructure mirrors THIS sequence operation for operation.
    // 1/P for P = d1 * det.  With c2 in [-1, 1]: d1 = c2 + 3.5 in [2.5, 4.5], det = 2.8125 - c2^2/4 in [2.5625, 2.8125], so
    // P lies in [6.4, 11.6] — no scaling, no special cases.  Quadratic minimax seed on [6.25, 11.75] (relative error 7.7e-3)
    // + two Newton steps r <- r + r(1 - P r): 6 full-rate fma instead of the 10-instruction IEEE division sequence around a
    // quarter-rate v_rcp_f32.  Result within 0.55 ulp of 1/P over the whole range (2e6 random P, docs/ledger.md §4a),
    // and — being fma only — reproduced bit for bit by the CPU restatement.
    template <class T>
    __device__ __forceinline__ static T recip_p(T P) {
        T r = vm::fmac(vm::fmac(P, vm::splat(P, 0x1.82ab8p-10f), -0x1.4640b2p-5f), P, 0x1.6677a2p-2f);
        r = vm::fma(r, vm::fmac(-P, r, 1.0f), r);
        r = vm::fma(r, vm::fmac(-P, r, 1.0f), r);
        return r;
    }

    // T = float: one env; T = f2: two envs of one thread, every line below one packed instruction per operation.
    template <class T>
    __device__ __forceinline__ static void dsdt(const T (&s)[4], T torque, T (&d)[4]) {
        const T th1 = s[0], th2 = s[1], A = s[2], B = s[3];
        T s1, c1, s2, c2;
        sincos_small<T>(th1, s1, c1);
        sincos_small<T>(th2, s2, c2);
        const T d1 = c2 + vm::splat(c2, 3.5f);                                    // 0.25 + (1.25 + c2) + 2
        const T d2 = vm::fmak(0.5f, c2, vm::splat(c2, 1.25f));                    // 0.25 + 0.5 c2 + 1
        const T phi2 = vm::splat(s1, 4.9f) * vm::fma(s1, c2, c1 * s2);            // m2 lc2 g sin(th1 + th2)
        const T phi1 = vm::fma(-(s2 * B), vm::fmak(0.5f, B, A), vm::fmak(14.7f, s1, phi2));   // -s2 B (B/2 + A) + 14.7 s1 + phi2
        const T h = vm::fma(-(vm::splat(A, 0.5f) * A), A * s2, torque - phi2);    // torque - A^2 s2 / 2 - phi2
        const T det = vm::fmak(1.25f, d1, -(d2 * d2));                            // (m2 lc2^2 + I2) d1 - d2^2
        const T num = vm::fma(h, d1, d2 * phi1);
        const T R = recip_p<T>(d1 * det);
        const T ddth2 = num * (R * d1);
        const T ddth1 = -vm::fma(d2, ddth2, phi1) * (R * det);
        d[0] = A; d[1] = B; d[2] = ddth1; d[3] = ddth2;
    }

    // upstream loops `while x > M: x -= diff` / `while x < m: x += diff` with m = -M.  One RK4 step of dt = 0.2 moves a clamped
    // state by at most ~10 rad, i.e. two wraps; this is the same repeated subtraction, at most FOUR times and written without a
    // loop: identical results whenever four suffice (every state the dynamics can produce), a non-finite or absurd state cannot
    // hang the GPU, and — no back-edge — the compiler's s_waitcnt bookkeeping stays exact across the step.  Only one of the two
    // upstream loops can ever run, and x + diff == -(|x| - diff) bit for bit, so the magnitude is wrapped and the sign put
    // back (the subtraction of a magnitude in (M, M + diff] lands in (-M, M]: the sign of the RESULT may flip, which is why
    // the sign is applied by multiplication-free xor of the original sign bit): 14 instructions per angle instead of 24.
    __device__ __forceinline__ static float wrap(float x, float M) {
        const float diff = M + M;
        const uint32_t sign = __float_as_uint(x) & 0x80000000u;
        float ax = fabsf(x);
        if (ax > M) { ax -= diff; if (ax > M) { ax -= diff; if (ax > M) { ax -= diff; if (ax > M) ax -= diff; } } }
        return __uint_as_float(__float_as_uint(ax) ^ sign);
    }

    // One RK4 step of dt = 0.2 with the torque held constant: y <- s + dt/6 (k1 + 2 k2 + 2 k3 + k4), 4 fma per component.
    template <class T>
    __device__ __forceinline__ static void rk4(const T (&s)[4], T torque, T (&y)[4]) {
        constexpr float dt = 0.2f;
        T k1[4], k2[4], k3[4], k4[4];
        dsdt<T>(s, torque, k1);
#pragma unroll
        for (int i = 0; i < 4; ++i) y[i] = vm::fmak(dt / 2.0f, k1[i], s[i]);
        dsdt<T>(y, torque, k2);
#pragma unroll
        for (int i = 0; i < 4; ++i) y[i] = vm::fmak(dt / 2.0f, k2[i], s[i]);
        dsdt<T>(y, torque, k3);
#pragma unroll
        for (int i = 0; i < 4; ++i) y[i] = vm::fmak(dt, k3[i], s[i]);
        dsdt<T>(y, torque, k4);
#pragma unroll
        for (int i = 0; i < 4; ++i) y[i] = vm::fmak(dt / 6.0f, vm::fmak(2.0f, k3[i], vm::fmak(2.0f, k2[i], k1[i])) + k4[i], s[i]);
    }

    // wrap the angles, clamp the velocities (per element: compares and selects have no packed form)
    __device__ __forceinline__ static void wrap_clamp(float (&y)[4]) {
        constexpr float mv1 = 4.0f * PI, mv2 = 9.0f * PI;
        y[0] = wrap(y[0], PI);
        y[1] = wrap(y[1], PI);
        // v_med3_f32: the clamp in one instruction; identical to the compare/select pair for every non-NaN velocity
        y[2] = __builtin_amdgcn_fmed3f(y[2], -mv1, mv1);
        y[3] = __builtin_amdgcn_fmed3f(y[3], -mv2, mv2);
    }

    // One env: RK4, wrap / clamp; the sin/cos of the new angles serve both the termination test and the observation.
    __device__ __forceinline__ static void step_observe(float (&s)[S], Action a, float &reward, bool &done, float (&o)[O]) {
#ifdef GYMNET_PROBE_ACROBOT_NOMATH   // probe builds only: the kernel's memory traffic with (almost) no arithmetic
        { const float t = (float)(a - 1) * 1e-3f; s[0] += t; s[1] -= t; s[2] += t; s[3] -= t;
          done = s[0] > 3.0f; reward = -1.0f; o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; o[3] = s[3]; o[4] = s[2]; o[5] = s[3]; return; }
#endif
        float y[4];
        rk4<float>(s, (float)(a - 1), y);
        wrap_clamp(y);
#pragma unroll
        for (int i = 0; i < 4; ++i) s[i] = y[i];
        // done = -cos(th1) - cos(th2 + th1) > 1, with cos(th1 + th2) = c1*c2 - s1*s2
        float s1, c1, s2, c2;
        sincos_small<float>(y[0], s1, c1);
        sincos_small<float>(y[1], s2, c2);
        done = (-c1 - fmaf(c1, c2, -(s1 * s2))) > 1.0f;
        reward = done ? 0.0f : -1.0f;
        o[0] = c1; o[1] = s1; o[2] = c2; o[3] = s2; o[4] = y[2]; o[5] = y[3];
    }

    // TWO envs of one thread at once (kernels.hip: advance_all with VEC == 2): the same operations as step_observe, the
    // arithmetic of both envs in the two halves of packed registers.  Bit-identical to two step_observe calls.
    __device__ __forceinline__ static void step_observe_x2(float (&s)[S][2], Action (&a)[2], float (&reward)[2], bool (&done)[2],
                                                           float (&o)[O][2]) {
        f2 sv[4], yv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) sv[i] = f2{s[i][0], s[i][1]};
        rk4<f2>(sv, f2{(float)(a[0] - 1), (float)(a[1] - 1)}, yv);
        float y0[4], y1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { y0[i] = yv[i].x; y1[i] = yv[i].y; }
        wrap_clamp(y0);
        wrap_clamp(y1);
#pragma unroll
        for (int i = 0; i < 4; ++i) { s[i][0] = y0[i]; s[i][1] = y1[i]; }
        f2 s1, c1, s2, c2;
        sincos_small<f2>(f2{y0[0], y1[0]}, s1, c1);
        sincos_small<f2>(f2{y0[1], y1[1]}, s2, c2);
        const f2 t = -c1 - vm::fma(c1, c2, -(s1 * s2));
        done[0] = t.x > 1.0f; done[1] = t.y > 1.0f;
        reward[0] = done[0] ? 0.0f : -1.0f; reward[1] = done[1] ? 0.0f : -1.0f;
        o[0][0] = c1.x; o[0][1] = c1.y; o[1][0] = s1.x; o[1][1] = s1.y; o[2][0] = c2.x; o[2][1] = c2.y;
        o[3][0] = s2.x; o[3][1] = s2.y; o[4][0] = y0[2]; o[4][1] = y1[2]; o[5][0] = y0[3]; o[5][1] = y1[3];
    }

    __device__ __forceinline__ static void step(float (&s)[S], Action a, float &reward, bool &done) {
        float o[O];
        step_observe(s, a, reward, done, o);
    }

    __device__ __forceinline__ static void reset(float (&s)[S], const PhiloxWords &r) {
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k] = -0.1f + 0.2f * u01_24(r.w[k]);
    }

    // any state (set_state hands over arbitrary angles).  Angles a step or a reset can produce (|x| <= pi, far inside sincos_small's
    // |x| < 24) take the step's OWN sin / cos, so SetState(GetState()) reproduces the observation the step wrote bit for bit
    // (sincos_f32 and sincos_small differ in the last bit for some arguments); anything larger keeps the full-range form.
    __device__ __forceinline__ static void observe(const float (&s)[S], float (&o)[O]) {
        float s1, c1, s2, c2;
        if (fabsf(s[0]) < 24.0f) sincos_small<float>(s[0], s1, c1); else sincos_f32(s[0], s1, c1);
        if (fabsf(s[1]) < 24.0f) sincos_small<float>(s[1], s2, c2); else sincos_f32(s[1], s2, c2);
        o[0] = c1; o[1] = s1; o[2] = c2; o[3] = s2; o[4] = s[2]; o[5] = s[3];
    }
    // a freshly reset state: angles in [-0.1, 0.1) — the step's own small-argument sin/cos
    __device__ __forceinline__ static void observe_fresh(const float (&s)[S], float (&o)[O]) {
        float s1, c1, s2, c2;
        sincos_small<float>(s[0], s1, c1);
        sincos_small<float>(s[1], s2, c2);
        o[0] = c1; o[1] = s1; o[2] = c2; o[3] = s2; o[4] = s[2]; o[5] = s[3];
    }
};

}  // namespace gymnet
