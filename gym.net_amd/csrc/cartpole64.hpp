// cartpole64.hpp — CartPole in the REFERENCE'S OWN arithmetic: binary64 state, the literal operation sequence of
// src/Gym.Environments/Envs/Classic/CartPoleEnv.cs:141-167 (paths relative to the Gym.NET tree), binary32-valued constants
// widened at use exactly as C# widens its `const float`s.  Selected per handle with GYMNET_FLAG_F64 (include/gymnet_amd.h).
//
// Why it exists: the reference keeps `state` as a float64 NDArray and returns THAT as the
// observation (:166,185).  The float32 engine is within 1e-5 per teacher-forced step, but free-running float32 leaves 1e-5 after
// ~50 steps and can end an episode one step early or late; this mode reproduces the reference's states to the last few ulps and
// therefore its episode lengths, free-running, for as long as the caller likes.  It moves 73 B per env-step instead of 41.
//
// Compiled with -ffp-contract=off like the rest of the library: every operation below is one IEEE-754 binary64 operation in
// the order written, `/` is correctly rounded division, so the CPU restatement the tests check against (the test
// infrastructure's float64 "kernel semantics" twin) reproduces every result BIT FOR BIT.  The single place where this file and the reference can
// differ is sin / cos: the reference calls Math.Sin / Math.Cos (the platform's libm, <= 1 ulp), this file evaluates its own
// (below, < 0.75 ulp for the angles of a live episode) — a last-bit difference in sin(theta) that moves a state by <= 1e-17 per step.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "philox.hpp"

namespace gymnet {

// ---------------------------------------------------------------------------------------------
// sin / cos in binary64 from IEEE mul / add / fma / rint only.
//   reduction: n = rint(x * 2/pi); r = ((x - n P1) - n P2) - n P3 with pi/2 = P1 + P2 + P3 + ..., P1 and P2 cut to 33 bits so
//     that n * P1 and n * P2 are exact for |n| < 2^20 (the fma then rounds once per step); |x| <= 2^19 * pi/2.
//   kernels on |r| <= pi/4: the minimax polynomials published with Sun's fdlibm (k_sin.c / k_cos.c: degree 13 / 14, the same
//     coefficients every libm descended from it uses), evaluated in Horner form; the cosine as w + ((1 - w) - z/2 + z^2 C(z))
//     with w = 1 - z/2, which keeps the rounding of 1 - z/2 out of the result.
// Accuracy against a 200-bit reference (tests/test_oracle.py): < 0.75 ulp for |x| <= pi/4 — every CartPole pole angle before
// the episode ends: n = 0, the reduction is the identity and the error is the polynomials' — and <= 2.1 ulp over |x| <= 8e5
// (the reduced argument is kept in one double, without a tail).
// Larger arguments (a pole left spinning for ~1e5 steps after `done`), infinities and NaN take the OCML routines.
// ---------------------------------------------------------------------------------------------
// SMALL = true: the caller guarantees |x| <= kSmallAngle64 (< pi/4).  There n = rint(x * 2/pi) = +-0, each fma(-n, P, r) returns r
// bit for bit — for r = +-0 it returns +0, which `x + 0.0` reproduces (IEEE: (-0) + (+0) = +0; never folded away without
// fast-math) — and the quadrant fix-up with q = 0 changes nothing: the SAME bits as the general path, without the multiply, the
// rint, three fma, the float -> int conversion and the quadrant selects.  CartPole's pole angle is below 0.21 rad until the
// episode ends, so the step takes this path whenever every lane of the wave qualifies (kernels64.hip), like the float32 kernel.
constexpr double kSmallAngle64 = 0.785;
template <bool SMALL = false>
__device__ __forceinline__ void sincos_f64(double x, double &s_out, double &c_out) {
    double n = 0.0, r;
    if constexpr (SMALL) {
        r = x + 0.0;
    } else {
        if (__builtin_expect(!(__builtin_fabs(x) <= 823549.0), 0)) {
            s_out = ::sin(x);
            c_out = ::cos(x);
            return;
        }
        n = __builtin_rint(x * 6.36619772367581382433e-01);
        r = __builtin_fma(-n, 1.57079632673412561417e+00, x);         // P1: first 33 bits of pi/2  (0x3FF921FB54400000)
        r = __builtin_fma(-n, 6.07710050630396597660e-11, r);         // P2: next 33 bits            (0x3DD0B4611A600000)
        r = __builtin_fma(-n, 2.02226624879595063154e-21, r);         // P3: pi/2 - P1 - P2 rounded  (0x3BA3198A2E037073)
    }
    const double z = r * r;
    // sin r = r + r^3 (S1 + z (S2 + z (S3 + z (S4 + z (S5 + z S6)))))
    double ps = __builtin_fma(1.58969099521155010221e-10, z, -2.50507602534068634195e-08);
    ps = __builtin_fma(ps, z, 2.75573137070700676789e-06);
    ps = __builtin_fma(ps, z, -1.98412698298579493134e-04);
    ps = __builtin_fma(ps, z, 8.33333333332248946124e-03);
    ps = __builtin_fma(ps, z, -1.66666666666666324348e-01);
    const double s = __builtin_fma(r * z, ps, r);
    // cos r = w + (((1 - w) - z/2) + z (z (C1 + z (C2 + z (C3 + z (C4 + z (C5 + z C6))))))),  w = 1 - z/2
    double pc = __builtin_fma(-1.13596475577881948265e-11, z, 2.08757232129817482790e-09);
    pc = __builtin_fma(pc, z, -2.75573143513906633035e-07);
    pc = __builtin_fma(pc, z, 2.48015872894767294178e-05);
    pc = __builtin_fma(pc, z, -1.38888888888741095749e-03);
    pc = __builtin_fma(pc, z, 4.16666666666666019037e-02);
    const double hz = 0.5 * z;
    const double w = 1.0 - hz;
    const double c = w + (((1.0 - w) - hz) + z * (z * pc));
    if constexpr (SMALL) { s_out = s; c_out = c; return; }
    // quadrant: sin x = {s, c, -s, -c}[n mod 4], cos x = {c, -s, -c, s}[n mod 4]
    const int q = (int)n & 3;
    const double ss = (q & 1) ? c : s, cc = (q & 1) ? s : c;
    s_out = (q & 2) ? -ss : ss;
    c_out = ((q + 1) & 2) ? -cc : cc;
}

// 53-bit uniform in [0, 1) from two Philox words, the construction NumPy's random_sample() uses for its doubles
// ((a >> 5) * 2^26 + (b >> 6)) / 2^53 — the reference draws its reset state with NumSharp's port of that generator,
// CartPoleEnv.cs:65): exactly representable, so low + (high - low) * u below rounds once per operation.
__host__ __device__ __forceinline__ double u01_53(uint32_t a, uint32_t b) {
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
}
// second Philox call of a float64 reset draw: same counter (lane, tick), key ^ this constant (a stream of its own)
constexpr uint64_t kStreamReset64 = 0xC2B2AE3D27D4EB4Full;

// x / total_mass for the binary64 x of the float64 mode, as TWO IEEE operations instead of the ~30-instruction correctly rounded
// division sequence: q = fma(x, ZH, x * ZL) with ZH = RN(1/C), ZL = RN(1/C - ZH) (Brisebarre, Muller, Raina: "Accelerating correctly
// rounded floating-point division when the divisor is known in advance"; the float32 engine's DivByTotalMass, envs.hpp).
// Three of the step's four divisions are by this constant (:149-151): ~55 of its ~270 VALU instructions per env-step.
//
// Why q equals IEEE x / C for EVERY binary64 x with 2^-900 <= |x| <= 2^1000 (and for +0 and NaN) — PROVED, not sampled
// (tools/prove_div_total_mass_f64.py does the arithmetic below in exact rationals; tests/test_oracle.py runs it and 1e8 samples):
//   C = total_mass is a binary32 value widened to binary64: C = Cn / 2^23 with the ODD integer Cn = 9227469.  For x = X * 2^e
//   (X a 53-bit integer) the exact quotient is X * 2^(e+23) / Cn.  A rounding breakpoint of binary64 in the quotient's binade is
//   m * 2^k with m an odd integer (a midpoint between neighbouring doubles), so the distance between the quotient and ANY
//   breakpoint is |X * 2^j - m * Cn| / Cn * 2^k with j = 24 or 25: the numerator is an ODD integer (X * 2^j is even, m * Cn is
//   odd * odd), so it is at least 1 in magnitude and the quotient stays >= 1/(2 Cn) = 2^-24.1 ulp away from every breakpoint
//   (attained: the tool constructs such x).  The fma pair computes RN(x * ZH + RN(x * ZL)) = RN(x / C + err) with
//   |err| <= |x| * |1/C - ZH - ZL| + ulp(x * ZL) / 2 < 2^-53 ulp of the quotient.
//   An error 2^29 times smaller than the distance to the nearest breakpoint cannot change the rounding: q == x / C.
//   (A divisor with 53 significant bits would not allow this argument: the 24-bit constant is what makes it a theorem.)
// Below 2^-900 the product x * ZL approaches the subnormal range and the bound degrades; the step never divides such a value
// (the three dividends are 10 + ..., masspole * cos^2 and polemass_length * thetaacc * cos: >= 2^-200 in magnitude or exactly 0).
// Outside the theorem because ZL < 0: x = -0 yields +0 (the division: -0) and x = +-inf would yield NaN (the division: +-inf).
// The zero cannot change a step's result: two of the quotients are subtracted from a non-zero term (4/3 - ..., temp - ...: a zero of
// either sign leaves it unchanged), the third dividend is +-10 + ... (never zero).  The infinity CAN be reached from a finite state
// (ADVICE r5): without auto-reset a lane stepped far past done grows without bound, polemass_length * theta_dot^2 * sin(theta)
// overflows, and where the reference carries +-inf on (and reports done: inf > threshold) a NaN would compare false.  apply()
// therefore returns x itself for an infinite x (= x / C, C > 0): one v_cmp_class per call and a branch nobody takes.  The CPU twin
// mirrors both, so GPU == twin holds for every input; tests/test_gpu_f64.py steps such states against the plain division.
struct DivByTotalMass64 {
    static constexpr double C = (double)(0.1f + 1.0f);                  // 0x1.19999ap+0 exactly
    static constexpr double ZH = 1.0 / C;                               // RN(1/C)  = 0x1.d1745c6e043b8p-1
    // RN(1/C - ZH), written out: the compile-time evaluation of (1/C - ZH) in long double is not portable between host and
    // device passes.  tools/prove_div_total_mass_f64.py derives both constants in exact arithmetic and checks these literals.
    static constexpr double ZL = -0x1.4633f3e678be9p-55;
    // GUARD_INF = false: for the kernels that fuse the auto-reset.  There an infinite dividend cannot matter: it needs theta_dot^2 (or
    // the acceleration it feeds) to overflow, |theta_dot| > 1e154, and such a lane's theta + tau * theta_dot is beyond the threshold in
    // the SAME step — done is computed from the positions, which use the OLD velocities — so its whole state is overwritten by the
    // reset draw.  (The one observable trace: a terminal observation kept under GYMNET_FLAG_FINAL_OBS shows NaN velocities for such a
    // lane where the division would show +-inf.)  The guard costs three v_cmp_class + branches per env-step — 8 of 142 VALU in the
    // float64 rollout, 5.14 -> 5.44 us per vector step — and stays wherever the state survives the step (no auto-reset).
    template <bool GUARD_INF = true>
    __host__ __device__ __forceinline__ static double apply(double x) {
        if constexpr (GUARD_INF) {
            if (__builtin_expect(__builtin_isinf(x), 0)) return x;      // x / C for an infinite x (the fma pair would say NaN: inf - inf)
        }
        return __builtin_fma(x, ZH, x * ZL);
    }
};

struct CartPole64 {
    // the Env interface of step_kernels.hpp (see CartPole in envs.hpp), with the state scalar the reference uses
    using Real = double;
    using Action = int32_t;                  // Discrete(2)                      (:47)
    static constexpr int S = 4;              // x, x_dot, theta, theta_dot       (:141-144)
    static constexpr int O = 4;              // observation == state             (:166,185)
    static constexpr bool OBS_ALIASES_STATE = true;
    static constexpr bool HAS_SBD = true;    // steps_beyond_done state machine  (:41,168-183)
    static constexpr bool BOX_ACTION = false;
    static constexpr bool PACKED2 = false;
    static constexpr bool PIPE_LANES = false;
    static constexpr bool PIPE_PAIRS = true; // step_kernel_pipe2: ITEMS lane pairs per thread (13.1 vs 14.4 us at 2^20 lanes, round 4)
    static constexpr bool RESET_TAKES_KEY = true;                   // reset() makes its own two Philox calls (53-bit uniforms)
    static constexpr const char *NAME = "CartPole64";
    static constexpr int32_t ACTION_N = 2;
    static constexpr bool HAS_SMALL_ANGLE_PATH = true;
    static constexpr int ANGLE_ROW = 2;
    static constexpr double SMALL_ANGLE_BOUND = kSmallAngle64;
    // :24-36 — the float32 VALUES of the C# consts (total_mass, polemass_length const-folded in float), widened at use
    static constexpr float gravity = 9.8f;
    static constexpr float masspole = 0.1f;
    static constexpr float total_mass = 0.1f + 1.0f;
    static constexpr float length = 0.5f;
    static constexpr float polemass_length = 0.1f * 0.5f;
    static constexpr float force_mag = 10.0f;
    static constexpr float tau = 0.02f;
    static constexpr float theta_threshold = 0.20943951606750488f;   // (float)(12 * 2 * Math.PI / 360)
    static constexpr float x_threshold = 2.4f;

    template <bool GUARD_INF>
    __device__ __forceinline__ static double div_tm(double x) { return DivByTotalMass64::apply<GUARD_INF>(x); }

    // :141-167, statement for statement; C#'s usual arithmetic conversions written out (float op double -> double)
    // AUTORESET: the caller overwrites a done lane's state in the same launch (see DivByTotalMass64::apply on what that allows)
    template <bool SMALL_ANGLE = false, bool AUTORESET = false>
    __device__ __forceinline__ static void step(double (&st)[S], int32_t a, float &reward, bool &done) {
        constexpr bool G = !AUTORESET;
        double x = st[0], x_dot = st[1], theta = st[2], theta_dot = st[3];                                  // :141-144
        const float force = a == 1 ? force_mag : -force_mag;                                                // :146
        double sintheta, costheta;
        sincos_f64<SMALL_ANGLE>(theta, sintheta, costheta);                                                 // :147-148
        // `/ total_mass` (:149-151) is IEEE division by a constant, evaluated as DivByTotalMass64 (proved bit-identical, above)
        const double temp = div_tm<G>((double)force + (double)polemass_length * theta_dot * theta_dot * sintheta);                    // :149
        const double thetaacc = ((double)gravity * sintheta - costheta * temp)
                                / ((double)length * (4.0 / 3.0 - div_tm<G>((double)masspole * costheta * costheta)));              // :150
        const double xacc = temp - div_tm<G>((double)polemass_length * thetaacc * costheta);                     // :151
        x = x + (double)tau * x_dot;                                                                        // :154
        x_dot = x_dot + (double)tau * xacc;                                                                 // :155
        theta = theta + (double)tau * theta_dot;                                                            // :156
        theta_dot = theta_dot + (double)tau * thetaacc;                                                     // :157
        st[0] = x; st[1] = x_dot; st[2] = theta; st[3] = theta_dot;                                         // :166
        done = x < -(double)x_threshold || x > (double)x_threshold
               || theta < -(double)theta_threshold || theta > (double)theta_threshold;                      // :167
        reward = 1.0f;   // the steps_beyond_done rule (:168-183) is applied by the kernel, which owns sbd
    }

    // :63-67 — state = uniform(-0.05, 0.05, 4) in float64: low + (high - low) * u, u with 53 random bits
    __device__ __forceinline__ static void reset(double (&st)[S], uint64_t key, uint64_t lane, uint64_t tick) {
        const PhiloxWords a = lane_words(reset_call_key(key, 0), lane, tick);   // the words the float32 engine draws from, too
        const PhiloxWords b = lane_words(reset_call_key(key, 1), lane, tick);
        reset_from_words(st, a, b);
    }
    // The same draw in its parts, for the kernels that spread ONE reset over two lanes (step_kernels.hpp, the wave-compacted resets:
    // lane 2r makes call 0, lane 2r + 1 call 1, so a Philox pass of the wave serves 32 resets instead of costing two passes).
    static constexpr int RESET_CALLS = 2;
    __host__ __device__ __forceinline__ static uint64_t reset_call_key(uint64_t key, uint32_t call) { return call ? key ^ kStreamReset64 : key; }
    __host__ __device__ __forceinline__ static void reset_from_words(double (&st)[S], const PhiloxWords &a, const PhiloxWords &b) {
#pragma unroll
        for (int k = 0; k < 4; ++k) st[k] = -0.05 + (0.05 - -0.05) * u01_53(a.w[k], b.w[k]);
    }

    __device__ __forceinline__ static void observe(const double (&s)[S], double (&o)[O]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = s[k];
    }
    __device__ __forceinline__ static void observe_fresh(const double (&s)[S], double (&o)[O]) { observe(s, o); }
};

}  // namespace gymnet
