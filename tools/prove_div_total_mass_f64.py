#!/usr/bin/env python3
"""PROOF (exact rational arithmetic, no floating point in the argument) that the float64 CartPole kernel's constant division

        q = fma(x, ZH, x * ZL),   ZH = RN(1/C),  ZL = RN(1/C - ZH),  C = total_mass = (double)(0.1f + 1.0f)

(gym.net_amd/csrc/cartpole64.hpp: DivByTotalMass64; CartPoleEnv.cs:149-151 divide by total_mass three times per step) returns the
correctly rounded IEEE-754 binary64 quotient x / C for EVERY binary64 x whose products stay in the normal range.

The method is Brisebarre / Muller / Raina, "Accelerating correctly rounded floating-point division when the divisor is known in
advance" (IEEE TC 2004).  For a general 53-bit divisor their theorem leaves a finite set of exceptional significands that has to be
enumerated (best rational approximations — continued-fraction convergents — of the divisor).  Here the divisor is a binary32
CONSTANT widened to binary64, and that makes the exceptional set EMPTY by a counting argument:

  1. C = Cn / 2^23 with Cn = 9227469, an ODD integer (C's binary32 significand).
  2. Take x = X * 2^e, X an integer in [2^52, 2^53).  Scaling by 2^e is exact on both sides, so take e = 0: Q = X / C = X * 2^23 / Cn,
     which lies in [2^51.86, 2^52.86): binade [2^51, 2^52) has ulp 1/2, binade [2^52, 2^53) has ulp 1.
  3. The rounding breakpoints (midpoints of neighbouring doubles) are m/4 (lower binade) or m/2 (upper binade), m an ODD integer.
        Q - m/2 = (X * 2^24 - m * Cn) / (2 Cn),        Q - m/4 = (X * 2^25 - m * Cn) / (4 Cn).
     X * 2^24 (or 2^25) is even, m * Cn is odd * odd = odd: the numerator is an odd integer, |numerator| >= 1.  Hence
        dist(Q, nearest breakpoint) >= 1 / (2 Cn) ulp  ~  2^-24.14 ulp            (D_MIN below; attained, see hardest_cases()).
  4. The fma computes q = RN(q') with q' = X * ZH + t EXACTLY, t = RN(X * ZL).  With delta = 1/C - ZH - ZL:
        |q' - Q| <= X * |delta| + ulp(X * ZL) / 2  <=  2^53 |delta| + 2^-55       (ERR_MAX below, ~2^-54 = 2^-53 ulp at most).
  5. ERR_MAX < D_MIN (by a factor of ~2^28): q' lies strictly on the same side of every breakpoint as Q, and is never ON one, so
     RN(q') = RN(Q).  QED.

Range: the scaling in step 2 is exact while x * ZL is a normal number and nothing overflows, i.e. 2^-966 < |x| < 2^1023 (the header
states the conservative 2^-900 .. 2^1000).  +0 and NaN give the division's result too; because ZL < 0, x = -0 gives +0 (division:
-0) and x = +-inf gives NaN (division: +-inf) — unobservable in the step, see the header comment in cartpole64.hpp.

The script evaluates every quantity above with fractions.Fraction, checks that the literals in cartpole64.hpp and in the oracle's
twin are exactly RN(1/C) and RN(1/C - ZH), then CONSTRUCTS the hardest dividends (X with X * 2^24 = +-1 mod Cn: quotients at the
minimal distance 1/(2 Cn) ulp from a breakpoint — the candidates an enumeration would produce), a sweep of the convergents of C,
and random dividends, and checks on each that the exactly simulated fma pair equals the exactly rounded quotient.
Exit status 0 = proved and verified.      python tools/prove_div_total_mass_f64.py [--samples N]
"""
import os
import random
import re
import struct
import sys
from fractions import Fraction

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def f32(x):
    return struct.unpack("<f", struct.pack("<f", x))[0]


def rn64(q):
    """Correctly rounded (nearest, ties to even) binary64 value of a positive Fraction in the normal range, as a Fraction."""
    assert q > 0
    e = q.numerator.bit_length() - q.denominator.bit_length()       # 2^(e-1) <= q < 2^(e+1)
    if Fraction(2) ** e > q:
        e -= 1                                                      # now 2^e <= q < 2^(e+1)
    scale = Fraction(2) ** (52 - e)                                 # q * scale in [2^52, 2^53)
    v = q * scale
    n, rem = divmod(v.numerator, v.denominator)
    twice = 2 * rem
    if twice > v.denominator or (twice == v.denominator and (n & 1)):
        n += 1
    return Fraction(n) / scale


def rn64_signed(q):
    return -rn64(-q) if q < 0 else (Fraction(0) if q == 0 else rn64(q))


def fma_pair(x, zh, zl):
    """The kernel's two operations, simulated exactly: t = RN(x * zl); q = RN(x * zh + t)."""
    t = rn64_signed(x * zl)
    return rn64_signed(x * zh + t)


def hexf(fr):
    return float(fr).hex()


def hardest_cases(cn):
    """Dividend significands X in [2^52, 2^53) whose quotient sits at the MINIMAL distance from a rounding breakpoint:
    X * 2^j = +-1 (mod Cn) for j = 24 (upper binade of the quotient) and 25 (lower binade)."""
    out = []
    for j in (24, 25):
        inv = pow(pow(2, j, cn), -1, cn)
        for r in (1, cn - 1):
            x0 = (inv * r) % cn                                    # X = x0 (mod Cn)
            k0 = ((1 << 52) - x0 + cn - 1) // cn
            for k in (k0, k0 + 1, k0 + (1 << 28), ((1 << 53) - 1 - x0) // cn):   # a few representatives over the binade
                X = x0 + k * cn
                if (1 << 52) <= X < (1 << 53):
                    out.append(X)
    return out


def convergent_cases(c):
    """Best rational approximations p/q of C (continued-fraction convergents and their multiples in range): the dividends a
    Brisebarre-Muller-Raina enumeration inspects."""
    out, a, h0, h1, k0, k1, x = [], [], 0, 1, 1, 0, c
    for _ in range(40):
        ai = x.numerator // x.denominator
        h0, h1 = h1, ai * h1 + h0
        k0, k1 = k1, ai * k1 + k0
        for p in (h1, k1):
            if p:
                m = ((1 << 52) + p - 1) // p
                for mult in (m, m + 1, ((1 << 53) - 1) // p):
                    X = p * mult
                    if (1 << 52) <= X < (1 << 53):
                        out.append(X)
        frac = x - ai
        if frac == 0:
            break
        x = 1 / frac
    return out


def main():
    samples = 20000
    if "--samples" in sys.argv:
        samples = int(sys.argv[sys.argv.index("--samples") + 1])
    c32 = f32(f32(0.1) + 1.0)                                      # 0.1f + 1.0f folded in binary32 (CartPoleEnv.cs:27-29)
    C = Fraction(c32)
    cn = int(C * (1 << 23))
    assert Fraction(cn, 1 << 23) == C and cn % 2 == 1 and cn == 9227469, cn
    y = 1 / C
    ZH = rn64(y)
    ZL = rn64_signed(y - ZH)
    delta = y - ZH - ZL
    print(f"C  = {c32!r} = {cn} / 2^23 (odd numerator)   hex {c32.hex()}")
    print(f"ZH = RN(1/C)      = {hexf(ZH)}")
    print(f"ZL = RN(1/C - ZH) = {hexf(ZL)}")
    print(f"delta = 1/C - ZH - ZL = {float(delta):.3e}  (|delta| = 2^{float(abs(delta)).hex().split('p')[1]} roughly)")
    # step 3: minimal distance of a quotient from a breakpoint, in units where the quotient of X in [2^52, 2^53) is X / C
    d_min = Fraction(1, 4 * cn)                                    # lower binade (ulp 1/2): 1/(4 Cn); upper binade: 1/(2 Cn) — take the smaller
    # step 4: error of the fma pair before its final rounding
    x_max = Fraction((1 << 53) - 1)
    t_bound = x_max * abs(ZL)
    assert t_bound < Fraction(1, 2)                                # |X * ZL| < 2^-1: ulp(X * ZL) <= 2^-54
    err_max = x_max * abs(delta) + Fraction(1, 1 << 55)
    print(f"D_MIN   (distance of any quotient from any breakpoint) >= 1/(4 Cn) = {float(d_min):.3e}")
    print(f"ERR_MAX (|x*ZH + RN(x*ZL) - x/C| over the binade)      <= {float(err_max):.3e}")
    assert err_max < d_min, "the counting argument fails"
    print(f"ERR_MAX < D_MIN by a factor of {float(d_min / err_max):.3e}  ->  RN(fma pair) == RN(x / C) for every x.  QED")

    # the literals the product and the oracle's twin carry are exactly these constants
    ok = True
    for path in ("gym.net_amd/csrc/cartpole64.hpp", "oracle/classic_control_ref.c"):
        txt = open(os.path.join(ROOT, path)).read()
        m = re.search(r"ZL\s*=\s*(-?0x[0-9a-fA-F.]+p[-+]?\d+)", txt)
        if not m:
            print(f"!! {path}: no ZL literal found")
            ok = False
            continue
        lit = Fraction(float.fromhex(m.group(1)))
        same = lit == ZL
        print(f"{path}: ZL literal {m.group(1)} {'==' if same else '!='} RN(1/C - ZH)")
        ok = ok and same
    assert Fraction(1.0 / c32) == ZH, "the host compiler's 1.0 / C must be RN(1/C) (it is: IEEE division)"

    # verification on the hardest dividends, the convergent sweep and random dividends (both signs, several exponents)
    cases = hardest_cases(cn) + convergent_cases(C)
    rng = random.Random(0xC0FFEE)
    cases += [rng.randrange(1 << 52, 1 << 53) for _ in range(samples)]
    cases += [(1 << 52), (1 << 53) - 1, cn << 29, (cn << 29) - 1, (cn << 29) + 1]
    worst = Fraction(10)
    bad = 0
    for X in cases:
        for e in (0, -1074 + 200, 900, -60):
            x = Fraction(X) * Fraction(2) ** e
            for sgn in (1, -1):
                q = fma_pair(sgn * x, ZH, ZL)
                want = rn64_signed(sgn * x / C)
                if q != want:
                    bad += 1
        # distance of this quotient from its nearest breakpoint, in ulps of the quotient
        Q = Fraction(X) / C
        ulp = Fraction(1, 2) if Q < (1 << 52) else Fraction(1)
        half = ulp / 2
        r = (Q - half) % ulp                                       # breakpoints sit at half + k * ulp
        dist = min(r, ulp - r) / ulp
        worst = min(worst, dist)
    print(f"checked {len(cases)} dividend significands x 4 exponents x 2 signs exactly: {bad} mismatches; "
          f"closest quotient to a breakpoint: {float(worst):.3e} ulp (bound 1/(2 Cn) = {float(Fraction(1, 2 * cn)):.3e})")
    assert worst >= Fraction(1, 2 * cn) and bad == 0
    if not ok:
        sys.exit(1)
    print("PROVED")


if __name__ == "__main__":
    main()
