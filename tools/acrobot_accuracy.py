#!/usr/bin/env python3
"""Acrobot accuracy budget — CPU only, on the oracle's twins.

For one RK4 step of dt = 0.2 from states drawn uniformly from a box, max and 99.9th-percentile |error| against the float64
restatement of upstream's formulas (oracle: ref_acrobot_step_f64) of
  (a) a LITERAL float32 transcription of upstream's dsdt / rk4 (libm sinf / cosf, IEEE division, upstream's association order), and
  (b) the SHIPPED instruction-diet form (envs.hpp Acrobot::dsdt = oracle acrobot_dsdt_f32: one polynomial reciprocal per stage,
      re-associated numerators, explicit fma, in-house sin / cos) — bit-identical to the HIP kernel (tests/test_gpu_bench_kernels.py).
If (b) is no worse than (a), the float32-vs-float64 tolerance the GPU tests need is RK4 amplifying float32 rounding, not the
diet.  Angles are compared on the circle (they wrap at +-pi).  Writes profiles/acrobot_accuracy_r04.txt.

    python tools/acrobot_accuracy.py [--n 2000000]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import capi as oracle  # noqa: E402  (tools/ is measurement infrastructure, like tests/)

BOXES = [  # name, |th| bound, |dth1| bound, |dth2| bound
    ("reset box (upstream: U(-0.1, 0.1)^4)", 0.1, 0.1, 0.1),
    ("calm (|dth| < 2)", 3.1, 2.0, 2.0),
    ("bench / 2^20 parity test box", 3.1, 4.0, 9.0),
    ("test fixture box (velocity clamps 4 pi, 9 pi)", 3.1, 12.0, 28.0),
]


def errors(got, want):
    dang = np.abs(np.angle(np.exp(1j * (got[:2].astype(np.float64) - want[:2]))))
    dvel = np.abs(got[2:].astype(np.float64) - want[2:])
    return dang.max(axis=0), dvel.max(axis=0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=2_000_000)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "acrobot_accuracy_r04.txt"))
    args = ap.parse_args()
    oracle.build()
    rng = np.random.default_rng(2024)
    lines = [__doc__.split("\n\n")[0], "",
             f"{args.n} states per box, actions uniform in {{0, 1, 2}}; errors against the float64 restatement, one step.",
             "", f"{'state box':48s} {'form':22s} {'angle max':>11s} {'angle p99.9':>11s} {'vel max':>11s} {'vel p99.9':>11s} {'done flips':>10s}"]
    worst = {}
    for name, th, v1, v2 in BOXES:
        s0 = np.stack([rng.uniform(-th, th, args.n), rng.uniform(-th, th, args.n), rng.uniform(-v1, v1, args.n),
                       rng.uniform(-v2, v2, args.n)]).astype(np.float32)
        a = rng.integers(0, 3, args.n).astype(np.int32)
        want, _, _, d64 = oracle.acrobot_step(s0.astype(np.float64), a, dtype=np.float64)
        lit = oracle.acrobot_step_f32_literal(s0, a)
        diet = oracle.acrobot_step(s0, a, dtype=np.float32)
        for form, res in (("(a) literal float32", lit), ("(b) shipped diet form", diet)):
            da, dv = errors(res[0], want)
            flips = int((res[3] != d64).sum())
            lines.append(f"{name:48s} {form:22s} {da.max():11.3e} {np.quantile(da, 0.999):11.3e} {dv.max():11.3e} {np.quantile(dv, 0.999):11.3e} {flips:10d}")
            worst[(name, form[:3])] = (da.max(), dv.max())
    lines += ["", "Reading: up to the bench box the shipped form is at or below the literal transcription at the maximum AND at the 99.9th",
              "percentile.  In the most energetic box the MAXIMUM over 2e6 states is a tail statistic of an ill-conditioned map (one RK4",
              "step of dt = 0.2 with |dth2| up to 28 moves an angle by ~5 rad; d(ddth)/d(state) grows with dth^2) and the two float32",
              "forms land within +-30 % of each other there, either way round depending on the sample; at the 99.9th percentile the",
              "shipped form is again the better one.  So the 1e-3 velocity tolerance the energetic GPU tests used is float32 rounding",
              "amplified by RK4, not a cost of the instruction diet.  The GPU tests no longer carry a fixed loose number: they evaluate",
              "the literal transcription on their own inputs and require the kernel's error to be no worse at the median / 99th",
              "percentile (+25 % sampling slack) and within 4 x at the single worst lane",
              "(tests/test_gpu_other_envs.py::test_acrobot_teacher_forced, tests/test_gpu_bench_kernels.py Acrobot)."]
    text = "\n".join(lines) + "\n"
    open(args.out, "w").write(text)
    print(text)
    return worst


if __name__ == "__main__":
    main()
