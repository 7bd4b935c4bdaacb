#!/usr/bin/env python3
"""Generates tests/golden/cartpole_f64_kernel.npz — bit patterns of the GYMNET_FLAG_F64 arithmetic, produced by the oracle's
float64 "kernel semantics" twin (oracle/classic_control_ref.c: ref_sincos_f64_kernel, ref_cartpole_step_f64_kernel,
ref_cartpole_reset_f64).  The parity tests compare the HIP kernel with the twin LIVE; this fixture pins the twin itself, so
that a change that moved kernel and twin together (a coefficient, the reduction constants, the reset construction) is caught by
committed numbers on both sides: tests/test_oracle.py (CPU) and tests/test_gpu_f64.py (HIP).

    python tests/golden/make_f64_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import capi as oracle  # noqa: E402


def build():
    oracle.build()
    rng = np.random.default_rng(20261004)
    # (1) the kernel's sin / cos on arguments in and out of the unreduced range, incl. exact multiples of pi/2 and zeros
    x = np.concatenate([rng.uniform(-0.785, 0.785, 300), rng.uniform(-0.21, 0.21, 100), rng.uniform(-12, 12, 200), rng.uniform(-8e5, 8e5, 100),
                        np.arange(-8, 9) * (np.pi / 2), np.array([0.0, -0.0, 1e-300, 0.785398163397448, -0.7853981633974484, 823549.0])])
    s, c = oracle.sincos_f64_kernel(x)
    # (2) the reset draw: seed, lane offsets above 2^32, several ticks
    resets = np.stack([oracle.cartpole_reset_f64(0x5EED, off, tick, 64) for off, tick in ((0, 0), (123_456_789_000, 7), (1 << 40, 2 ** 33 + 5))])
    # (3) a free-running trace of 16 lanes x 250 steps with the fused auto-reset (state, reward, done after every step)
    n, steps, seed, off = 16, 250, 0xF64, 1000
    st = oracle.cartpole_reset_f64(seed, off, 0, n)
    acts = rng.integers(0, 2, (steps, n)).astype(np.int32)
    states, dones = [], []
    for t in range(steps):
        st, r, d = oracle.cartpole_autoreset_step_f64(seed, off, 1 + t, st, acts[t])
        assert (r == 1.0).all()
        states.append(st.copy()); dones.append(d.copy())
    return dict(sincos_x=x, sincos_s=s, sincos_c=c, resets=resets, trace_seed=np.uint64(seed), trace_offset=np.int64(off),
                trace_actions=acts, trace_states=np.stack(states), trace_done=np.stack(dones))


if __name__ == "__main__":
    out = os.path.join(HERE, "cartpole_f64_kernel.npz")
    np.savez_compressed(out, **build())
    print("wrote", out, os.path.getsize(out), "bytes")
