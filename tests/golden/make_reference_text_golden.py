#!/usr/bin/env python3
"""Generates tests/golden/cartpole_reference_text.npz: CartPoleEnv.Step input -> output vectors obtained by EVALUATING THE
REFERENCE'S OWN SOURCE TEXT (oracle/evaluate_reference_text.py: an interpreter for the expression / statement subset that
method uses, run over /root/reference/src/Gym.Environments/Envs/Classic/CartPoleEnv.cs:24-36,137-186 where it lies).
Build container only — the GPU box has no reference tree; the .npz (numbers only: inputs and outputs) is committed and
travels.  Not an execution of the C# (there is no .NET here): what these vectors pin is that the oracle's restatement and the
reference's text denote the same arithmetic under C#'s numeric-promotion rules, which the interpreter implements.

    python tests/golden/make_reference_text_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle.evaluate_reference_text import ReferenceText  # noqa: E402


def inputs():
    rng = np.random.default_rng(20261002)
    n_in, n_wide, n_edge = 2400, 600, 200
    f32 = np.float32
    s_in = np.stack([rng.uniform(-2.4, 2.4, n_in), rng.uniform(-2.5, 2.5, n_in), rng.uniform(-0.2095, 0.2095, n_in), rng.uniform(-3, 3, n_in)])
    s_wide = np.stack([rng.uniform(-6, 6, n_wide), rng.uniform(-20, 20, n_wide), rng.uniform(-30, 30, n_wide), rng.uniform(-40, 40, n_wide)])
    # states whose NEXT x / theta lands within a few float32 ulps of a threshold (strict inequalities, CartPoleEnv.cs:167)
    xt, tt, tau = float(f32(2.4)), float(f32(0.20943951606750488)), float(f32(0.02))
    e = np.zeros((4, n_edge))
    k = np.arange(n_edge)
    sign = np.where(k % 2 == 0, 1.0, -1.0)
    ulps = (k // 4 % 5 - 2) * np.where(k % 4 < 2, np.spacing(f32(2.4)), np.spacing(f32(0.2094)))
    e[1] = rng.uniform(-1, 1, n_edge); e[3] = rng.uniform(-1, 1, n_edge)
    on_x = k % 4 < 2
    e[0] = np.where(on_x, sign * (xt + ulps) - tau * e[1], rng.uniform(-1, 1, n_edge))
    e[2] = np.where(~on_x, sign * (tt + ulps) - tau * e[3], rng.uniform(-0.1, 0.1, n_edge))
    # states are what a float32 engine can hold: binary32 values, widened to binary64 (the reference computes in double)
    state = np.concatenate([s_in, s_wide, e], axis=1).astype(f32).astype(np.float64)
    n = state.shape[1]
    action = rng.integers(0, 2, n).astype(np.int32)
    action[::17] = rng.integers(-3, 5, action[::17].shape)          # anything != 1 pushes left (CartPoleEnv.cs:139,146)
    sbd = np.full(n, -1, np.int32)
    sbd[::5] = 0
    sbd[1::11] = 3
    return state, action, sbd


def main():
    ref = ReferenceText()
    state, action, sbd = inputs()
    n = state.shape[1]
    nxt = np.zeros_like(state); reward = np.zeros(n, np.float32); done = np.zeros(n, np.uint8); sbd_out = np.zeros(n, np.int32)
    for i in range(n):
        s, r, d, b = ref.step(state[:, i], int(action[i]), int(sbd[i]))
        nxt[:, i] = s; reward[i] = r; done[i] = d; sbd_out[i] = b
    out = os.path.join(HERE, "cartpole_reference_text.npz")
    np.savez_compressed(out, state=state, action=action, sbd=sbd, next_state=nxt, reward=reward, done=done, sbd_out=sbd_out,
                        constants=np.array([float(ref.constants[k].v) for k in
                                            ("gravity", "masscart", "masspole", "total_mass", "length", "polemass_length", "force_mag", "tau",
                                             "theta_threshold_radians", "x_threshold")]))
    print(f"wrote {out}: {n} instances, {int(done.sum())} done, rewards {sorted(set(reward.tolist()))}")


if __name__ == "__main__":
    main()
