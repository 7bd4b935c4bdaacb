"""tests/golden/make_golden.py — regenerates the committed golden vectors.

PROVENANCE: the reference (C#) cannot be executed in this image and its own tests pin no CartPole
numbers (tests/Gym.Tests/Envs/Classic/CartpoleEnvironment.cs:14-35 has no Assert), so these vectors
are produced by the CPU restatement in oracle/classic_control_ref.c ("parity unpinned").  They are
data: inputs + expected outputs.  Their job is (a) to freeze the restatement so it cannot drift
silently, and (b) to give the GPU box — where /root/reference does not exist — fixed cases.

Run:  python tests/golden/make_golden.py      (writes tests/golden/*.npz, deterministic)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import capi  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
f32 = np.float32


def cartpole_teacher_forced(rng):
    n_rand, n_wide = 3072, 1024
    s = np.zeros((4, n_rand + n_wide), dtype=f32)
    # in-range states: what a random-action rollout actually visits
    s[0, :n_rand] = rng.uniform(-2.4, 2.4, n_rand)
    s[1, :n_rand] = rng.uniform(-3.0, 3.0, n_rand)
    s[2, :n_rand] = rng.uniform(-0.21, 0.21, n_rand)
    s[3, :n_rand] = rng.uniform(-3.5, 3.5, n_rand)
    # large-angle / large-velocity states (step-after-done territory)
    s[0, n_rand:] = rng.uniform(-10, 10, n_wide)
    s[1, n_rand:] = rng.uniform(-20, 20, n_wide)
    s[2, n_rand:] = rng.uniform(-3.2, 3.2, n_wide)
    s[3, n_rand:] = rng.uniform(-20, 20, n_wide)
    a = rng.integers(0, 2, s.shape[1]).astype(np.int32)
    a[::97] = 2      # invalid actions: reference pushes LEFT for anything != 1 (CartPoleEnv.cs:146)
    a[5::101] = -1
    ns, rew, done, sbd = capi.cartpole_step(s.astype(np.float64), a)
    return dict(state=s, action=a, next_state=ns, reward=rew, done=done, sbd=sbd)


def cartpole_edges():
    c = capi.cartpole_constants()
    xt, tt = f32(c["x_threshold"]), f32(c["theta_threshold_radians"])
    rows = []
    # With x_dot = 0 (theta_dot = 0) explicit Euler leaves x (theta) EXACTLY unchanged, so the
    # done flag is decided by the input alone: strict inequalities => equal-to-threshold is NOT done.
    for sign in (+1, -1):
        for thr, idx in ((xt, 0), (tt, 2)):
            t = f32(sign) * thr
            below = np.nextafter(t, f32(0))
            above = np.nextafter(t, f32(sign) * f32(np.inf))
            for v in (below, t, above):
                for a in (0, 1):
                    for other in (0.0, 0.01, -0.01):
                        st = [0.0, 0.0, 0.0, 0.0]
                        st[idx] = float(v)
                        st[2 - idx] = other          # the other position-like component, well inside
                        rows.append((st, a))
    # upright at rest; symmetric pair; NaN and inf propagate, comparisons with NaN are false
    rows.append(([0.0, 0.0, 0.0, 0.0], 1))
    rows.append(([0.0, 0.0, 0.0, 0.0], 0))
    rows.append(([0.5, -0.3, 0.1, 0.7], 1))
    rows.append(([-0.5, 0.3, -0.1, -0.7], 0))
    rows.append(([np.nan, 0.0, 0.0, 0.0], 1))
    rows.append(([0.0, 0.0, np.inf, 0.0], 0))
    s = np.array([r[0] for r in rows], dtype=f32).T.copy()
    a = np.array([r[1] for r in rows], dtype=np.int32)
    with np.errstate(all="ignore"):
        ns, rew, done, sbd = capi.cartpole_step(s.astype(np.float64), a)
    return dict(state=s, action=a, next_state=ns, reward=rew, done=done, sbd=sbd)


def cartpole_steps_beyond_done():
    # one lane pushed right until it falls, then stepped 6 more times without Reset:
    # reward stream 1,...,1,1(done),0,0,... and sbd -1,...,-1,0,1,2,... (CartPoleEnv.cs:168-183)
    start = np.array([0.0, 0.0, 0.05, 0.0], dtype=f32)
    s = start.astype(np.float64).reshape(4, 1)
    sbd = np.array([-1], dtype=np.int32)
    states, rewards, dones, sbds = [], [], [], []
    extra = 0
    while extra < 6:
        s, r, d, sbd = capi.cartpole_step(s, np.array([1], dtype=np.int32), sbd)
        states.append(s[:, 0].copy()); rewards.append(r[0]); dones.append(d[0]); sbds.append(sbd[0])
        if d[0]:
            extra += 1
    return dict(start=start, states=np.array(states),
                reward=np.array(rewards, dtype=f32), done=np.array(dones, dtype=np.uint8),
                sbd=np.array(sbds, dtype=np.int32))


def cartpole_reference_test_trace(rng):
    # Shape of tests/Gym.Tests/Envs/Classic/CartpoleEnvironment.cs:19-27 and README.md:34-47:
    # 1000 iterations of  `if done: Reset() else Step(i % 2)`.  Reset draws come from a recorded
    # list (f32-representable), so the un-vendored NumSharp RNG is factored out.
    resets = rng.uniform(-0.05, 0.05, (64, 4)).astype(f32)
    done = True
    k = 0
    s = None
    sbd = None
    it_done, it_state, it_was_step, ep_len, cur = [], [], [], [], 0
    for i in range(1000):
        if done:
            s = resets[k].astype(np.float64).reshape(4, 1); k += 1
            sbd = np.array([-1], dtype=np.int32)
            done = False
            it_was_step.append(0)
            if cur:
                ep_len.append(cur)
            cur = 0
        else:
            s, r, d, sbd = capi.cartpole_step(s, np.array([i % 2], dtype=np.int32), sbd)
            done = bool(d[0]); cur += 1
            it_was_step.append(1)
        it_done.append(int(done)); it_state.append(s[:, 0].copy())
    return dict(resets=resets, resets_used=np.int32(k), it_done=np.array(it_done, dtype=np.uint8),
                it_was_step=np.array(it_was_step, dtype=np.uint8), it_state=np.array(it_state),
                episode_lengths=np.array(ep_len, dtype=np.int32))


def philox_and_resets():
    seeds = np.array([0, 1, 0x5EED, 0xDEADBEEFCAFEF00D], dtype=np.uint64)
    lane0 = np.array([0, 7, 1 << 20, (1 << 33) + 5], dtype=np.uint64)
    ticks = np.array([0, 1, 12345, (1 << 32) + 9], dtype=np.uint64)
    cp = np.stack([capi.cartpole_reset(int(s), int(l), int(t), 16) for s, l, t in zip(seeds, lane0, ticks)])
    pe = np.stack([capi.pendulum_reset(int(s), int(l), int(t), 16) for s, l, t in zip(seeds, lane0, ticks)])
    mc = np.stack([capi.mountaincar_reset(int(s), int(l), int(t), 16) for s, l, t in zip(seeds, lane0, ticks)])
    ac = np.stack([capi.acrobot_reset(int(s), int(l), int(t), 16) for s, l, t in zip(seeds, lane0, ticks)])
    ds = np.stack([capi.discrete_sample(int(s), int(l), int(t), 3, 0, 32) for s, l, t in zip(seeds, lane0, ticks)])
    bx = np.stack([capi.box_uniform_sample(int(s), int(l), int(t), -2.0, 2.0, 32) for s, l, t in zip(seeds, lane0, ticks)])
    return dict(seeds=seeds, lane0=lane0, ticks=ticks, cartpole=cp, pendulum=pe, mountaincar=mc, acrobot=ac,
                discrete3=ds, box_pm2=bx)


def other_envs(rng):
    n = 1024
    out = {}
    s = np.stack([rng.uniform(-4, 4, n), rng.uniform(-8, 8, n)]).astype(f32)
    a = rng.uniform(-2.5, 2.5, n).astype(f32)
    ns, obs, rew, _ = capi.pendulum_step(s.astype(np.float64), a.astype(np.float64))
    out.update(pe_state=s, pe_action=a, pe_next=ns, pe_obs=obs, pe_reward=rew)
    s = np.stack([rng.uniform(-1.2, 0.6, n), rng.uniform(-0.07, 0.07, n)]).astype(f32)
    s[0, :8] = f32(-1.2); s[1, :8] = f32(-0.05)        # left-wall inelastic stop
    s[0, 8:16] = f32(0.49); s[1, 8:16] = f32(0.06)     # goal crossing
    a = rng.integers(0, 3, n).astype(np.int32)
    ns, rew, done = capi.mountaincar_step(s.astype(np.float64), a)
    out.update(mc_state=s, mc_action=a, mc_next=ns, mc_reward=rew, mc_done=done)
    s = np.stack([rng.uniform(-3.1, 3.1, n), rng.uniform(-3.1, 3.1, n),
                  rng.uniform(-12, 12, n), rng.uniform(-28, 28, n)]).astype(f32)
    a = rng.integers(0, 3, n).astype(np.int32)
    ns, obs, rew, done = capi.acrobot_step(s.astype(np.float64), a)
    out.update(ac_state=s, ac_action=a, ac_next=ns, ac_obs=obs, ac_reward=rew, ac_done=done)
    return out


def main():
    """python make_golden.py [file stem ...] — all fixtures, or only the named ones (round 6 regenerated philox_resets alone, for
    action stream v2: the draws of discrete3 / box_pm2 moved, the reset draws did not)."""
    import sys
    capi.build()
    rng = np.random.default_rng(20261001)
    makers = [("cartpole_teacher_forced", lambda: cartpole_teacher_forced(rng)), ("cartpole_edges", cartpole_edges),
              ("cartpole_steps_beyond_done", cartpole_steps_beyond_done),
              ("cartpole_reference_test_trace", lambda: cartpole_reference_test_trace(rng)),
              ("philox_resets", philox_and_resets), ("other_envs", lambda: other_envs(rng))]
    only = set(sys.argv[1:])
    for stem, make in makers:
        data = make()                    # always evaluated: the makers share one generator, and its order fixes their inputs
        if not only or stem in only:
            np.savez_compressed(os.path.join(OUT, stem + ".npz"), **data)
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
