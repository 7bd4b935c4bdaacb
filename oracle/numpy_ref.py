"""oracle/numpy_ref.py — second, independent (NumPy) restatement of the CartPole hot path.

TEST INFRASTRUCTURE ONLY: imported by tests/ (and tests/golden/make_golden.py), never by the
product package.  PARITY UNPINNED (see oracle/classic_control_ref.c header): the reference is C#
and cannot run here; independence comes from writing the algorithm twice from the C# text —
here vectorised in NumPy, there scalar in C — and requiring bit-identical float64 results.

Restates (paths relative to /root/reference):
  CartPoleEnv constants  src/Gym.Environments/Envs/Classic/CartPoleEnv.cs:24-36
  CartPoleEnv.Step       src/Gym.Environments/Envs/Classic/CartPoleEnv.cs:137-186
  CartPoleEnv.Reset      src/Gym.Environments/Envs/Classic/CartPoleEnv.cs:63-67
"""
import numpy as np

f32 = np.float32
f64 = np.float64

# CartPoleEnv.cs:24-36 — `const float`; sums/products of consts fold in float.
GRAVITY = f32(9.8)
MASSCART = f32(1.0)
MASSPOLE = f32(0.1)
TOTAL_MASS = f32(MASSPOLE + MASSCART)
LENGTH = f32(0.5)
POLEMASS_LENGTH = f32(MASSPOLE * LENGTH)
FORCE_MAG = f32(10.0)
TAU = f32(0.02)
THETA_THRESHOLD = f32(12 * 2 * np.pi / 360)      # (float)(12 * 2 * Math.PI / 360)
X_THRESHOLD = f32(2.4)

# CartPoleEnv.cs:46 — observation-space bound `high`
OBS_HIGH = np.array([X_THRESHOLD * f32(2), np.finfo(f32).max, THETA_THRESHOLD * f32(2), np.finfo(f32).max], dtype=f32)


def cartpole_step(state, action, sbd, dtype=f64):
    """Vectorised CartPoleEnv.Step (CartPoleEnv.cs:137-186).

    state: [4, n] array (x, x_dot, theta, theta_dot); action: [n] ints; sbd: [n] int32
    (steps_beyond_done, -1 after Reset).  dtype=f64 is the reference's arithmetic; dtype=f32 is the
    HIP kernel's.  Returns (new_state[4,n], reward f32[n], done bool[n], new_sbd int32[n]).
    """
    t = dtype
    state = np.asarray(state, dtype=t)
    x, x_dot, theta, theta_dot = state[0], state[1], state[2], state[3]
    action = np.asarray(action)
    sbd = np.asarray(sbd, dtype=np.int32)
    force = np.where(action == 1, FORCE_MAG, -FORCE_MAG).astype(t)                  # :146
    costheta = np.cos(theta)                                                        # :147
    sintheta = np.sin(theta)                                                        # :148
    pml, tm, g, ln, mp, tau = (t(POLEMASS_LENGTH), t(TOTAL_MASS), t(GRAVITY), t(LENGTH), t(MASSPOLE), t(TAU))
    four_thirds = t(4.0) / t(3.0)
    temp = (force + pml * theta_dot * theta_dot * sintheta) / tm                    # :149
    thetaacc = (g * sintheta - costheta * temp) / (ln * (four_thirds - mp * costheta * costheta / tm))  # :150
    xacc = temp - pml * thetaacc * costheta / tm                                    # :151
    nx = x + tau * x_dot                                                            # :154
    nx_dot = x_dot + tau * xacc                                                     # :155
    ntheta = theta + tau * theta_dot                                                # :156
    ntheta_dot = theta_dot + tau * thetaacc                                         # :157
    # :167 on the float64 sums of :154,156 for either dtype: the kernel (dtype=f32) keeps float32 state but derives the integer
    # done flag from the reference's own float64 comparison (envs.hpp CartPole::step)
    vx = x.astype(f64) + f64(TAU) * x_dot.astype(f64)
    vt = theta.astype(f64) + f64(TAU) * theta_dot.astype(f64)
    xt, tt = f64(X_THRESHOLD), f64(THETA_THRESHOLD)
    with np.errstate(invalid="ignore"):
        done = (vx < -xt) | (vx > xt) | (vt < -tt) | (vt > tt)                      # :167
    first = done & (sbd == -1)
    later = done & (sbd != -1)
    reward = np.where(later, f32(0.0), f32(1.0)).astype(f32)                        # :168-183
    new_sbd = np.where(first, 0, np.where(later, sbd + 1, sbd)).astype(np.int32)
    return np.stack([nx, nx_dot, ntheta, ntheta_dot]).astype(t), reward, done, new_sbd


_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = np.uint32(0x9E3779B9)
_W1 = np.uint32(0xBB67AE85)
_LO = np.uint64(0xFFFFFFFF)


def philox4x32_10(ctr, key):
    """Philox4x32-10 (Random123).  ctr: uint32[4, n], key: uint32[2, n] -> uint32[4, n]."""
    c = [np.asarray(ctr[i], dtype=np.uint32).copy() for i in range(4)]
    k0 = np.asarray(key[0], dtype=np.uint32).copy()
    k1 = np.asarray(key[1], dtype=np.uint32).copy()
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = _M0 * c[0].astype(np.uint64)
            p1 = _M1 * c[2].astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & _LO).astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & _LO).astype(np.uint32)
            c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
            k0 = k0 + _W0
            k1 = k1 + _W1
    return np.stack(c)


def reset_words(seed, lanes, tick):
    """Engine reset draw: counter=(lane_lo, lane_hi, tick_lo, tick_hi), key=(seed_lo, seed_hi)."""
    lanes = np.asarray(lanes, dtype=np.uint64)
    n = lanes.shape[0]
    tick = np.uint64(tick)
    seed = np.uint64(seed)
    ctr = np.stack([(lanes & _LO).astype(np.uint32), (lanes >> np.uint64(32)).astype(np.uint32),
                    np.full(n, tick & _LO, dtype=np.uint32), np.full(n, tick >> np.uint64(32), dtype=np.uint32)])
    key = np.stack([np.full(n, seed & _LO, dtype=np.uint32), np.full(n, seed >> np.uint64(32), dtype=np.uint32)])
    return philox4x32_10(ctr, key)


ACTION_STREAM = 0x9E3779B97F4A7C15
AUX_STREAM = 0xD6E8FEB86659FD93


def action_words(seed, lanes, tick):
    """Words A and B of the engine's action stream, version 2 (csrc/philox.hpp): global lane L takes word (L & 3) of the call
    with counter (L >> 2, tick), key seed ^ ACTION_STREAM (A: the ActionSpace.Sample() word) / seed ^ AUX_STREAM (B: the
    epsilon-greedy coin, the normal regime's second uniform)."""
    lanes = np.asarray(lanes, dtype=np.uint64)
    pick = (lanes & np.uint64(3)).astype(np.intp)
    idx = np.arange(lanes.shape[0])
    a = reset_words(int(seed) ^ ACTION_STREAM, lanes >> np.uint64(2), tick)[pick, idx]
    b = reset_words(int(seed) ^ AUX_STREAM, lanes >> np.uint64(2), tick)[pick, idx]
    return a, b


def discrete_sample(seed, lanes, tick, nvals, start=0):
    """Discrete.Sample() (Discrete.cs:17-28): start + hi32(word A * n)."""
    a, _ = action_words(seed, lanes, tick)
    return (np.int64(start) + ((a.astype(np.uint64) * np.uint64(nvals)) >> np.uint64(32)).astype(np.int64)).astype(np.int32)


def u01_24(words):
    return (words >> np.uint32(8)).astype(f32) * f32(1.0 / 16777216.0)


def cartpole_reset(seed, lanes, tick):
    """CartPoleEnv.Reset (CartPoleEnv.cs:63-67) with the engine's Philox stream, kernel (f32) semantics."""
    u = u01_24(reset_words(seed, lanes, tick))
    return (f32(-0.05) + f32(0.1) * u).astype(f32)
