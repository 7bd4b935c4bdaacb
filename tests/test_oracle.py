"""CPU tests of the oracle itself: the restatement against its independent twin, hand-derived
closed forms, symmetry, the committed golden vectors and the published Philox known answers.

The reference pins nothing for CartPole (tests/Gym.Tests/Envs/Classic/CartpoleEnvironment.cs:14-35
asserts nothing) => parity unpinned; these tests are what stands in for the missing pins.
"""
import math
import os

import numpy as np
import pytest

from oracle import numpy_ref as nr

f32 = np.float32
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_constants_are_the_float32_values_of_the_csharp_consts(oracle):
    # SURVEY Appendix A hex values (CartPoleEnv.cs:24-36)
    c = oracle.cartpole_constants()
    assert c["gravity"] == float.fromhex("0x1.39999ap+3")
    assert c["masspole"] == float.fromhex("0x1.99999ap-4")
    assert c["total_mass"] == float.fromhex("0x1.19999ap+0")
    assert c["polemass_length"] == float.fromhex("0x1.99999ap-5")
    assert c["tau"] == float.fromhex("0x1.47ae14p-6")
    assert c["theta_threshold_radians"] == float.fromhex("0x1.aceeap-3")
    assert c["x_threshold"] == float.fromhex("0x1.333334p+1")
    assert c["length"] == 0.5 and c["force_mag"] == 10.0 and c["masscart"] == 1.0
    # numpy twin agrees
    assert float(nr.TOTAL_MASS) == c["total_mass"] and float(nr.POLEMASS_LENGTH) == c["polemass_length"]
    assert float(nr.THETA_THRESHOLD) == c["theta_threshold_radians"] and float(nr.TAU) == c["tau"]
    # observation-space bound, CartPoleEnv.cs:46
    assert nr.OBS_HIGH[0] == f32(4.8000002) and nr.OBS_HIGH[2] == f32(0.41887903)


def test_c_and_numpy_restatements_are_bit_identical_in_f64(oracle):
    rng = np.random.default_rng(1)
    n = 100_000
    s = np.stack([rng.uniform(-3, 3, n), rng.uniform(-4, 4, n), rng.uniform(-0.3, 0.3, n), rng.uniform(-4, 4, n)])
    s = s.astype(f32).astype(np.float64)
    a = rng.integers(0, 2, n).astype(np.int32)
    sbd = rng.integers(-1, 3, n).astype(np.int32)
    cs, cr, cd, cb = oracle.cartpole_step(s, a, sbd)
    ps, pr, pd, pb = nr.cartpole_step(s, a, sbd)
    assert np.array_equal(cs, ps)
    assert np.array_equal(cr, pr) and np.array_equal(cd, pd.astype(np.uint8)) and np.array_equal(cb, pb)


def test_hand_derived_upright_at_rest(oracle):
    # state 0, action 1: sin=0, cos=1 => temp = 10/M; thetaacc = -temp/(L*(4/3 - mp/M)); xacc = temp - pml*thetaacc/M
    c = oracle.cartpole_constants()
    M, L, mp, pml, tau = c["total_mass"], c["length"], c["masspole"], c["polemass_length"], c["tau"]
    temp = 10.0 / M
    thetaacc = (0.0 - 1.0 * temp) / (L * (4.0 / 3.0 - mp / M))
    xacc = temp - pml * thetaacc / M
    s, r, d, b = oracle.cartpole_step(np.zeros((4, 1)), np.array([1], dtype=np.int32))
    assert s[0, 0] == 0.0 and s[2, 0] == 0.0
    assert s[1, 0] == tau * xacc and s[3, 0] == tau * thetaacc
    assert r[0] == 1.0 and d[0] == 0 and b[0] == -1
    # magnitudes as physics says: cart accelerates right, pole tips left
    assert s[1, 0] == pytest.approx(0.19512195, abs=1e-7) and s[3, 0] == pytest.approx(-0.29268293, abs=1e-7)


def test_mirror_symmetry_is_exact(oracle):
    # step(-s, a=0) == -step(s, a=1): every operation is sign-symmetric in IEEE arithmetic
    rng = np.random.default_rng(2)
    n = 20_000
    s = np.stack([rng.uniform(-3, 3, n), rng.uniform(-4, 4, n), rng.uniform(-0.3, 0.3, n), rng.uniform(-4, 4, n)])
    a1 = np.ones(n, dtype=np.int32)
    p = oracle.cartpole_step(s, a1)
    m = oracle.cartpole_step(-s, 1 - a1)
    assert np.array_equal(p[0], -m[0]) and np.array_equal(p[2], m[2])


def test_invalid_action_pushes_left_like_release_build(oracle):
    # CartPoleEnv.cs:139 is a Debug.Assert (no-op in Release); :146 `iaction == 1 ? +F : -F`
    s = np.array([[0.1], [0.2], [0.05], [-0.3]])
    left = oracle.cartpole_step(s, np.array([0], dtype=np.int32))[0]
    for bad in (2, -1, 7):
        assert np.array_equal(oracle.cartpole_step(s, np.array([bad], dtype=np.int32))[0], left)
    assert oracle.lib().ref_discrete_contains(0, 2) == 1 and oracle.lib().ref_discrete_contains(1, 2) == 1
    assert oracle.lib().ref_discrete_contains(2, 2) == 0 and oracle.lib().ref_discrete_contains(-1, 2) == 0


def test_golden_teacher_forced(oracle, golden):
    g = golden("cartpole_teacher_forced")
    ns, rew, done, sbd = oracle.cartpole_step(g["state"].astype(np.float64), g["action"])
    assert np.array_equal(ns, g["next_state"])
    assert np.array_equal(rew, g["reward"]) and np.array_equal(done, g["done"]) and np.array_equal(sbd, g["sbd"])
    # numpy twin reproduces the fixture too
    ps = nr.cartpole_step(g["state"].astype(np.float64), g["action"], np.full(ns.shape[1], -1, np.int32))
    assert np.array_equal(ps[0], g["next_state"]) and np.array_equal(ps[2].astype(np.uint8), g["done"])
    assert 0 < int(g["done"].sum()) < g["done"].size


def test_golden_threshold_edges_strict_inequalities(oracle, golden):
    g = golden("cartpole_edges")
    with np.errstate(all="ignore"):
        ns, rew, done, sbd = oracle.cartpole_step(g["state"].astype(np.float64), g["action"])
    assert np.array_equal(ns, g["next_state"], equal_nan=True)
    assert np.array_equal(done, g["done"])
    c = oracle.cartpole_constants()
    xt, tt = f32(c["x_threshold"]), f32(c["theta_threshold_radians"])
    st = g["state"]
    for i in range(st.shape[1]):
        x, th = st[0, i], st[2, i]
        if st[1, i] == 0 and st[3, i] == 0 and np.isfinite(st[:, i]).all():
            # position-like components are unchanged by explicit Euler when velocities are 0
            expect = (x < -xt) or (x > xt) or (th < -tt) or (th > tt)
            assert bool(g["done"][i]) == bool(expect), (i, st[:, i])
    # exactly-at-threshold is NOT done (strict < and >, CartPoleEnv.cs:167)
    eq = (np.abs(st[0]) == xt) | (np.abs(st[2]) == tt)
    assert eq.sum() >= 8 and not g["done"][eq].any()
    # NaN state: all comparisons false => not done
    nan_col = np.isnan(st[0])
    assert nan_col.sum() == 1 and g["done"][nan_col][0] == 0


def test_golden_steps_beyond_done_sequence(oracle, golden):
    g = golden("cartpole_steps_beyond_done")
    s = g["start"].astype(np.float64).reshape(4, 1)
    sbd = np.array([-1], dtype=np.int32)
    for t in range(g["reward"].shape[0]):
        s, r, d, sbd = oracle.cartpole_step(s, np.array([1], dtype=np.int32), sbd)
        assert np.array_equal(s[:, 0], g["states"][t])
        assert r[0] == g["reward"][t] and d[0] == g["done"][t] and sbd[0] == g["sbd"][t]
    first = int(np.argmax(g["done"]))
    assert np.all(g["reward"][: first + 1] == 1.0)          # incl. the step on which the pole fell
    assert np.all(g["reward"][first + 1:] == 0.0)           # every later step
    assert list(g["sbd"][first:]) == list(range(0, len(g["sbd"]) - first))


def test_golden_reference_test_shaped_trace(oracle, golden):
    # 1000 x (Reset-if-done else Step(i % 2)): CartpoleEnvironment.cs:19-27 / README.md:34-47
    g = golden("cartpole_reference_test_trace")
    done, k, s, sbd, lens, cur = True, 0, None, None, [], 0
    for i in range(1000):
        if done:
            s = g["resets"][k].astype(np.float64).reshape(4, 1); k += 1
            sbd = np.array([-1], dtype=np.int32); done = False
            if cur:
                lens.append(cur)
            cur = 0
            assert g["it_was_step"][i] == 0
        else:
            s, r, d, sbd = oracle.cartpole_step(s, np.array([i % 2], dtype=np.int32), sbd)
            done = bool(d[0]); cur += 1
            assert r[0] == 1.0          # reset-on-done callers only ever see reward 1
        assert int(done) == g["it_done"][i]
        assert np.array_equal(s[:, 0], g["it_state"][i])
    assert k == int(g["resets_used"]) and lens == list(g["episode_lengths"])
    assert 20 <= np.mean(lens) <= 60   # alternating-action episodes average ~37 steps


def test_f32_kernel_semantics_stay_within_1e5_of_reference_semantics(oracle, golden):
    # north_star tolerance: 1e-5 abs on float32 state, teacher-forced single step
    g = golden("cartpole_teacher_forced")
    inr = slice(0, 3072)                      # the in-range block of the fixture
    s32 = g["state"][:, inr]
    ns32, r32, d32, _ = oracle.cartpole_step(s32, g["action"][inr], dtype=np.float32)
    err = np.abs(ns32.astype(np.float64) - g["next_state"][:, inr])
    assert err.max() <= 1e-5 and err.max() < 2e-6
    assert np.array_equal(d32, g["done"][inr]) and np.array_equal(r32, g["reward"][inr])
    # numpy f32 twin (different sinf/cosf implementation) agrees to a few ulp
    p32 = nr.cartpole_step(s32, g["action"][inr], np.full(s32.shape[1], -1, np.int32), dtype=f32)[0]
    assert np.abs(p32.astype(np.float64) - ns32.astype(np.float64)).max() < 2e-6


def test_f32_free_run_drift_table(oracle):
    # regression guard on the tolerance claim: free-running f32 leaves 1e-5 only
    # after tens of steps; the first 20 steps stay below 1e-5
    n = 4096
    s64 = oracle.cartpole_reset(0x5EED, 0, 0, n).astype(np.float64)
    s32 = s64.astype(f32)
    rng = np.random.default_rng(3)
    alive = np.ones(n, dtype=bool)
    worst20 = 0.0
    for t in range(20):
        a = rng.integers(0, 2, n).astype(np.int32)
        s64, _, d64, _ = oracle.cartpole_step(s64, a)
        s32, _, d32, _ = oracle.cartpole_step(s32, a, dtype=np.float32)
        alive &= (d64 == 0) & (d32 == 0)
        worst20 = max(worst20, np.abs(s32.astype(np.float64) - s64)[:, alive].max())
    assert worst20 < 1e-5


# ------------------------------------------------------------------------------------------------
# Philox4x32-10: PINNED by the published Random123 known-answer vectors (kat_vectors, philox4x32 10)
# ------------------------------------------------------------------------------------------------
KAT = [
    ([0, 0, 0, 0], [0, 0], [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]),
    ([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2, [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]),
    ([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0],
     [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]),
]


@pytest.mark.parametrize("ctr,key,expect", KAT)
def test_philox_known_answers(oracle, ctr, key, expect):
    assert list(oracle.philox4x32_10(ctr, key)) == expect
    tw = nr.philox4x32_10(np.array(ctr, dtype=np.uint32).reshape(4, 1), np.array(key, dtype=np.uint32).reshape(2, 1))
    assert list(tw[:, 0]) == expect


def test_reset_distribution_and_golden(oracle, golden):
    g = golden("philox_resets")
    for k in range(len(g["seeds"])):
        seed, lane0, tick = int(g["seeds"][k]), int(g["lane0"][k]), int(g["ticks"][k])
        got = oracle.cartpole_reset(seed, lane0, tick, 16)
        assert np.array_equal(got, g["cartpole"][k])
        lanes = np.arange(lane0, lane0 + 16, dtype=np.uint64)
        assert np.array_equal(nr.cartpole_reset(seed, lanes, tick), got)
        assert np.array_equal(oracle.pendulum_reset(seed, lane0, tick, 16), g["pendulum"][k])
        assert np.array_equal(oracle.mountaincar_reset(seed, lane0, tick, 16), g["mountaincar"][k])
        assert np.array_equal(oracle.acrobot_reset(seed, lane0, tick, 16), g["acrobot"][k])
        assert np.array_equal(oracle.discrete_sample(seed, lane0, tick, 3, 0, 32), g["discrete3"][k])
        assert np.array_equal(oracle.box_uniform_sample(seed, lane0, tick, -2.0, 2.0, 32), g["box_pm2"][k])
        # action stream v2 (csrc/philox.hpp): lane L takes word (L & 3) of the call with counter (L >> 2, tick) — the C oracle, lane
        # by lane, against the NumPy Philox twin, on lane offsets that start inside a group (7, 2^33 + 5) and on aligned ones
        lanes32 = np.arange(lane0, lane0 + 32, dtype=np.uint64)
        wa, wb = oracle.action_words(seed, lane0, tick, 32)
        na, nb = nr.action_words(seed, lanes32, tick)
        assert np.array_equal(wa, na) and np.array_equal(wb, nb) and not np.array_equal(wa, wb)
        assert np.array_equal(nr.discrete_sample(seed, lanes32, tick, 3), g["discrete3"][k])
        assert np.array_equal((f32(-2.0) + f32(4.0) * nr.u01_24(na)).astype(f32), g["box_pm2"][k])
        # four lanes of a group share ONE call: its four words, in lane order
        first = int(-lane0 % 4)                                                   # first lane of the batch that starts a group
        call = nr.reset_words(seed ^ nr.ACTION_STREAM, np.array([(lane0 + first) >> 2], dtype=np.uint64), tick)[:, 0]
        assert np.array_equal(wa[first:first + 4], call)
        # epsilon-greedy: explores iff u01_24(word B) <= epsilon, and then takes exactly the Discrete.Sample() draw
        pol = np.full(32, 7, np.int32)
        comp = oracle.compose_discrete(seed, lane0, tick, 3, 0.4, pol)
        explore = nr.u01_24(nb) <= f32(0.4)
        assert np.array_equal(comp, np.where(explore, g["discrete3"][k], pol))
    # CartPoleEnv.Reset (CartPoleEnv.cs:63-67): 4 iid U(-0.05, 0.05) components
    big = oracle.cartpole_reset(7, 0, 3, 200_000)
    assert big.min() >= -0.05 and big.max() < 0.05
    assert abs(big.mean()) < 2e-4 and abs(big.std() - 0.1 / math.sqrt(12)) < 2e-4
    assert np.abs(np.corrcoef(big)[np.triu_indices(4, 1)]).max() < 0.01
    # sharding invariance by construction: lane offset == global lane id
    assert np.array_equal(oracle.cartpole_reset(7, 1000, 3, 64), big[:, 1000:1064])


def test_other_envs_golden(oracle, golden):
    # Pendulum / MountainCar / Acrobot are absent from the reference (README.md:69-76); these vectors
    # only freeze our restatement of the upstream algorithms.
    g = golden("other_envs")
    ns, obs, rew, _ = oracle.pendulum_step(g["pe_state"].astype(np.float64), g["pe_action"].astype(np.float64))
    assert np.array_equal(ns, g["pe_next"]) and np.array_equal(obs, g["pe_obs"]) and np.array_equal(rew, g["pe_reward"])
    assert np.abs(ns[1]).max() <= 8.0 and rew.max() <= 0.0
    ns, rew, done = oracle.mountaincar_step(g["mc_state"].astype(np.float64), g["mc_action"])
    assert np.array_equal(ns, g["mc_next"]) and np.array_equal(done, g["mc_done"])
    assert np.all(ns[1, :8] == 0.0) and np.all(ns[0, :8] == -1.2)     # inelastic left wall
    assert np.all(done[8:16] == 1)                                     # goal reached
    ns, obs, rew, done = oracle.acrobot_step(g["ac_state"].astype(np.float64), g["ac_action"])
    assert np.array_equal(ns, g["ac_next"]) and np.array_equal(obs, g["ac_obs"]) and np.array_equal(done, g["ac_done"])
    assert np.abs(ns[0]).max() <= math.pi and np.abs(ns[2]).max() <= 4 * math.pi and np.abs(ns[3]).max() <= 9 * math.pi
    assert np.array_equal(rew, np.where(done == 1, 0.0, -1.0))
    # f32 kernel semantics track f64 within the bars the GPU tests use
    s32 = oracle.pendulum_step(g["pe_state"], g["pe_action"], dtype=np.float32)
    assert np.abs(s32[0].astype(np.float64) - g["pe_next"]).max() < 1e-5
    s32 = oracle.mountaincar_step(g["mc_state"], g["mc_action"], dtype=np.float32)
    assert np.abs(s32[0].astype(np.float64) - g["mc_next"]).max() < 1e-6


def test_cpu_baseline_runs_and_counts(oracle):
    r = oracle.cpu_baseline(4096, 16, 2, alloc_faithful=True)
    assert r["env_steps"] == 4096 * 16 and r["seconds"] > 0 and r["dones"] >= 0
    r2 = oracle.cpu_baseline(4096, 16, 2, alloc_faithful=False)
    assert r2["checksum"] == r["checksum"]       # same arithmetic with and without the allocations


# ------------------------------------------------------------------------------------------------
# The kernel's two "cheap arithmetic" substitutions, proven / bounded on the CPU where the same IEEE
# operations (mul, fma, rint) give the same bits as on the GPU.
# ------------------------------------------------------------------------------------------------
def test_kernel_constant_division_equals_ieee_division_exhaustively(oracle):
    # fma(x, RN(1/C), x*RN(1/C - RN(1/C))) == x / C for C = total_mass: every one of the 2^23 significands,
    # EVERY finite float with biased exponent 23..254 (|x| from 2^-104 up to FLT_MAX), both signs: 3.9e9 cases
    assert oracle.check_div_total_mass(list(range(23, 255))) == 0
    # ...and it is NOT exact below that (the x*zl term goes subnormal): documented limit of the claim
    assert oracle.check_div_total_mass([22]) > 0 and oracle.check_div_total_mass([1]) > 0
    for v in (0.0, -0.0, np.inf, -np.inf):
        got = oracle.lib().ref_div_total_mass_kernel(v)
        assert got == v and np.signbit(got) == np.signbit(v)
    assert np.isnan(oracle.lib().ref_div_total_mass_kernel(float("nan")))


def test_kernel_sincos_accuracy(oracle):
    rng = np.random.default_rng(4)
    for lo, hi, abs_bar in ((0.0, 0.3, 7e-8), (0.0, 10.0, 1e-7), (10.0, 65536.0, 1e-7)):
        x = np.concatenate([rng.uniform(lo, hi, 40_000), -rng.uniform(lo, hi, 40_000)]).astype(f32)
        s, c = oracle.sincos_kernel(x)
        xs = x.astype(np.float64)
        assert np.abs(s - np.sin(xs)).max() <= abs_bar and np.abs(c - np.cos(xs)).max() <= abs_bar
    s, c = oracle.sincos_kernel(np.array([0.0, -0.0], dtype=f32))
    assert s[0] == 0.0 and c[0] == 1.0 and s[1] == 0.0 and c[1] == 1.0     # sin(-0) = +0 (documented)
    x = rng.uniform(-4, 4, 20_000).astype(f32)                     # odd / even symmetry is exact
    s1, c1 = oracle.sincos_kernel(x); s2, c2 = oracle.sincos_kernel(-x)
    assert np.array_equal(s1, -s2) and np.array_equal(c1, c2)
    s, c = oracle.sincos_kernel(np.array([np.nan, np.inf], dtype=f32))
    assert np.isnan(s).all() and np.isnan(c).all()


def test_small_argument_sincos_of_the_acrobot_kernel(oracle):
    """envs.hpp sincos_small (two-constant Cody-Waite, Horner cosine; Acrobot's RK4 stage angles, |x| < 12): accuracy against
    float64 over its whole intended range and beyond, exact odd / even symmetry, and the reduction's premise — n * C1 is
    exact for every |n| <= 15."""
    rng = np.random.default_rng(44)
    x = np.concatenate([rng.uniform(-16, 16, 60_000), rng.uniform(-3.2, 3.2, 60_000), np.linspace(-24, 24, 4001)]).astype(f32)
    s, c = oracle.sincos_kernel(x, small=True)
    xs = x.astype(np.float64)
    assert np.abs(s - np.sin(xs)).max() <= 1.2e-7 and np.abs(c - np.cos(xs)).max() <= 1.2e-7
    s2, c2 = oracle.sincos_kernel(-x, small=True)
    assert np.array_equal(s, -s2) and np.array_equal(c, c2)
    c1 = np.float32(float.fromhex("0x1.921fap+0"))
    for n in range(-15, 16):
        assert float(np.float32(n) * c1) == n * float(c1)                       # exact product: 20 + 4 bits
    rest = np.pi / 2 - float(c1)
    assert abs(float(np.float32(float.fromhex("0x1.54442ep-20"))) - rest) <= 2.0 ** -44          # half an ulp of C2


def test_restatement_against_exact_rational_arithmetic(oracle):
    """Third, evaluation-order-independent anchor: the CartPoleEnv.Step formulas (CartPoleEnv.cs:146-157) evaluated
    in EXACT rational arithmetic (sin/cos by Taylor series on Fractions, 40 terms) with the float32-valued
    constants.  The float64 restatement must sit within a few float64 ulps of the exact real-number result."""
    from fractions import Fraction as F

    def sincos(x):
        s = c = F(0)
        term_s, term_c = x, F(1)
        for k in range(40):
            s += term_s; c += term_c
            term_s = -term_s * x * x / ((2 * k + 2) * (2 * k + 3))
            term_c = -term_c * x * x / ((2 * k + 1) * (2 * k + 2))
        return s, c

    cst = oracle.cartpole_constants()
    g, mp, M, L, pml, tau = (F(cst[k]) for k in ("gravity", "masspole", "total_mass", "length", "polemass_length", "tau"))
    rng = np.random.default_rng(17)
    for _ in range(12):
        st = np.array([rng.uniform(-2, 2), rng.uniform(-2, 2), rng.uniform(-0.25, 0.25), rng.uniform(-3, 3)]).astype(f32)
        a = int(rng.integers(0, 2))
        x, xd, th, thd = (F(float(v)) for v in st)
        force = F(10) if a == 1 else F(-10)
        s, c = sincos(th)
        temp = (force + pml * thd * thd * s) / M
        thetaacc = (g * s - c * temp) / (L * (F(4, 3) - mp * c * c / M))
        xacc = temp - pml * thetaacc * c / M
        exact = [x + tau * xd, xd + tau * xacc, th + tau * thd, thd + tau * thetaacc]
        got = oracle.cartpole_step(st.astype(np.float64).reshape(4, 1), np.array([a], dtype=np.int32))[0][:, 0]
        for e, v in zip(exact, got):
            # 4.0/3.0 as a double literal and ~15 roundings: allow 64 ulp of slack around the exact value
            assert abs(float(e) - v) <= 64 * np.spacing(abs(v)) + 1e-300


def test_reset_stream_is_uniform_across_lanes_and_across_ticks(oracle):
    """Philox(key = seed, counter = (lane, tick)): the reset draws must look iid uniform whether one walks along the lanes
    at a fixed tick or along the ticks at a fixed lane (the two directions a rollout consumes them in)."""
    from scipy import stats
    seed, bins = 0xC0FFEE, 32
    lanes = oracle.cartpole_reset(seed, 12345, 9, 200_000)                      # [4, n] at one tick
    ticks = np.concatenate([oracle.cartpole_reset(seed, 777, t, 1) for t in range(20_000)], axis=1)   # one lane, many ticks
    for sample in (lanes, ticks):
        for comp in sample:
            u = (comp.astype(np.float64) + 0.05) / 0.1
            counts = np.histogram(u, bins=bins, range=(0.0, 1.0))[0]
            p = stats.chisquare(counts).pvalue
            assert p > 1e-4, p
        # neighbours (adjacent lanes / consecutive ticks) are uncorrelated, and so are the four components of one draw
        x = sample[0].astype(np.float64)
        assert abs(np.corrcoef(x[:-1], x[1:])[0, 1]) < 4.0 / np.sqrt(x.size)
        assert np.abs(np.corrcoef(sample.astype(np.float64))[np.triu_indices(4, 1)]).max() < 4.0 / np.sqrt(x.size)


def test_restatement_token_stream_equals_the_reference_text():
    """oracle/check_against_reference.py: the C restatement's Step assignments, reward machine, Reset and constants against
    the TEXT of CartPoleEnv.cs under /root/reference (build container only; the GPU box has no reference tree).  It does not
    lift "parity unpinned" (nothing runs the C#), it removes transcription slips — and a seeded slip must be caught."""
    import importlib.util
    ref = "/root/reference/src/Gym.Environments/Envs/Classic/CartPoleEnv.cs"
    if not os.path.exists(ref):
        pytest.skip("reference tree not present (GPU box)")
    spec = importlib.util.spec_from_file_location("check_against_reference", os.path.join(ROOT, "oracle", "check_against_reference.py"))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    assert chk.main() == 0
    c = open(os.path.join(ROOT, "oracle", "classic_control_ref.c")).read()
    want, _, integ = chk.reference_step(open(ref, encoding="utf-8-sig").read())
    assert integ == "euler"
    for a, b, var in (("costheta * temp)", "temp * costheta)", "thetaacc"), ("(double)CP_TAU * x_dot;", "(double)CP_TAU * xacc;", "x"),
                      ("4.0 / 3.0", "(4.0 / 3.0)", "thetaacc"), ("x > (double)CP_X_THRESHOLD", "x >= (double)CP_X_THRESHOLD", "done")):
        assert a in c
        got, _ = chk.restatement_step(c.replace(a, b))
        assert [v for v in chk.STEP_VARS if want[v] != got[v]] == [var], (a, b)


def test_reference_constant_initialisers_are_parsed_not_evaluated():
    """the text under /root/reference is untrusted, so the `(float)(...)` initialisers are walked as an AST of
    literals, Math.PI, + - * / and parentheses — nothing else is accepted, in particular nothing eval() would run."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_against_reference", os.path.join(ROOT, "oracle", "check_against_reference.py"))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    assert chk._arith("(12 * 2 * Math.PI / 360)") == 12 * 2 * np.pi / 360
    assert chk._arith("-(1 + 2) * 0.5") == -1.5
    got = chk.reference_constants("private const float a = 0.1f; private const float b = a * 0.5f; private const float c = (float) (12 * 2 * Math.PI / 360);")
    assert got["b"] == np.float32(0.1) * np.float32(0.5) and got["c"] == np.float32(12 * 2 * np.pi / 360)
    for hostile in ("().__class__.__base__.__subclasses__()", "__import__('os').system('true')", "[1][0]", "(lambda: 1)()", "x + 1",
                    "Math.E", "1 if 1 else 2", "2 ** 3"):
        with pytest.raises((ValueError, SyntaxError)):
            chk._arith(hostile)
    assert "eval(" not in open(os.path.join(ROOT, "oracle", "check_against_reference.py")).read().replace("to eval()", "")


def test_pendulum_fast_fmod_is_exact(oracle):
    """envs.hpp Pendulum::fmod_2pi (one multiply, trunc, two fma and a repair step instead of OCML's iterative fmodf) must be
    EXACT, because the float32 restatement keeps calling libm's fmodf: every 97th binary32 pattern below 2^22 * 2 pi in both
    signs (2.6e7 arguments) and the 7 neighbours of each of the first 400 000 multiples of 2 pi, bit for bit; infinities, NaN
    and huge arguments take the fallback."""
    assert oracle.check_fmod_2pi(97, 400_000) == 0
    L = oracle.lib()
    for x in (0.0, -0.0, 6.2831855, -6.2831855, 1e30, -1e30, float("inf")):
        want = np.fmod(np.float32(x), np.float32(2 * np.float32(np.pi))) if np.isfinite(x) else np.float32("nan")
        got = np.float32(L.ref_fmod_2pi_kernel(x))
        assert (np.isnan(want) and np.isnan(got)) or (want == got and np.signbit(want) == np.signbit(got)), x


def test_oracle_equals_vectors_evaluated_from_the_reference_text(oracle, golden):
    """tests/golden/cartpole_reference_text.npz holds 3200 CartPoleEnv.Step input -> output vectors obtained by evaluating the
    REFERENCE'S OWN SOURCE TEXT statement by statement (oracle/evaluate_reference_text.py: an interpreter for the subset of C#
    that method uses, with C#'s numeric promotion — not a .NET runtime; generated in the build container by
    tests/golden/make_reference_text_golden.py).  The float64 restatement must reproduce them BIT FOR BIT: states, the float
    reward, the done flag, steps_beyond_done — in-range states, wide states, +-2 float32 ulps around both thresholds, actions
    other than 0 / 1, and all three phases of the reward machine (sbd = -1, 0, > 0)."""
    g = golden("cartpole_reference_text")
    s, r, d, b = oracle.cartpole_step(g["state"], g["action"], g["sbd"], dtype=np.float64)
    assert np.array_equal(s, g["next_state"])
    assert np.array_equal(r, g["reward"]) and np.array_equal(d, g["done"]) and np.array_equal(b, g["sbd_out"])
    assert np.array_equal(np.array(list(oracle.cartpole_constants().values())), g["constants"])
    assert 500 < int(g["done"].sum()) < 2000 and set(g["reward"].tolist()) == {0.0, 1.0} and (g["sbd_out"] > 1).any()
    # the independent NumPy writing agrees with the same vectors too
    from oracle import numpy_ref
    s2, r2, d2, b2 = numpy_ref.cartpole_step(g["state"], g["action"], g["sbd"])
    assert np.array_equal(s2, g["next_state"]) and np.array_equal(r2, g["reward"]) and np.array_equal(d2, g["done"].astype(bool))
    assert np.array_equal(b2, g["sbd_out"])


def test_kernel_semantics_give_the_reference_integer_outputs_on_every_vector(oracle, golden):
    """Row a3: float32 state, but done / reward / steps_beyond_done exactly the reference's for EVERY float32
    input — the kernel semantics derive the flag from the float64 sums x + tau*x_dot, theta + tau*theta_dot the reference
    compares (CartPoleEnv.cs:154,156,167), not from their float32 roundings.  All 3200 reference-text vectors (200 of them within
    +-2 float32 ulps of a threshold) and the 78 constructed edge cases, for both twins (C and NumPy); a float32 comparison
    would flip 22 of them."""
    from oracle import numpy_ref
    g = golden("cartpole_reference_text")
    assert np.array_equal(g["state"], g["state"].astype(np.float32).astype(np.float64))     # the inputs ARE float32 values
    s, r, d, b = oracle.cartpole_step(g["state"].astype(np.float32), g["action"], g["sbd"], dtype=np.float32)
    assert np.array_equal(d, g["done"]) and np.array_equal(r, g["reward"]) and np.array_equal(b, g["sbd_out"])
    s2, r2, d2, b2 = numpy_ref.cartpole_step(g["state"].astype(np.float32), g["action"], g["sbd"], dtype=np.float32)
    assert np.array_equal(d2, g["done"].astype(bool)) and np.array_equal(r2, g["reward"]) and np.array_equal(b2, g["sbd_out"])
    # what a float32 comparison of the float32 next state would have answered: differs on the +-2-ulp block (the bug this guards)
    xt, tt = np.float32(2.4), np.float32(0.20943951606750488)
    with np.errstate(invalid="ignore"):
        naive = (s[0] < -xt) | (s[0] > xt) | (s[2] < -tt) | (s[2] > tt)
    assert 1 <= int((naive != g["done"].astype(bool)).sum()) <= 64
    e = golden("cartpole_edges")
    _, _, de, _ = oracle.cartpole_step(e["state"], e["action"], dtype=np.float32)
    assert np.array_equal(de, e["done"])


def test_reference_text_vectors_are_reproducible_and_the_interpreter_is_strict():
    """Build container only (needs /root/reference): regenerating the vectors from the reference text reproduces the committed
    fixture exactly, and the interpreter refuses anything outside its grammar instead of guessing (the text is untrusted: it is
    tokenised and parsed, never eval()'d)."""
    import importlib.util
    if not os.path.exists("/root/reference/src/Gym.Environments/Envs/Classic/CartPoleEnv.cs"):
        pytest.skip("reference tree not present (GPU box)")
    spec = importlib.util.spec_from_file_location("make_reference_text_golden", os.path.join(ROOT, "tests", "golden", "make_reference_text_golden.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    from oracle.evaluate_reference_text import Parser, ReferenceText, Value, _tokens
    ref = ReferenceText()
    state, action, sbd = mk.inputs()
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "cartpole_reference_text.npz")))
    assert np.array_equal(state, g["state"]) and np.array_equal(action, g["action"]) and np.array_equal(sbd, g["sbd"])
    for i in list(range(0, state.shape[1], 37)) + list(range(3000, 3200)):
        s, r, d, b = ref.step(state[:, i], int(action[i]), int(sbd[i]))
        assert np.array_equal(s, g["next_state"][:, i]) and r == g["reward"][i] and d == bool(g["done"][i]) and b == g["sbd_out"][i]
    assert ref.reads == ["x", "x_dot", "theta", "theta_dot"] == ref.writes and ref.integrator_literal == "euler"
    # C#'s numeric promotion as the interpreter implements it
    env = {"f": Value(np.float32(0.1), "float"), "d": Value(0.1, "double")}
    assert Parser(_tokens("f * f"), env).expr().t == "float" and Parser(_tokens("f * d"), env).expr().t == "double"
    assert Parser(_tokens("(float) (12 * 2 * Math.PI / 360)"), env).expr().v == np.float32(12 * 2 * np.pi / 360)
    assert Parser(_tokens("7 / 2"), env).expr().v == 3
    for hostile in ("__import__", "f . g", "System.IO.File", "f [ 0 ]", "new Foo ( )"):
        with pytest.raises(ValueError):
            p = Parser(_tokens(hostile), env)
            p.expr()
            if p.peek() is not None:
                raise ValueError("trailing tokens")
    assert "eval(" not in open(os.path.join(ROOT, "oracle", "evaluate_reference_text.py")).read().replace("handed to eval()", "")


def test_restatement_fixtures_agree_with_the_reference_text(golden):
    """Build container only: the fixtures that were generated BY THE RESTATEMENT (teacher-forced tuples, threshold edges, the
    steps_beyond_done stream and the reference-test-shaped 1000-iteration trace of CartpoleEnvironment.cs:19-27) re-derived with the
    interpreter over the reference's own text (oracle/evaluate_reference_text.py) — bit for bit, including the free-running
    trace's 1000 states and its episode lengths."""
    if not os.path.exists("/root/reference/src/Gym.Environments/Envs/Classic/CartPoleEnv.cs"):
        pytest.skip("reference tree not present (GPU box)")
    from oracle.evaluate_reference_text import ReferenceText
    ref = ReferenceText()
    g = golden("cartpole_teacher_forced")
    for i in range(0, g["state"].shape[1], 7):
        s, r, d, _ = ref.step(g["state"][:, i].astype(np.float64), int(g["action"][i]), -1)
        assert np.array_equal(s, g["next_state"][:, i]) and r == g["reward"][i] and d == bool(g["done"][i])
    g = golden("cartpole_edges")
    for i in range(g["state"].shape[1]):
        s, r, d, _ = ref.step(g["state"][:, i].astype(np.float64), int(g["action"][i]), -1)
        assert np.array_equal(s, g["next_state"][:, i], equal_nan=True) and d == bool(g["done"][i]), i
    g = golden("cartpole_steps_beyond_done")
    s, sbd = g["start"].astype(np.float64), -1
    for t in range(g["reward"].shape[0]):
        s, r, d, sbd = ref.step(s, 1, sbd)
        assert np.array_equal(s, g["states"][t]) and r == g["reward"][t] and d == bool(g["done"][t]) and sbd == g["sbd"][t]
    g = golden("cartpole_reference_test_trace")                    # done = true; for i: if (done) Reset() else Step(i % 2)
    done, k, lens, cur = True, 0, [], 0
    s, sbd = None, -1
    for i in range(1000):
        if done:
            s, sbd = g["resets"][k].astype(np.float64), -1
            k += 1
            done = False
            if cur:
                lens.append(cur)
            cur = 0
        else:
            s, r, done, sbd = ref.step(s, i % 2, sbd)
            cur += 1
            assert r == 1.0
        assert np.array_equal(s, g["it_state"][i]) and int(done) == g["it_done"][i], i
    assert k == int(g["resets_used"]) and lens == list(g["episode_lengths"])


def test_observation_space_bounds_evaluated_from_the_reference_text(gymnet):
    """CartPoleEnv's ctor builds `high = np.array(x_threshold * 2, float.MaxValue, theta_threshold_radians * 2, float.MaxValue)`
    (CartPoleEnv.cs:46) and `ObservationSpace = new Box(-1 * high, high, np.float32)`.  The engine's gymnet_env_describe must
    report exactly those bounds: the two products are evaluated from the reference's text (float * int folds in binary32)."""
    if not os.path.exists("/root/reference/src/Gym.Environments/Envs/Classic/CartPoleEnv.cs"):
        pytest.skip("reference tree not present (GPU box)")
    import re as _re
    from oracle.evaluate_reference_text import Parser, _tokens, load_reference, reference_constants
    text = load_reference()
    env = reference_constants(text)
    args = _re.search(r"var high = np\.array\(([^;]*)\);", text).group(1).split(",")
    assert len(args) == 4
    fmax = float(np.finfo(np.float32).max)
    want = []
    for a in args:
        a = a.strip()
        if a == "float.MaxValue":
            want.append(fmax)
        else:
            v = Parser(_tokens(a), env).expr()
            assert v.t == "float"
            want.append(float(v.v))
    info = gymnet.env_describe(0)
    assert list(info.obs_high[:4]) == want and list(info.obs_low[:4]) == [-w for w in want]
    assert _re.search(r"ActionSpace = new Discrete\((\d+)\);", text).group(1) == str(info.action_n) == "2"


def test_discrete_contains_evaluated_from_the_reference_text(oracle):
    """Discrete.Contains(int) (src/Gym/Spaces/Discrete.cs:38-40) — the rule GYMNET_FLAG_VALIDATE_ACTIONS applies on the device —
    evaluated from the reference's text for every x in [-3, N + 3) and several N, against the oracle's restatement (which the
    GPU test test_validate_actions compares the kernel with)."""
    path = "/root/reference/src/Gym/Spaces/Discrete.cs"
    if not os.path.exists(path):
        pytest.skip("reference tree not present (GPU box)")
    import re as _re
    from oracle.evaluate_reference_text import Parser, Value, _strip_comments, _tokens
    text = _strip_comments(open(path, encoding="utf-8-sig").read())
    body = _re.search(r"public bool Contains\(int x\)\s*\{\s*return ([^;]+);\s*\}", text).group(1)
    toks = _tokens(body)
    L = oracle.lib()
    for n in (1, 2, 3, 7):
        for x in range(-3, n + 3):
            want = Parser(toks, {"x": Value(x, "int"), "N": Value(n, "int"), "Start": Value(0, "int")}).expr()
            assert want.t == "bool" and bool(L.ref_discrete_contains(x, n)) == bool(want.v), (x, n)


def test_reset_distribution_parameters_evaluated_from_the_reference_text(oracle):
    """CartPoleEnv.Reset() (CartPoleEnv.cs:63-67): `steps_beyond_done = -1; state = random.uniform(low, high, count)`.  The three
    arguments are evaluated from the reference's text; the engine's Philox reset (a different generator by design: north_star)
    must draw `count` components per lane inside [low, high) with the uniform's mean and variance, and leave sbd at the text's value."""
    if not os.path.exists("/root/reference/src/Gym.Environments/Envs/Classic/CartPoleEnv.cs"):
        pytest.skip("reference tree not present (GPU box)")
    import re as _re
    from oracle.evaluate_reference_text import Parser, _tokens, load_reference
    text = load_reference()
    body = text[text.index("public override NDArray Reset()"):]
    body = body[:body.index("return")]
    sbd0 = Parser(_tokens(_re.search(r"steps_beyond_done = ([^;]+);", body).group(1)), {}).expr()
    args = _re.search(r"state = random\.uniform\(([^;]*)\);", body).group(1).split(",")
    low, high, count = (Parser(_tokens(a), {}).expr() for a in args)
    assert (sbd0.v, low.v, high.v, count.v) == (-1, -0.05, 0.05, 4) and count.t == "int"
    s = oracle.cartpole_reset(0x5EED, 0, 0, 200_000)
    assert s.shape[0] == count.v and s.min() >= low.v and s.max() < high.v
    assert abs(float(s.mean()) - (low.v + high.v) / 2) < 2e-4 and abs(float(s.std()) - (high.v - low.v) / np.sqrt(12)) < 2e-4
    st, r, d, b = oracle.cartpole_step(s.astype(np.float64), np.ones(s.shape[1], np.int32))
    assert (b[d == 0] == sbd0.v).all()                                  # a fresh episode starts from the text's steps_beyond_done


# ------------------------------------------------------------------------------------------------
# GYMNET_FLAG_F64: the float64 "kernel semantics" twin (the HIP kernel's own sin / cos, 53-bit reset draws)
# ------------------------------------------------------------------------------------------------
def test_f64_kernel_sincos_accuracy_against_200_bit_arithmetic(oracle):
    """ref_sincos_f64_kernel (= gym.net_amd/csrc/cartpole64.hpp sincos_f64, operation for operation): < 0.75 ulp for
    |x| <= pi/4 — every CartPole pole angle of a live episode — and <= 2.1 ulp over the reduced range |x| <= 8e5; exact
    identities at 0; libm beyond the range."""
    import mpmath
    mpmath.mp.prec = 200
    rng = np.random.default_rng(0)

    def worst_ulp(xs):
        s, c = oracle.sincos_f64_kernel(xs)
        ws = wc = 0.0
        for x, gs, gc in zip(xs, s, c):
            ts, tc = mpmath.sin(mpmath.mpf(float(x))), mpmath.cos(mpmath.mpf(float(x)))
            ws = max(ws, float(abs(mpmath.mpf(float(gs)) - ts) / mpmath.mpf(float(np.spacing(abs(float(ts)))))))
            wc = max(wc, float(abs(mpmath.mpf(float(gc)) - tc) / mpmath.mpf(float(np.spacing(abs(float(tc)))))))
        return ws, wc
    small = np.concatenate([rng.uniform(-0.785, 0.785, 4000), rng.uniform(-0.21, 0.21, 2000), rng.uniform(-1e-4, 1e-4, 500)])
    assert max(worst_ulp(small)) < 0.75
    wide = np.concatenate([rng.uniform(-10, 10, 3000), rng.uniform(-8e5, 8e5, 3000), np.arange(-40, 41) * (np.pi / 2)])
    assert max(worst_ulp(wide)) <= 2.1
    s, c = oracle.sincos_f64_kernel([0.0, 1e-300, -1e-300])
    assert s[0] == 0.0 and c[0] == 1.0 and s[1] == 1e-300 and s[2] == -1e-300 and c[1] == 1.0
    big = np.array([1e6, -3e9, 1e300])
    s, c = oracle.sincos_f64_kernel(big)
    assert np.array_equal(s, np.sin(big)) and np.array_equal(c, np.cos(big))                 # beyond the reduced range: libm
    s, c = oracle.sincos_f64_kernel([np.inf, np.nan])
    assert np.isnan(s).all() and np.isnan(c).all()


def test_f64_kernel_semantics_twin_against_the_reference_arithmetic(oracle, golden):
    """The float64 twin (kernel's own sin / cos) against the reference-arithmetic restatement (libm) and hence against the
    vectors evaluated from the reference's text: integer outputs identical on all 3200, x' and theta' bit-identical (no
    transcendental in them), velocities within 4 ulp-ish (1e-13 relative)."""
    g = golden("cartpole_reference_text")
    s, r, d, b = oracle.cartpole_step(g["state"], g["action"], g["sbd"], dtype=np.float64, kernel_sincos=True)
    assert np.array_equal(d, g["done"]) and np.array_equal(r, g["reward"]) and np.array_equal(b, g["sbd_out"])
    assert np.array_equal(s[[0, 2]], g["next_state"][[0, 2]])
    rel = np.abs(s - g["next_state"]) / np.maximum(1.0, np.abs(g["next_state"]))
    assert rel.max() <= 1e-13, rel.max()
    # free-running on the recorded reference-test-shaped trace: every done flag and episode length, states within 1e-12
    t = golden("cartpole_reference_test_trace")
    done, k, lens, cur, worst = True, 0, [], 0, 0.0
    st = np.zeros((4, 1)); sbd = np.array([-1], np.int32)
    for i in range(1000):
        if done:
            st = t["resets"][k].reshape(4, 1).copy(); k += 1
            sbd[:] = -1; done = False
            if cur:
                lens.append(cur)
            cur = 0
        else:
            st, _, dd, sbd = oracle.cartpole_step(st, np.array([i % 2], np.int32), sbd, dtype=np.float64, kernel_sincos=True)
            done = bool(dd[0]); cur += 1
            worst = max(worst, float(np.abs(st[:, 0] - t["it_state"][i]).max()))
        assert int(done) == t["it_done"][i]
    assert lens == list(t["episode_lengths"]) and worst <= 1e-12


def test_f64_reset_draw_is_a_53_bit_uniform_on_the_reference_interval(oracle):
    """CartPoleEnv.cs:63-67 in float64: low + (high - low) * u, u = ((a >> 5) * 2^26 + (b >> 6)) / 2^53 from two Philox calls
    (key, key ^ 0xC2B2AE3D27D4EB4F) at the lane's counter; the first call is the float32 engine's."""
    n = 200_000
    s = oracle.cartpole_reset_f64(0x5EED, 12345, 3, n)
    assert s.shape == (4, n) and s.min() >= -0.05 and s.max() < 0.05
    assert abs(s.mean()) < 2e-4 and abs(s.std() - 0.1 / np.sqrt(12)) < 2e-4
    assert len(np.unique(s)) == 4 * n                                                        # 53 random bits: no collisions
    # reconstructed from the pinned Philox words (NumPy restatement): bit-identical
    lanes = np.arange(12345, 12345 + n, dtype=np.uint64)
    a = nr.reset_words(0x5EED, lanes, 3).astype(np.uint64)
    b = nr.reset_words(0x5EED ^ 0xC2B2AE3D27D4EB4F, lanes, 3).astype(np.uint64)
    u = ((a >> np.uint64(5)).astype(np.float64) * 67108864.0 + (b >> np.uint64(6)).astype(np.float64)) * (1.0 / 9007199254740992.0)
    assert np.array_equal(s, -0.05 + (0.05 - -0.05) * u)
    # the float32 engine's draw uses the top 24 bits of the same first word: the two agree to float32 resolution
    s32 = oracle.cartpole_reset(0x5EED, 12345, 3, n)
    assert np.abs(s - s32.astype(np.float64)).max() < 2e-8
    per_lane = np.arange(n, dtype=np.uint64) + 9
    assert np.array_equal(oracle.cartpole_reset_f64(0, 0, 3, n, lane_seed=per_lane)[:, 5], oracle.cartpole_reset_f64(int(per_lane[5]), 5, 3, 1)[:, 0])


def test_f64_twin_reproduces_its_committed_bit_patterns(oracle, golden):
    """tests/golden/cartpole_f64_kernel.npz (tests/golden/make_f64_golden.py): sin / cos bit patterns, reset draws and a
    250-step auto-reset trace of the float64 twin.  The HIP kernel is compared with the twin live (tests/test_gpu_f64.py) and
    with this fixture; here the twin itself is pinned, so kernel and twin cannot drift together unnoticed."""
    g = golden("cartpole_f64_kernel")
    s, c = oracle.sincos_f64_kernel(g["sincos_x"])
    assert np.array_equal(s.view(np.uint64), g["sincos_s"].view(np.uint64)) and np.array_equal(c.view(np.uint64), g["sincos_c"].view(np.uint64))
    for k, (off, tick) in enumerate(((0, 0), (123_456_789_000, 7), (1 << 40, 2 ** 33 + 5))):
        assert np.array_equal(oracle.cartpole_reset_f64(0x5EED, off, tick, 64), g["resets"][k])
    seed, off = int(g["trace_seed"]), int(g["trace_offset"])
    st = oracle.cartpole_reset_f64(seed, off, 0, 16)
    for t in range(g["trace_actions"].shape[0]):
        st, r, d = oracle.cartpole_autoreset_step_f64(seed, off, 1 + t, st, g["trace_actions"][t])
        assert np.array_equal(st, g["trace_states"][t]) and np.array_equal(d, g["trace_done"][t]), t
    assert 5 < int(g["trace_done"].sum()) < 16 * 250 / 8                                      # episodes do end and restart in it


def test_f64_kernel_sincos_dense_sweep_against_libm(oracle):
    """4 x 10^6 arguments: the kernel's float64 sin / cos never differ from glibc's (themselves < 1 ulp) by more than 1 ulp on
    |x| <= pi/4 — so a CartPole step in the float64 mode differs from the reference-arithmetic restatement by last-bit effects
    only — and by more than 3 ulp on the reduced range; identical results for the vast majority."""
    rng = np.random.default_rng(11)

    def ulp_diff(a, b):
        return np.abs(a.view(np.int64) - b.view(np.int64))      # same sign and binade neighbourhood: distance in representable values
    x = rng.uniform(-np.pi / 4, np.pi / 4, 2_000_000)
    s, c = oracle.sincos_f64_kernel(x)
    ok = np.sign(s) == np.sign(np.sin(x))
    assert ok.all() and ulp_diff(s, np.sin(x)).max() <= 1 and ulp_diff(c, np.cos(x)).max() <= 1
    assert (s == np.sin(x)).mean() > 0.95 and (c == np.cos(x)).mean() > 0.93
    x = np.concatenate([rng.uniform(-50, 50, 1_000_000), rng.uniform(-8e5, 8e5, 1_000_000)])
    s, c = oracle.sincos_f64_kernel(x)
    ws, wc = np.sin(x), np.cos(x)
    big = (np.abs(ws) > 1e-6) & (np.abs(wc) > 1e-6)             # away from the zeros, where an ulp is tiny and the reduction error shows
    assert ulp_diff(s[big], ws[big]).max() <= 3 and ulp_diff(c[big], wc[big]).max() <= 3
    assert np.abs(s - ws).max() < 3e-16 and np.abs(c - wc).max() < 3e-16


def test_f64_constant_division_is_proved_and_sampled(oracle):
    """The float64 kernel's `/ total_mass` (CartPoleEnv.cs:149-151) is fma(x, ZH, x * ZL) (cartpole64.hpp DivByTotalMass64, mirrored
    in the oracle's twin).  (1) tools/prove_div_total_mass_f64.py PROVES in exact rational arithmetic that it equals IEEE division
    for every binary64 dividend — the divisor is a 24-bit constant, so no quotient comes within 1/(2 * 9227469) ulp of a rounding
    breakpoint while the fma pair is within 2^-53 ulp of the quotient — checks the ZL literals of the product and of the twin, and
    verifies the constructed hardest dividends; (2) 1e8 pseudo-random dividends against the hardware's division: 0 mismatches;
    (3) the twin built on it still reproduces the bit patterns committed in round 4 from the twin that divided
    (test_f64_twin_reproduces_its_committed_bit_patterns) and equals the plain-division restatement on random states."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "prove_div_total_mass_f64.py"), "--samples", "1500"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("PROVED"), r.stdout[-1500:] + r.stderr[-1500:]
    assert "0 mismatches" in r.stdout and "== RN(1/C - ZH)" in r.stdout
    assert oracle.check_div_total_mass_f64(0x5EED, 100_000_000) == 0
    L = oracle.lib()
    C = oracle.cartpole_constants()["total_mass"]
    for x in (10.0, -10.0, 0.1000000015, 3.3e-9, -7.25e11, 1e-200, 0.0):
        assert L.ref_div_total_mass_f64_kernel(x) == x / C
    # outside the theorem, documented (ZL < 0): the sign of a zero quotient.  An infinite dividend is handed through like the division
    # does (round 6, ADVICE r5: the fma pair alone said NaN, and a lane stepped far past done can get there from a finite state)
    assert L.ref_div_total_mass_f64_kernel(-0.0) == 0.0
    assert L.ref_div_total_mass_f64_kernel(np.inf) == np.inf and L.ref_div_total_mass_f64_kernel(-np.inf) == -np.inf
    # ... so a state whose theta_dot^2 overflows steps like the reference's arithmetic: done (inf compares), not NaN-and-false
    big = np.array([[0.0, 0.0, 0.0], [0.0, 1.0, -2.0], [0.3, -0.1, 0.05], [1e160, -1e170, 1e200]])
    a = np.array([0, 1, 1], np.int32)
    twin_s, twin_r, twin_d, _ = oracle.cartpole_step(big, a, dtype=np.float64, kernel_sincos=True)
    ref_s, ref_r, ref_d, _ = oracle.cartpole_step(big, a)
    assert np.array_equal(twin_d, ref_d) and np.array_equal(twin_r, ref_r)
    assert np.array_equal(np.isinf(twin_s), np.isinf(ref_s)) and np.array_equal(np.isnan(twin_s), np.isnan(ref_s))
    nxt_t = oracle.cartpole_step(twin_s, a, dtype=np.float64, kernel_sincos=True)
    nxt_r = oracle.cartpole_step(ref_s, a)
    assert np.array_equal(nxt_t[2], nxt_r[2]) and nxt_r[2].all()                 # the step after: theta' = theta + tau * (+-inf) -> done
    # twin (fma pair) == the same operation sequence with the plain division, on states incl. large angles and velocities
    rng = np.random.default_rng(5)
    n = 200_000
    s = np.stack([rng.uniform(-3, 3, n), rng.uniform(-30, 30, n), rng.uniform(-8, 8, n), rng.uniform(-40, 40, n)])
    a = rng.integers(0, 2, n).astype(np.int32)
    got = oracle.cartpole_step(s, a, dtype=np.float64, kernel_sincos=True)[0]
    sn, cs = oracle.sincos_f64_kernel(s[2])
    f = np.where(a == 1, 10.0, -10.0)
    pml, mp, g, ln, tau = (oracle.cartpole_constants()[k] for k in ("polemass_length", "masspole", "gravity", "length", "tau"))
    temp = (f + pml * s[3] * s[3] * sn) / C
    thacc = (g * sn - cs * temp) / (ln * (4.0 / 3.0 - mp * cs * cs / C))
    xacc = temp - pml * thacc * cs / C
    want = np.stack([s[0] + tau * s[1], s[1] + tau * xacc, s[2] + tau * s[3], s[3] + tau * thacc])
    assert np.array_equal(got, want)


def test_action_stream_is_sharding_invariant_for_ragged_shard_plans(oracle):
    """Action stream v2 keys a call by the GLOBAL lane's group (lane >> 2): a batch cut into shards whose offsets are not multiples of
    four (rank r owns [r N / G, (r + 1) N / G), here 10 lanes over 3 ranks and 1003 over 8) samples, composes and draws its
    masks exactly as the whole batch does — shard by shard, from the shard's own lane offset."""
    for n, g in ((10, 3), (1003, 8), (4099, 5)):
        whole_a, whole_b = oracle.action_words(0xABCD, 0, 9, n)
        whole_d = oracle.discrete_sample(0xABCD, 0, 9, 3, 0, n)
        whole_x = oracle.box_uniform_sample(0xABCD, 0, 9, -2.0, 2.0, n)
        pol = (np.arange(n) % 3).astype(np.int32)
        whole_c = oracle.compose_discrete(0xABCD, 0, 9, 3, 0.5, pol)
        off = 0
        for r in range(g):
            assert off == n * r // g
            cnt = n * (r + 1) // g - n * r // g                          # gym.net_amd/sharding.py ShardPlan.offset / count
            a, b = oracle.action_words(0xABCD, off, 9, cnt)
            assert np.array_equal(a, whole_a[off:off + cnt]) and np.array_equal(b, whole_b[off:off + cnt])
            assert np.array_equal(oracle.discrete_sample(0xABCD, off, 9, 3, 0, cnt), whole_d[off:off + cnt])
            assert np.array_equal(oracle.box_uniform_sample(0xABCD, off, 9, -2.0, 2.0, cnt), whole_x[off:off + cnt])
            assert np.array_equal(oracle.compose_discrete(0xABCD, off, 9, 3, 0.5, pol[off:off + cnt]), whole_c[off:off + cnt])
            off += cnt
        assert off == n
