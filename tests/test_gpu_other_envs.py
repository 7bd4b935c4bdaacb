"""GPU parity for Pendulum / MountainCar / Acrobot.  These envs are ABSENT from the reference
(README.md:69-76 lists them as unchecked roadmap items), so the oracle here restates the upstream
openai/gym algorithms and parity is unpinned by construction; the tests hold the
HIP kernels to the same bars anyway: integer outputs exact, float32 state within 1e-5 per step."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SEED = 0x5EED


def test_pendulum_teacher_forced(gpu_pkg, golden, oracle):
    g = golden("other_envs")
    n = g["pe_state"].shape[1]
    with gpu_pkg.VectorEnv("Pendulum-v1", n, seed=SEED) as env:
        r0 = env.Reset()
        st0 = env.GetState()
        assert np.array_equal(st0, oracle.pendulum_reset(SEED, 0, 0, n))          # theta~U(-pi,pi), thdot~U(-1,1)
        assert np.abs(r0[:, 0] - np.cos(st0[0])).max() < 1e-6 and np.array_equal(r0[:, 2], st0[1])
        env.SetState(g["pe_state"])
        out = env.Step(g["pe_action"])                                            # Box action, clipped to [-2, 2]
        got = env.GetState().astype(np.float64)
    assert np.abs(got - g["pe_next"]).max() <= 1e-5
    assert np.abs(out.Observation.astype(np.float64) - g["pe_obs"].T).max() <= 1e-5
    assert np.abs(out.Reward.astype(np.float64) - g["pe_reward"]).max() <= 1e-4 * 16.3   # |reward| <= 16.27
    assert not out.Done.any()                                                     # Pendulum never terminates
    assert out.Observation.shape == (n, 3)


def test_mountaincar_teacher_forced(gpu_pkg, golden, oracle):
    g = golden("other_envs")
    n = g["mc_state"].shape[1]
    with gpu_pkg.VectorEnv("MountainCar-v0", n, seed=SEED) as env:
        env.Reset()
        assert np.array_equal(env.GetState(), oracle.mountaincar_reset(SEED, 0, 0, n))
        env.SetState(g["mc_state"])
        out = env.Step(g["mc_action"])
        got = env.GetState().astype(np.float64)
    assert np.abs(got - g["mc_next"]).max() <= 1e-6
    # done = (p >= 0.5 and v >= 0): exact except where float64 p' is within rounding of 0.5
    near = np.abs(g["mc_next"][0] - 0.5) < 1e-6
    assert np.array_equal(out.Done[~near], g["mc_done"].astype(bool)[~near]) and near.sum() <= 1
    assert np.all(out.Reward == -1.0)
    assert np.all(got[1, :8] == 0.0)                                               # inelastic left wall


def test_acrobot_teacher_forced(gpu_pkg, golden, oracle):
    g = golden("other_envs")
    n = g["ac_state"].shape[1]
    with gpu_pkg.VectorEnv("Acrobot-v1", n, seed=SEED) as env:
        env.Reset()
        assert np.array_equal(env.GetState(), oracle.acrobot_reset(SEED, 0, 0, n))
        env.SetState(g["ac_state"])
        out = env.Step(g["ac_action"])
        got = env.GetState().astype(np.float64)
    want = g["ac_next"]
    # angles wrap at +-pi: compare on the circle.  RK4 over dt = 0.2 with |velocities| up to 28 amplifies float32 rounding; how
    # much is MEASURED, not assumed: a literal float32 transcription of upstream's formulas (libm sinf / cosf, IEEE division,
    # upstream's association: oracle ref_acrobot_step_f32_literal) is evaluated on the same inputs, and the kernel's instruction-diet
    # form must stay within 2 x its error (profiles/acrobot_accuracy_r04.txt: the diet is not the cause)
    dang = np.abs(np.angle(np.exp(1j * (got[:2] - want[:2]))))
    dvel = np.abs(got[2:] - want[2:])
    lit = oracle.acrobot_step_f32_literal(g["ac_state"], g["ac_action"])[0].astype(np.float64)
    lang = np.abs(np.angle(np.exp(1j * (lit[:2] - want[:2]))))
    lvel = np.abs(lit[2:] - want[2:])
    # typical lanes (median, 99th percentile): the kernel is no worse than the literal transcription (+25 % sampling slack);
    # the single worst of 1024 lanes is a tail statistic of an ill-conditioned map: within 4 x
    for q in (0.5, 0.99):
        assert np.quantile(dang, q) <= 1.25 * np.quantile(lang, q) and np.quantile(dvel, q) <= 1.25 * np.quantile(lvel, q), q
    assert dang.max() <= 4 * lang.max() and dvel.max() <= 4 * lvel.max(), (dang.max(), lang.max(), dvel.max(), lvel.max())
    assert lang.max() <= 2e-5 and lvel.max() <= 2e-4                                  # the yardstick itself: 5 x tighter than round 3's 1e-4 / 1e-3
    calm = (np.abs(g["ac_state"][2]) < 2) & (np.abs(g["ac_state"][3]) < 2)
    assert np.abs(got[2:, calm] - want[2:, calm]).max() <= 2e-5 and dang[:, calm].max() <= 1e-5
    s32 = oracle.acrobot_step(g["ac_state"], g["ac_action"], dtype=np.float32)       # kernel semantics: tight
    d32 = np.abs(np.angle(np.exp(1j * (got[:2] - s32[0][:2].astype(np.float64)))))
    assert d32.max() <= 2e-5
    margin = np.abs((-np.cos(want[0]) - np.cos(want[1] + want[0])) - 1.0) < 1e-4
    assert np.array_equal(out.Done[~margin], g["ac_done"].astype(bool)[~margin])
    assert np.array_equal(out.Reward, np.where(out.Done, 0.0, -1.0).astype(np.float32))
    assert out.Observation.shape == (n, 6)
    assert np.abs(out.Observation[:, 0].astype(np.float64) - np.cos(got[0])).max() < 1e-6


@pytest.mark.parametrize("name,nact", [("Pendulum-v1", None), ("MountainCar-v0", 3), ("Acrobot-v1", 3)])
def test_autoreset_rollout_stays_in_bounds_and_matches_f32_oracle(gpu_pkg, oracle, name, nact):
    n, steps = 8192, 50
    rng = np.random.default_rng(12)
    step32 = {"Pendulum-v1": oracle.pendulum_step, "MountainCar-v0": oracle.mountaincar_step, "Acrobot-v1": oracle.acrobot_step}[name]
    with gpu_pkg.VectorEnv(name, n, seed=SEED, auto_reset=True) as env:
        env.Reset()
        lo, hi = env.ObservationSpace.Low, env.ObservationSpace.High
        for t in range(steps):
            a = rng.uniform(-2, 2, n).astype(np.float32) if nact is None else rng.integers(0, nact, n).astype(np.int32)
            sub = slice(0, 256)
            s = env.GetState()
            out = env.Step(a)
            obs = out.Observation
            assert np.all(obs >= lo - 1e-6) and np.all(obs <= hi + 1e-6)           # ObservationSpace.Contains
            ref = step32(s[:, sub], a[sub], dtype=np.float32)
            keep = ~out.Done[sub]
            got = env.GetState()[:, sub]
            if keep.any():
                assert np.abs(got[:, keep] - ref[0][:, keep]).max() <= 3e-5


@pytest.mark.parametrize("name", ["Pendulum-v1", "MountainCar-v0", "Acrobot-v1"])
@pytest.mark.parametrize("n", [4096, 1 << 20])
def test_kernels_bit_identical_to_float32_restatement(gpu_pkg, oracle, name, n):
    # same claim as for CartPole: every float32 operation of the kernel is restated in the oracle's
    # kernel-semantics functions (own sin/cos, IEEE ops) => bit-for-bit equal states, observations, flags.
    # n = 2^20 is BASELINE's batch (configs 3 and 4): there the launch policy picks the kernels the bench times
    # (dwordx4 lanes; for Acrobot the multi-lane step_kernel_pipe) — the small case runs the scalar-lane forms.
    rng = np.random.default_rng(31)
    if name == "Pendulum-v1":
        s = np.stack([rng.uniform(-8, 8, n), rng.uniform(-8, 8, n)]).astype(np.float32)
        # a free-running Pendulum never wraps theta: large angles of both signs (the kernel's exact fast fmod), the neighbourhood of
        # multiples of 2 pi (a quotient off by one would show there) and angles beyond 2^22 * 2 pi (the fmodf fallback)
        s[0, ::5] = rng.uniform(-6e4, 6e4, s[0, ::5].shape).astype(np.float32)
        k = rng.integers(-4000, 4000, s[0, 1::7].shape)
        s[0, 1::7] = (k * (2 * np.pi) - np.pi + rng.uniform(-1e-3, 1e-3, k.shape)).astype(np.float32)
        s[0, 2::97] = rng.uniform(-4e8, 4e8, s[0, 2::97].shape).astype(np.float32)
        a = rng.uniform(-2.5, 2.5, n).astype(np.float32)
        want = oracle.pendulum_step(s, a, dtype=np.float32)
        ws, wo, wr, wd = want
    elif name == "MountainCar-v0":
        s = np.stack([rng.uniform(-1.2, 0.6, n), rng.uniform(-0.07, 0.07, n)]).astype(np.float32)
        a = rng.integers(0, 3, n).astype(np.int32)
        ws, wr, wd = oracle.mountaincar_step(s, a, dtype=np.float32)
        wo = ws
    else:
        s = np.stack([rng.uniform(-3.1, 3.1, n), rng.uniform(-3.1, 3.1, n), rng.uniform(-12, 12, n), rng.uniform(-28, 28, n)]).astype(np.float32)
        a = rng.integers(0, 3, n).astype(np.int32)
        ws, wo, wr, wd = oracle.acrobot_step(s, a, dtype=np.float32)
    with gpu_pkg.VectorEnv(name, n, seed=SEED) as env:
        if n == 1 << 20:
            pol = env.LaunchPolicy()
            assert pol["sequential_lanes_per_thread"] == (4 if name == "Acrobot-v1" else 1) and pol["envs_per_thread"] == (1 if name == "Acrobot-v1" else 4)
        env.Reset(); env.SetState(s)
        out = env.Step(a)
        # sin / cos beyond |x| = 65536 fall back to OCML on the GPU and libm on the CPU (they may differ by an ulp, envs.hpp): state
        # and observation are compared where the angle is inside that range, the reward (no trigonometry in it) everywhere
        ok = np.abs(s[0]) <= 65536.0 if name == "Pendulum-v1" else np.ones(n, bool)
        assert ok.sum() > 0.98 * n
        assert np.array_equal(env.GetState()[:, ok], ws[:, ok])
        assert np.array_equal(out.Observation[ok], wo.T[ok])
        assert np.array_equal(out.Reward, wr.astype(np.float32)) and np.array_equal(out.Done, wd.astype(bool))


@pytest.mark.parametrize("name", ["Pendulum-v1", "MountainCar-v0", "Acrobot-v1"])
def test_episode_bookkeeping_on_every_env(gpu_pkg, name):
    # done-list compaction, episode statistics and the time-limit extension are env-agnostic kernel features
    n, steps, limit = 6000, 45, 20
    rng = np.random.default_rng(41)
    with gpu_pkg.VectorEnv(name, n, seed=SEED, auto_reset=True, done_list=True, episode_stats=True, final_obs=True,
                           max_episode_steps=limit) as env:
        env.Reset()
        ln = np.zeros(n, np.int32); ret = np.zeros(n, np.float64)
        fin_len = np.zeros(n, np.int32); fin_ret = np.zeros(n, np.float64)
        got_len = np.zeros(n, np.int32); got_ret = np.zeros(n, np.float64)
        for t in range(steps):
            a = rng.uniform(-2, 2, n).astype(np.float32) if name == "Pendulum-v1" else rng.integers(0, 3, n).astype(np.int32)
            out = env.Step(a)
            ln += 1; ret += out.Reward
            d = out.Done
            assert sorted(env.DoneLanes().tolist()) == np.nonzero(d)[0].tolist()
            fin_len[d] = ln[d]; fin_ret[d] = ret[d]; ln[d] = 0; ret[d] = 0
            rec = env.DoneRecords()                                                  # this step's compact records
            assert sorted(rec["lanes"].tolist()) == np.nonzero(d)[0].tolist() and rec["final_obs"].shape == (int(d.sum()), env.ObsDim)
            got_len[rec["lanes"]] = rec["length"]; got_ret[rec["lanes"]] = rec["return"]
            assert np.isfinite(rec["final_obs"]).all()
            if d.any():
                assert np.array_equal(env.FinalObs()[rec["lanes"]], rec["final_obs"])   # dense view: the latest records applied
        assert np.array_equal(got_len, fin_len) and got_len.max() == limit       # truncation at the limit
        assert np.abs(got_ret - fin_ret).max() <= 1e-3 * max(1.0, np.abs(fin_ret).max())
        if name == "Pendulum-v1":
            assert set(np.unique(got_len)) == {limit}                                # Pendulum only ever ends by truncation
        d_ret, d_len = env.EpisodeStats()                                            # dense view after the last step's records
        assert np.array_equal(d_len[rec["lanes"]], rec["length"])


def test_long_fused_rollout_stays_deterministic_and_in_bounds(gpu_pkg):
    import torch
    n, T, ring = 1 << 14, 6000, 16
    dev = torch.device("cuda", 0)
    with gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED, auto_reset=True) as f, gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED, auto_reset=True) as e:
        acts = torch.empty((ring, n), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        for t in range(ring):
            f.SampleActionsDevice(acts[t], seed=5, tick=t)
        f.Sync()
        f.ResetDevice(); e.ResetDevice()
        f.RolloutFusedDevice(acts, T, n, ring)
        e.RolloutDevice(acts, T, n, ring)                # graph replay at this size
        f.Sync(); e.Sync()
        a, b = f.GetState(), e.GetState()
        assert np.array_equal(a, b) and np.isfinite(a).all()
        assert np.abs(a[0]).max() <= 2.5 and np.abs(a[2]).max() <= 0.3
        assert f.Tick == e.Tick == T + 1 and f.Counters()["tick"] == T + 1 == e.Counters()["tick"]


# tools/determinism_probe.py, identical across runs and across MI355X boxes.  Re-pinned in round 2 for ONE reason: space
# sampling moved to its own Philox stream (key ^ 0x9E3779B97F4A7C15), so the device-sampled actions this test
# feeds changed; reset draws and all arithmetic are unchanged (the round-1 values were cba10d62…, 7732d586…, ebcf1106…,
# 014e67b1…; with the new action stream and round 1's Acrobot arithmetic: 23693d80a6e0b6cfafddc3b5).
# Acrobot was then re-pinned a second time, for a change of ARITHMETIC: its float32 evaluation scheme was rewritten for
# instruction count (explicit fma, one reproducible reciprocal per RK4 stage; envs.hpp Acrobot::dsdt) — 682 -> 454 VALU per
# env-step.  The CPU twin was rewritten with it and test_kernels_bit_identical_to_float32_restatement still holds; the other
# three envs kept their values through every kernel-side change of the round (sign-bit quadrant logic in sincos_f32, SLP
# vectoriser off, load / compute split), which is the evidence that those changes did not move a bit.
# Round 3 re-pinned Acrobot once more, again for a change of ARITHMETIC (its sin/cos became the small-argument form sincos_small:
# two-constant reduction, Horner cosine — last-bit differences; round 2's value was d9b63ec699510c7a8a681f23).  The wrap written
# on the magnitude and the v_med3 clamps are bit-neutral.  CartPole / Pendulum / MountainCar kept their round-2 values through
# round 3's kernel changes (wave-compacted reset, state rows stored once in the observation array, re-shaped fused rollout).
# Round 4 re-pinned CartPole, for a change of its INTEGER output: the done flag now comes from the float64 sums the reference
# compares (CartPoleEnv.cs:154,156,167; envs.hpp CartPole::step) instead of their float32 roundings — over 2 * 10^9 env-steps a
# few lanes terminate one step earlier / later than before, and their later reset draws move with them (round 3's value was
# 4b33a229e81d658d7240b98f).  The float32 state arithmetic did not change; test_kernels_bit_identical_to_float32_restatement and
# the 2^20-lane replay in test_gpu_bench_kernels.py hold against the CPU twin, which received the same change.
# Round 6 re-pinned all four, for a change of the ACTION STREAM (version 2, csrc/philox.hpp / ABI 6: one Philox call per group of four
# global lanes): the test's action ring is device-sampled, so every lane takes different actions than before.  No env arithmetic,
# reset draw, tick protocol or lane mapping changed — the ring-action parity tests and the 2^20-lane oracle replays
# (test_gpu_bench_kernels.py, caller-supplied actions) kept passing unchanged across the switch.  Round 5's values were
# 7341b9f5e1f437f81b892a97 / 2bfb470ea7fe5499e34837d8 / 4a4759dc8c8d567686ac7ba0 / 1bcaf2e8b8c02e0a99185330.
LONG_ROLLOUT_SHA256 = {
    "CartPole-v1": "b214434fa1fa7088410a99ad",
    "Pendulum-v1": "d78e9197e55aa2416fc6979d",
    "MountainCar-v0": "fbd3abdf9f00b0b15a5163c4",
    "Acrobot-v1": "0bba16a09e4f1d27513041a1",
}


@pytest.mark.parametrize("name", sorted(LONG_ROLLOUT_SHA256))
def test_full_size_long_rollout_checksum(gpu_pkg, name):
    """Checksum pin at BASELINE's full batch: 2^20 lanes, 1000 one-launch steps + 1000 fused steps with auto-reset and
    device-sampled actions.  Every operation on the path is IEEE-exact and ordered, so the final state is a fixed bit
    pattern; any change to the arithmetic, the Philox stream, the tick protocol or the lane mapping moves it."""
    import hashlib
    import torch
    n, ring = 1 << 20, 32
    with gpu_pkg.VectorEnv(name, n, seed=SEED, auto_reset=True) as env:
        adt = torch.float32 if name == "Pendulum-v1" else torch.int32
        acts = torch.empty((ring, n), dtype=adt, device="cuda")
        torch.cuda.synchronize()
        for t in range(ring):
            env.SampleActionsDevice(acts[t], seed=1, tick=t)
        env.ResetDevice()
        env.RolloutDevice(acts, 1000, n, ring)
        env.RolloutFusedDevice(acts, 1000, n, ring)
        env.Sync()
        assert env.Tick == 2001
        assert hashlib.sha256(env.GetState().tobytes()).hexdigest()[:24] == LONG_ROLLOUT_SHA256[name]


def test_scalar_broadcast_on_a_box_action_space(gpu_pkg):
    # IVecEnv.Step(int) (IVecEnv.cs:15) on Pendulum: the int is the (scalar) torque for every lane
    n = 2000
    with gpu_pkg.VectorEnv("Pendulum-v1", n, seed=SEED) as a, gpu_pkg.VectorEnv("Pendulum-v1", n, seed=SEED) as b:
        a.Reset(); b.Reset()
        for u in (1, -2, 0, 5):                               # 5 is clipped to max_torque = 2 like np.clip(u, -2, 2)
            oa = a.Step(u)
            ob = b.Step(np.full(n, float(u), dtype=np.float32))
            assert np.array_equal(oa.Observation, ob.Observation) and np.array_equal(oa.Reward, ob.Reward)
        oc = a.StepAsync(1).Result(); od = b.Step(np.ones(n, np.float32))
        assert np.array_equal(oc.Observation, od.Observation)


def test_acrobot_kernel_forms_are_bit_identical(gpu_pkg, monkeypatch):
    """Acrobot's step has three kernel forms: one lane per thread (one-shot), launch policy vec = 2 — both envs of a thread ride the
    v_pk_*_f32 instructions (envs.hpp step_observe_x2, dwordx2 streams; opt-in: half the VALU count and still slower,
    docs/ledger.md §4a) — and sequential_lanes = k — k lanes per thread, all loads first, then compute / store lane after lane
    (step_kernel_pipe; the default around 2^20 lanes).  Per lane all three run the same IEEE sequence, so everything must
    agree bit for bit: one-launch steps with an odd lane count (clamped loads / suppressed stores in the tail), the
    bookkeeping variant (which falls back to the one-shot kernel), the fused rollout, with and without auto-reset."""
    import torch
    n, ring = 8192 + 7, 6
    out = {}
    for vec, items in ((1, 1), (2, 1), (1, 2), (1, 3), (1, 4), (1, 5)):
        res = []
        for auto, stats in ((True, False), (False, False), (True, True)):
            # (the bookkeeping variant has the one-shot kernel only: an explicit multi-lane request for it is refused)
            with gpu_pkg.VectorEnv("Acrobot-v1", n, seed=SEED, auto_reset=auto, episode_stats=stats, done_list=stats,
                                   launch_policy={"vec": vec, "sequential_lanes": 1 if stats else items, "lds_pipe": 0}) as env:
                if stats and items > 1:
                    with pytest.raises(ValueError, match="would not take effect"):
                        env.SetLaunchPolicy(sequential_lanes=items)
                pol = env.LaunchPolicy()
                assert pol["envs_per_thread"] == vec and pol["sequential_lanes_per_thread"] == (1 if stats else items)
                acts = torch.empty((ring, n + (n % 2)), dtype=torch.int32, device="cuda")
                torch.cuda.synchronize()
                for t in range(ring):
                    env.SampleActionsDevice(acts[t], seed=5, tick=t)
                env.ResetDevice()
                rng = np.random.default_rng(4)
                s0 = np.stack([rng.uniform(-3.1, 3.1, n), rng.uniform(-3.1, 3.1, n), rng.uniform(-12, 12, n), rng.uniform(-28, 28, n)]).astype(np.float32)
                env.SetState(s0)                                    # fast states: wraps, clamps and terminations all happen
                env.RolloutDevice(acts, 40, n + (n % 2), ring)
                if not stats:
                    env.RolloutFusedDevice(acts, 25, n + (n % 2), ring)
                env.Sync()
                r = env.Read()
                res.append((env.GetState(), r.Observation, r.Reward, r.Done, env.EpisodeStats() if stats else None))
        out[(vec, items)] = res
    # the multi-lane kernel writing into the OTHER observation buffer (GYMNET_FLAG_DOUBLE_BUFFER)
    with gpu_pkg.VectorEnv("Acrobot-v1", n, seed=SEED, auto_reset=True, double_buffer=True, launch_policy={"vec": 1, "sequential_lanes": 4}) as env:
        acts = torch.empty((ring, n + (n % 2)), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        for t in range(ring):
            env.SampleActionsDevice(acts[t], seed=5, tick=t)
        env.ResetDevice()
        rng = np.random.default_rng(4)
        s0 = np.stack([rng.uniform(-3.1, 3.1, n), rng.uniform(-3.1, 3.1, n), rng.uniform(-12, 12, n), rng.uniform(-28, 28, n)]).astype(np.float32)
        env.SetState(s0)
        env.RolloutDevice(acts, 40, n + (n % 2), ring)
        env.RolloutFusedDevice(acts, 25, n + (n % 2), ring)
        env.Sync()
        r = env.Read()
        out[("double buffer", 4)] = [(env.GetState(), r.Observation, r.Reward, r.Done, None)]
    ref = out[(1, 1)]
    for key, res in out.items():
        for a, b in zip(ref, res):
            for x, y in zip(a[:4], b[:4]):
                assert np.array_equal(x, y, equal_nan=True), key
            if a[4] is not None:
                assert np.array_equal(a[4][0], b[4][0]) and np.array_equal(a[4][1], b[4][1]), key
    assert ref[0][3].any() or ref[1][3].any()


@pytest.mark.parametrize("tiles", [2, 3, 4, 5])
def test_acrobot_producer_consumer_kernel_is_bit_identical(gpu_pkg, monkeypatch, tiles):
    """step_kernel_lds (launch policy lds_pipe = 1): computing waves prefetch the next tile and hand results to a storing wave through
    LDS.  Same per-lane code and Philox counters as the one-shot kernel, so states, observations, rewards and done flags must
    agree bit for bit — workgroups with fewer tiles than TPB (ragged grid), with and without auto-reset, energetic states so the
    fused reset runs."""
    import torch
    n, ring = 512 * 37, 6                       # 37 tiles: the last workgroup of every TPB owns fewer tiles than the others
    rng = np.random.default_rng(4)
    s0 = np.stack([rng.uniform(-3.1, 3.1, n), rng.uniform(-3.1, 3.1, n), rng.uniform(-12, 12, n), rng.uniform(-28, 28, n)]).astype(np.float32)
    res = {}
    for form in ("one-shot", "lds"):
        pol = {"vec": 1, "sequential_lanes": 1 if form == "one-shot" else tiles, "lds_pipe": 0 if form == "one-shot" else 1}
        out = []
        for auto in (True, False):
            with gpu_pkg.VectorEnv("Acrobot-v1", n, seed=SEED, auto_reset=auto, launch_policy=pol) as env:
                want = f"step_kernel_lds<Acrobot,{tiles},{str(auto).lower()},15>" if form == "lds" else f"step_kernel<Acrobot,1,{str(auto).lower()},false,15,0>"
                assert env.KernelName() == want, env.KernelName()
                acts = torch.empty((ring, n), dtype=torch.int32, device="cuda")
                torch.cuda.synchronize()
                for t in range(ring):
                    env.SampleActionsDevice(acts[t], seed=5, tick=t)
                env.ResetDevice()
                env.SetState(s0)
                env.RolloutDevice(acts, 25, n, ring)
                env.Sync()
                r = env.Read()
                out.append((env.GetState(), r.Observation, r.Reward, r.Done))
                assert r.Done.any() or not auto
        res[form] = out
    for x, y in zip(res["one-shot"], res["lds"]):
        for u, v in zip(x, y):
            assert np.array_equal(u, v, equal_nan=True)
    with gpu_pkg.VectorEnv("Acrobot-v1", 512 * 8 + 3, seed=SEED, auto_reset=True, launch_policy={"sequential_lanes": 4}) as env:
        with pytest.raises(ValueError):                                                        # not whole 512-lane tiles: refused ...
            env.SetLaunchPolicy(lds_pipe=1)
        assert env.KernelName() == "step_kernel_pipe<Acrobot,4,true,15>"                         # ... and nothing changed


@pytest.mark.parametrize("name", ["Acrobot-v1", "Pendulum-v1"])
def test_set_state_of_get_state_leaves_the_observation_unchanged(gpu_pkg, name):
    """Acrobot's step / reset write their observation with sincos_small, set_state recomputed it with
    sincos_f32 — a last-bit difference for some angles, which broke the bit-exactness Checkpoint / Restore promise for
    observations.  observe() now uses the step's own sin / cos for every angle a step can produce: after stepping into energetic
    states, SetState(GetState()) must leave every observation bit unchanged."""
    n = 50_000
    rng = np.random.default_rng(12)
    with gpu_pkg.VectorEnv(name, n, seed=SEED, auto_reset=True) as env:
        env.Reset()
        if name == "Acrobot-v1":
            env.SetState(np.stack([rng.uniform(-3.1, 3.1, n), rng.uniform(-3.1, 3.1, n), rng.uniform(-12, 12, n), rng.uniform(-28, 28, n)]).astype(np.float32))
            acts = rng.integers(0, 3, (6, n)).astype(np.int32)
        else:
            acts = rng.uniform(-2, 2, (6, n)).astype(np.float32)
        for t in range(6):
            env.Step(acts[t])
        before = env.Read().Observation.copy()
        env.SetState(env.GetState())
        after = env.Read().Observation
        assert np.array_equal(before, after)


def test_acrobot_lane_pair_multi_lane_kernel_is_bit_identical(gpu_pkg):
    """step_kernel_pipe2 (launch policy vec = 2 with sequential_lanes = 2..4; round 4 probe): ITEMS lane PAIRS per thread, 8-byte
    accesses, scalar arithmetic lane after lane, all loads first.  Same per-lane code and Philox counters as the one-shot kernel:
    bit-identical states / observations / rewards / done flags with and without auto-reset; for batches that are not whole
    2 * ITEMS * 256-lane groups the request is refused and the packed two-lane one-shot kernel stays."""
    import torch
    n, ring = 2 * 256 * 12 * 3, 6
    rng = np.random.default_rng(4)
    s0 = np.stack([rng.uniform(-3.1, 3.1, n), rng.uniform(-3.1, 3.1, n), rng.uniform(-12, 12, n), rng.uniform(-28, 28, n)]).astype(np.float32)
    acts = torch.randint(0, 3, (ring, n), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    res = {}
    for auto in (True, False):
        for vec, items in ((1, 1), (2, 2), (2, 3), (2, 4)):
            with gpu_pkg.VectorEnv("Acrobot-v1", n, seed=SEED, auto_reset=auto, launch_policy={"vec": vec, "sequential_lanes": items}) as env:
                want = f"step_kernel<Acrobot,1,{str(auto).lower()},false,15,0>" if vec == 1 else f"step_kernel_pipe2<Acrobot,{items},{str(auto).lower()},15>"
                assert env.KernelName() == want, env.KernelName()
                env.ResetDevice(); env.SetState(s0)
                env.RolloutDevice(acts, 25, n, ring)
                env.Sync()
                r = env.Read()
                res[(auto, vec, items)] = (env.GetState(), r.Observation, r.Reward, r.Done)
        for key in ((auto, 2, 2), (auto, 2, 3), (auto, 2, 4)):
            assert all(np.array_equal(u, v, equal_nan=True) for u, v in zip(res[(auto, 1, 1)], res[key])), key
        assert res[(auto, 1, 1)][3].any()
    with gpu_pkg.VectorEnv("Acrobot-v1", n + 2, seed=SEED, auto_reset=True, launch_policy={"vec": 2}) as env:
        with pytest.raises(ValueError, match="would not take effect"):
            env.SetLaunchPolicy(sequential_lanes=4)                              # not whole 2 * 4 * 256-lane groups
        assert env.KernelName() == "step_kernel<Acrobot,2,true,false,15,0>"
