"""GPU parity tests for the CartPole hot path — every call goes through the C ABI (ctypes ->
libgymnet_amd.so -> HIP kernels) and is checked against the oracle / the committed golden vectors.

Bars (north_star): done flags, rewards, step counts bit-exact; float32 state within 1e-5 abs of the
float64 restatement per teacher-forced step.  Shapes of the loops follow the reference's own test
(tests/Gym.Tests/Envs/Classic/CartpoleEnvironment.cs:14-35) and README example (README.md:32-52).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-5          # north_star: 1e-5 abs on float32 state
SEED = 0x5EED


def _states(rng, n, wide=False):
    k = 1.15 if wide else 1.0
    return np.stack([rng.uniform(-2.4 * k, 2.4 * k, n), rng.uniform(-2.5, 2.5, n),
                     rng.uniform(-0.2095 * k, 0.2095 * k, n), rng.uniform(-3, 3, n)]).astype(np.float32)


def _near_threshold(ns64, margin=2e-6):
    """lanes whose float64 next state sits within `margin` of a termination threshold: the only place
    where a float32 kernel may legitimately disagree with the float64 reference on `done`."""
    xt, tt = np.float32(2.4).astype(np.float64), np.float32(0.20943951606750488).astype(np.float64)
    return (np.abs(np.abs(ns64[0]) - xt) < margin) | (np.abs(np.abs(ns64[2]) - tt) < margin)


def test_teacher_forced_golden(gpu_pkg, golden):
    g = golden("cartpole_teacher_forced")
    n = g["state"].shape[1]
    with gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED) as env:
        env.Reset()
        env.SetState(g["state"])
        out = env.Step(g["action"])
        got = env.GetState().astype(np.float64)
    want = g["next_state"]
    err = np.abs(got - want)
    assert err[:, :3072].max() <= TOL                                   # in-range block: absolute bar
    assert (err / np.maximum(1.0, np.abs(want))).max() <= TOL           # wide block: values up to ~1e2
    near = _near_threshold(want)
    assert near.sum() == 0                                              # fixture has no ambiguous lane
    assert np.array_equal(out.Done, g["done"].astype(bool))
    assert np.array_equal(out.Reward, g["reward"])
    assert np.array_equal(out.Observation, got.T.astype(np.float32))    # observation IS the new state (CartPoleEnv.cs:166,185)
    assert out.Observation.dtype == np.float32


def test_threshold_edges_bit_exact(gpu_pkg, golden):
    g = golden("cartpole_edges")
    n = g["state"].shape[1]
    with gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED) as env:
        env.Reset()
        env.SetState(g["state"])
        out = env.Step(g["action"])
        got = env.GetState()
    assert np.array_equal(out.Done, g["done"].astype(bool))              # strict < / > on float32 thresholds
    fin = np.isfinite(g["next_state"]).all(axis=0)
    assert np.abs(got[:, fin].astype(np.float64) - g["next_state"][:, fin]).max() <= TOL
    assert np.isnan(got[:, ~fin]).any(axis=0).all() or np.isinf(got[:, ~fin]).any(axis=0).all()


def test_steps_beyond_done_reward_stream(gpu_pkg, golden):
    # reward 1,...,1,1(done),0,0,... and a counted (not printed) step-after-done warning (CartPoleEnv.cs:168-183)
    g = golden("cartpole_steps_beyond_done")
    with gpu_pkg.VectorEnv("CartPole-v1", 1, seed=SEED) as env:
        env.Reset()
        env.SetState(g["start"].reshape(4, 1))
        for t in range(g["reward"].shape[0]):
            out = env.Step(1)                                            # IVecEnv.Step(int): scalar broadcast
            assert out.Reward[0] == g["reward"][t] and bool(out.Done[0]) == bool(g["done"][t])
            assert env.GetStepsBeyondDone()[0] == g["sbd"][t]
            assert np.abs(env.GetState()[:, 0].astype(np.float64) - g["states"][t]).max() <= 1e-4   # free-running
        after = int(g["done"].sum()) - 1
        assert env.Counters()["stepped_after_done"] == after


def test_reference_test_loop_shape(gpu_pkg, golden):
    """1000 x (Reset-if-done else Step(i % 2)) — CartpoleEnvironment.cs:19-27 — on the single-instance Env
    façade, with the reset states taken from the recorded list so that RNG is factored out."""
    g = golden("cartpole_reference_test_trace")
    cp = gpu_pkg.CartPoleEnv(seed=SEED)
    try:
        done, k, lens, cur = True, 0, [], 0
        for i in range(1000):
            if done:
                cp.Reset()
                cp._v.SetState(g["resets"][k].reshape(4, 1)); k += 1
                done = False
                if cur:
                    lens.append(cur)
                cur = 0
            else:
                observation, reward, _done, information = cp.Step(i % 2)      # Step.Deconstruct
                done = _done; cur += 1
                assert reward == 1.0 and information is None
                assert np.abs(observation.astype(np.float64) - g["it_state"][i]).max() <= 1e-4   # free-running episode
            assert int(done) == g["it_done"][i], i                             # integer: exact
        assert k == int(g["resets_used"]) and lens == list(g["episode_lengths"])  # episode step counts exact
    finally:
        cp.CloseEnvironment()


def test_teacher_forced_random_rollout_vs_oracle(gpu_pkg, oracle):
    """Teacher-forced single steps along a real rollout (the SURVEY F9 protocol): each step starts from the
    engine's own float32 state, the float64 restatement is applied to that same state."""
    n, steps = 1 << 16, 40
    rng = np.random.default_rng(5)
    worst, mism, dones = 0.0, 0, 0
    with gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED) as env:
        env.Reset()
        for t in range(steps):
            s = env.GetState()
            a = rng.integers(0, 2, n).astype(np.int32)
            out = env.Step(a)
            want_s, want_r, want_d, _ = oracle.cartpole_step(s.astype(np.float64), a)
            got = env.GetState().astype(np.float64)
            worst = max(worst, np.abs(got - want_s).max())
            mism += int((out.Done != want_d.astype(bool)).sum()); dones += int(want_d.sum())
            assert np.array_equal(out.Reward, want_r)
            env.ResetWhere()                                                    # the caller's `if (done) Reset()`
    assert worst <= TOL and worst < 2e-6
    # the integer done flag is the reference's on EVERY lane — no near-threshold exemption: the kernel derives it from the
    # same float64 sums the reference compares (CartPoleEnv.cs:154,156,167)
    assert mism == 0 and dones > n                                               # ~4.5 % of lanes finish per step


def test_fused_autoreset_matches_oracle_philox(gpu_pkg, oracle):
    n, steps, off = 50_000, 25, 123_456_789_000                                  # lane ids above 2^32 too
    rng = np.random.default_rng(6)
    with gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED, auto_reset=True, lane_offset=off) as env:
        first = env.Reset()
        assert np.array_equal(first.T, oracle.cartpole_reset(SEED, off, 0, n))   # bit-exact Philox reset
        assert first.min() >= -0.05 and first.max() < 0.05                      # U(-0.05, 0.05), CartPoleEnv.cs:65
        total = 0
        for t in range(steps):
            s = env.GetState()
            tick = env.Tick
            a = rng.integers(0, 2, n).astype(np.int32)
            out = env.Step(a)
            want_s, _, want_d, _ = oracle.cartpole_step(s.astype(np.float64), a)
            d = want_d.astype(bool)
            got = env.GetState()
            assert np.array_equal(out.Done, d)
            assert np.all(out.Reward == 1.0)                                    # sbd is always -1 at entry
            assert np.abs(got[:, ~d].astype(np.float64) - want_s[:, ~d]).max() <= TOL
            fresh = oracle.cartpole_reset(SEED, off, tick, n)
            assert np.array_equal(got[:, d], fresh[:, d])                       # reset lanes: oracle's draw, bitwise
            assert np.array_equal(out.Observation, got.T)                       # obs of a finished lane = next episode's first obs
            total += int(d.sum())
        assert total > 0 and env.Tick == steps + 1
        assert env.Counters()["lane_steps"] == steps * n


@pytest.mark.parametrize("n", [1, 3, 4, 63, 65, 257, 1000])
def test_ragged_batch_sizes(gpu_pkg, oracle, n):
    rng = np.random.default_rng(n)
    s = _states(rng, n, wide=True)
    a = rng.integers(0, 2, n).astype(np.int32)
    want_s, want_r, want_d, _ = oracle.cartpole_step(s.astype(np.float64), a)
    for auto in (False, True):
        with gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED, auto_reset=auto) as env:
            env.Reset(); env.SetState(s)
            out = env.Step(a)
            got = env.GetState()
            d = want_d.astype(bool)
            assert np.array_equal(out.Done, d) and out.Observation.shape == (n, 4)
            keep = ~d if auto else np.ones(n, bool)
            if keep.any():
                assert np.abs(got[:, keep].astype(np.float64) - want_s[:, keep]).max() <= TOL


def test_mirror_symmetry_bitwise_at_full_batch(gpu_pkg):
    # size-independent property at BASELINE's full batch (2^20): step(-s, a=0) == -step(s, a=1), exactly
    n = 1 << 20
    rng = np.random.default_rng(7)
    s = _states(rng, n, wide=True)
    a = rng.integers(0, 2, n).astype(np.int32)
    with gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED) as e1, gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED) as e2:
        e1.Reset(); e2.Reset()
        e1.SetState(s); e2.SetState(-s)
        o1 = e1.Step(a); o2 = e2.Step(1 - a)
        assert np.array_equal(e1.GetState(), -e2.GetState())
        assert np.array_equal(o1.Done, o2.Done) and 0 < o1.Done.sum() < n


@pytest.mark.parametrize("force_graph", ["1", "0"])
def test_graph_rollout_equals_eager_steps_bitwise(gpu_pkg, monkeypatch, force_graph):
    # hipGraph replay (frozen kernel arguments, device-side tick) must give the eager result, bit for bit.
    # The library picks graph vs eager launches by batch size; gymnet_launch_policy.graph forces each path at full size.
    import torch
    n, ring, steps = 1 << 20, 8, 8 * 5 + 3
    dev = torch.device("cuda", 0)
    with gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED, auto_reset=True, launch_policy={"graph": int(force_graph)}) as g, \
            gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED, auto_reset=True) as e:
        assert g.GetLaunchPolicy()["graph"] == int(force_graph) and e.GetLaunchPolicy()["graph"] == -1
        acts = torch.empty((ring, n), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        for t in range(ring):
            g.SampleActionsDevice(acts[t], seed=SEED + 1, tick=t)
        g.Sync()
        a_host = acts.cpu().numpy()
        assert set(np.unique(a_host)) == {0, 1} and abs(a_host.mean() - 0.5) < 1e-3
        g.ResetDevice(); e.ResetDevice()
        g.RolloutDevice(acts, steps, n, ring)
        g.RolloutDevice(acts, ring * 2, n, ring)         # second call replays the cached graph
        for t in range(steps):
            e.StepDevice(acts[t % ring])
        for t in range(ring * 2):
            e.StepDevice(acts[t % ring])
        g.Sync(); e.Sync()
        assert g.Tick == e.Tick == 1 + steps + 2 * ring
        assert np.array_equal(g.GetState(), e.GetState())
        assert g.Counters()["tick"] == g.Tick            # the device-side tick agrees with the host mirror
        r1, r2 = g.Read(), e.Read()
        assert np.array_equal(r1.Done, r2.Done) and np.array_equal(r1.Reward, r2.Reward)
        st = g.GetState()
        assert np.isfinite(st).all() and np.abs(st[0]).max() < 2.6 and np.abs(st[2]).max() < 0.3   # auto-reset keeps lanes in range


def test_sharding_invariance_bitwise(gpu_pkg):
    # G logical shards with global lane offsets == one big batch, incl. Philox resets
    n, G, steps = 1 << 16, 4, 30
    rng = np.random.default_rng(8)
    acts = rng.integers(0, 2, (steps, n)).astype(np.int32)
    with gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED, auto_reset=True) as whole:
        whole.Reset()
        for t in range(steps):
            whole.Step(acts[t])
        ref = whole.GetState()
    parts = []
    for r in range(G):
        lo, hi = r * n // G, (r + 1) * n // G
        with gpu_pkg.VectorEnv("CartPole-v1", hi - lo, seed=SEED, auto_reset=True, lane_offset=lo) as sh:
            sh.Reset()
            for t in range(steps):
                sh.Step(acts[t, lo:hi])
            parts.append(sh.GetState())
    assert np.array_equal(np.concatenate(parts, axis=1), ref)


@pytest.mark.parametrize("with_list,vec", [(True, 4), (True, 1), (False, 4)])
def test_done_list_episode_stats_final_obs(gpu_pkg, oracle, with_list, vec, monkeypatch):
    """Episode bookkeeping.  With the done list the step kernel writes COMPACT records — (lane, return, length,
    terminal observation) at the lane's position in the list — checked per step against the host-side bookkeeping of
    BasePlaySession.cs:58-69.  With or without the list the kernel maintains the dense per-lane arrays itself (round 4; the
    compact-records-only behaviour of round 3 is an opt-in, tested in test_dense_episode_views_stay_current_...), so they are
    complete whether read every step or only at the end."""
    n, steps = 20_000, 60
    rng = np.random.default_rng(9)
    ret = np.zeros(n, np.float32); ln = np.zeros(n, np.int32)
    fin_ret = np.zeros(n, np.float32); fin_len = np.zeros(n, np.int32)
    fin_obs = np.zeros((n, 4), np.float32)
    # vec 4: the dwordx4 bookkeeping kernel a 2^20-lane batch runs (wave-compacted reset)
    with gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED, auto_reset=True, done_list=with_list, episode_stats=True,
                           final_obs=True, launch_policy={"vec": vec, "reset_form": 1 if vec == 4 else 0}) as env:
        assert env.KernelName() == f"step_kernel<CartPole,{vec},true,true,15,{1 if vec == 4 else 0}>"
        env.Reset()
        for t in range(steps):
            s = env.GetState()
            a = rng.integers(0, 2, n).astype(np.int32)
            out = env.Step(a)
            term = oracle.cartpole_step(s, a, dtype=np.float32)[0]                  # terminal obs, kernel semantics
            ret += out.Reward; ln += 1
            d = out.Done
            fin_ret[d] = ret[d]; fin_len[d] = ln[d]
            fin_obs[d] = term.T[d]
            if with_list:
                lanes = env.DoneLanes()
                assert sorted(lanes.tolist()) == np.nonzero(d)[0].tolist()           # wave-ballot compaction == mask
                rec = env.DoneRecords()
                order = np.argsort(rec["lanes"])
                assert np.array_equal(rec["lanes"][order], np.nonzero(d)[0])
                assert np.array_equal(rec["length"][order], ln[d]) and np.array_equal(rec["return"][order], ret[d])   # integer counts exact
                assert np.array_equal(rec["final_obs"][order], term.T[d])
                if t % 20 == 0:                                                     # device-side gathers of the 256 shards
                    import torch
                    d_l = torch.full((n,), -1, dtype=torch.int32, device="cuda"); d_c = torch.zeros(1, dtype=torch.int32, device="cuda")
                    d_r = torch.zeros(n, dtype=torch.float32, device="cuda"); d_n = torch.zeros(n, dtype=torch.int32, device="cuda")
                    d_o = torch.zeros((n, 4), dtype=torch.float32, device="cuda")
                    torch.cuda.synchronize()
                    env.DoneLanesDevice(d_l, d_c); env.Sync()
                    c = int(d_c.item())
                    assert c == len(lanes) and sorted(d_l[:c].cpu().tolist()) == sorted(lanes.tolist())
                    env.DoneRecordsDevice(d_l, d_r, d_n, d_o, n, d_c); env.Sync()
                    assert int(d_c.item()) == c
                    o2 = np.argsort(d_l[:c].cpu().numpy())
                    assert np.array_equal(d_n[:c].cpu().numpy()[o2], ln[d]) and np.array_equal(d_o[:c].cpu().numpy()[o2], term.T[d])
                    env.DoneRecordsDevice(d_l, None, None, None, 3, d_c); env.Sync()      # a short buffer: true count, first records only
                    assert int(d_c.item()) == c
                assert env.Counters()["last_done_count"] == int(d.sum())
                # (no dense-view getter inside the loop: the kernel keeps those arrays current without the caller's help — ADVICE r3)
            ret[d] = 0; ln[d] = 0
        got_ret, got_len = env.EpisodeStats()
        assert np.array_equal(got_len, fin_len) and np.array_equal(got_ret, fin_ret)   # integer step counts exact
        assert np.array_equal(env.FinalObs(), fin_obs)
        assert 15 < fin_len[fin_len > 0].mean() < 30                                  # mean 22.25


def test_time_limit_extension(gpu_pkg):
    n = 512
    with gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED, auto_reset=True, episode_stats=True, max_episode_steps=5) as env:
        env.Reset()
        longest = 0
        for t in range(12):
            out = env.Step(t % 2)
            _, ln = env.EpisodeStats()
            longest = max(longest, int(ln.max()))
        assert longest == 5            # no episode outlives the limit; the reference itself has none
    with pytest.raises(ValueError):
        gpu_pkg.VectorEnv("CartPole-v1", n, max_episode_steps=5)        # needs episode_stats


def test_seed_semantics(gpu_pkg):
    n = 4096
    with gpu_pkg.VectorEnv("CartPole-v1", n, seed=1) as a, gpu_pkg.VectorEnv("CartPole-v1", n, seed=2) as b:
        ra, rb = a.Reset(), b.Reset()
        assert not np.array_equal(ra, rb)
        b.Seed(1)                                                         # Env.Seed(int): same seed, same stream
        assert np.array_equal(b.Reset(), ra)
        assert len(np.unique(ra[:, 0])) > n * 0.99                        # lanes draw different values (documented deviation)
        with pytest.raises(ValueError, match="Number of seeds"):          # VecEnv.cs:49 ArgumentException
            a.Seed([1, 2, 3])
        a.Seed(list(range(n)))                                            # VecEnv.Seed(int[])
        r1 = a.Reset()
        a.Seed(list(range(n)))
        assert np.array_equal(a.Reset(), r1) and not np.array_equal(r1, ra)


def test_invalid_actions(gpu_pkg, oracle):
    n = 1000
    s = _states(np.random.default_rng(10), n)
    acts = np.zeros(n, np.int32); acts[17] = 2; acts[900] = -1
    # default = Release-build CartPole: anything != 1 pushes left (CartPoleEnv.cs:139,146)
    with gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED) as env:
        env.Reset(); env.SetState(s)
        env.Step(acts)
        left = oracle.cartpole_step(s.astype(np.float64), np.zeros(n, np.int32))[0]
        assert np.abs(env.GetState().astype(np.float64) - left).max() <= TOL
    # validate_actions = LunarLander-style throw before any state change (LunarLanderEnv.cs:604-607)
    with gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED, validate_actions=True) as env:
        env.Reset(); env.SetState(s)
        with pytest.raises(gpu_pkg.InvalidActionError, match="outside of the configured action space"):
            env.Step(acts)
        with pytest.raises(gpu_pkg.InvalidActionError):
            env.Step(2)
        assert np.array_equal(env.GetState(), s)
        env.Step(np.ones(n, np.int32))
        import torch                                                      # device rollouts validate every slice up front
        acts_dev = torch.ones((4, n), dtype=torch.int32, device="cuda")
        acts_dev[2, 5] = 7
        torch.cuda.synchronize()                                          # torch's stream is not the engine's stream
        before = env.GetState()
        with pytest.raises(gpu_pkg.InvalidActionError):
            env.RolloutDevice(acts_dev, 8, n, 4)
        assert np.array_equal(env.GetState(), before)
        env.RolloutDevice(acts_dev, 2, n, 4)                              # slices 0 and 1 only: valid
        env.Sync()
    cp = gpu_pkg.CartPoleEnv()
    with pytest.raises(TypeError):                                       # (int)action InvalidCastException, CartPoleEnv.cs:138
        cp.Step(0.5)
    cp.Close()


def test_step_async_and_errors(gpu_pkg):
    n = 2048
    with gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED) as env, gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED) as ref:
        env.Reset(); ref.Reset()
        with pytest.raises(gpu_pkg.NotSteppingError):
            env.StepWait()
        task = env.StepAsync(1)                                           # VecEnv.StepAsync(int)
        with pytest.raises(gpu_pkg.AlreadySteppingError):
            env.StepAsync(0)
        with pytest.raises(gpu_pkg.AlreadySteppingError):
            env.Step(0)
        got = task.Result()
        want = ref.Step(1)
        assert np.array_equal(got.Observation, want.Observation) and np.array_equal(got.Done, want.Done)
        assert isinstance(got[0], gpu_pkg.Step) and len(got.ToSteps()) == n
    with pytest.raises(ValueError):
        gpu_pkg.VectorEnv("CartPole-v1", 0)
    with pytest.raises(ValueError):
        gpu_pkg.VectorEnv("CartPole-v1", 16, device=99)


def test_external_obs_buffer_and_unaligned_fallback(gpu_pkg):
    # state living inside a caller-provided buffer (the all-gather layout), aligned (dwordx4 path) and
    # deliberately misaligned (scalar fallback): identical results
    import torch
    n, steps = 4096 + 8, 12
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(11)
    acts = rng.integers(0, 2, (steps, n)).astype(np.int32)
    with gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED, auto_reset=True) as own:
        own.Reset()
        for t in range(steps):
            own.Step(acts[t])
        ref = own.GetState()
    for shift in (0, 1):
        buf = torch.zeros(4 * n + 8, dtype=torch.float32, device=dev)
        view = buf[shift:shift + 4 * n]
        torch.cuda.synchronize()                                          # torch's fill must land before the engine owns the buffer
        with gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED, auto_reset=True, ext_obs=view.data_ptr(), ext_obs_stride=n) as env:
            env.Reset()
            for t in range(steps):
                env.Step(acts[t])
            env.Sync()
            assert np.array_equal(env.GetState(), ref)
            assert np.array_equal(view.view(4, n).cpu().numpy(), ref)     # zero-copy: the buffer IS the state


def test_batched_space_sampling(gpu_pkg, oracle):
    n = 10_000
    with gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED, lane_offset=77) as env:
        a = env.SampleActions(seed=9, tick=4)
        assert np.array_equal(a, oracle.discrete_sample(9, 77, 4, 2, 0, n))   # Discrete.Sample(): start + randint(0, N)
        assert all(env.ActionSpace.Contains(int(x)) for x in a[:100])
    with gpu_pkg.VectorEnv("Pendulum-v1", n, seed=SEED) as env:
        a = env.SampleActions(seed=9, tick=4)
        assert np.array_equal(a, oracle.box_uniform_sample(9, 0, 4, -2.0, 2.0, n))   # Box.Sample(), bounded regime
        assert a.min() >= -2.0 and a.max() <= 2.0                                   # BoxTest.cs:36-41


def test_launch_policy_variants_agree_bitwise(gpu_pkg, monkeypatch):
    # the launch policy (scalar vs dwordx4 lanes, which streams are non-temporal) is a performance choice
    # made from the batch size; every variant must compute the same bits
    n, steps = 8192 + 5, 20
    rng = np.random.default_rng(13)
    acts = rng.integers(0, 2, (steps, n)).astype(np.int32)
    results = []
    # (lanes per thread, non-temporal mask, reset form: 0 = per-thread drain loop, 1 = wave-compacted through LDS)
    for vec, nt, rf in ((1, 0, 0), (1, 12, 0), (1, 15, 0), (4, 0, 0), (4, 12, 0), (4, 15, 0), (4, 0, 1), (4, 12, 1), (4, 15, 1)):
        pol = {"vec": vec, "nt": nt, "reset_form": rf}
        for auto in (True, False):
            with gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED, auto_reset=auto, launch_policy=pol) as env:
                assert env.KernelName() == f"step_kernel<CartPole,{vec},{str(auto).lower()},false,{nt},{rf if auto else 0}>"
                with gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED, auto_reset=auto, done_list=True, episode_stats=True, launch_policy=pol) as ex:
                    assert ex.KernelName() == f"step_kernel<CartPole,{vec},{str(auto).lower()},true,{nt},{rf if auto else 0}>"
                env.Reset()
                dones = 0
                for t in range(steps):
                    dones += int(env.Step(acts[t]).Done.sum())
                results.append((auto, env.GetState(), dones))
    for auto in (True, False):
        same = [r for r in results if r[0] == auto]
        assert len(same) == 9 and same[0][2] > 0
        for r in same[1:]:
            assert np.array_equal(r[1], same[0][1], equal_nan=True) and r[2] == same[0][2]


def test_kernel_is_bit_identical_to_the_float32_restatement(gpu_pkg, oracle):
    """Beyond the 1e-5 bar: the kernel's float32 arithmetic (IEEE mul/add/fma/div, its own Cody-Waite sin/cos,
    constant division as fma pairs) is restated operation for operation in the oracle's "kernel semantics"
    functions, so states, rewards and done flags must agree BIT FOR BIT — at the full 2^20 batch."""
    n = 1 << 20
    rng = np.random.default_rng(21)
    s = _states(rng, n, wide=True)
    s[2, ::7] = rng.uniform(-30, 30, s[2, ::7].shape).astype(np.float32)       # large angles: all four quadrants
    s[3, ::5] = rng.uniform(-40, 40, s[3, ::5].shape).astype(np.float32)
    a = rng.integers(0, 2, n).astype(np.int32)
    want_s, want_r, want_d, want_b = oracle.cartpole_step(s, a, dtype=np.float32)
    with gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED) as env:
        env.Reset(); env.SetState(s)
        out = env.Step(a)
        assert np.array_equal(env.GetState(), want_s)
        assert np.array_equal(out.Done, want_d.astype(bool)) and np.array_equal(out.Reward, want_r)
        assert np.array_equal(env.GetStepsBeyondDone(), want_b)
        # and a free-running rollout stays bit-identical step after step (no drift between the two float32 paths)
        cur = want_s
        sbd = want_b
        for t in range(10):
            a = rng.integers(0, 2, n).astype(np.int32)
            cur, r, d, sbd = oracle.cartpole_step(cur, a, sbd, dtype=np.float32)
            out = env.Step(a)
            assert np.array_equal(out.Reward, r) and np.array_equal(out.Done, d.astype(bool))
        assert np.array_equal(env.GetState(), cur, equal_nan=True)


@pytest.mark.parametrize("name,auto,n", [("CartPole-v1", True, 1 << 16), ("CartPole-v1", False, 5000), ("CartPole-v1", True, 1003),
                                         ("Pendulum-v1", True, 4096), ("MountainCar-v0", True, 4096), ("Acrobot-v1", True, 4096)])
def test_fused_rollout_equals_stepwise_and_records(gpu_pkg, name, auto, n):
    """T steps fused into one launch (state in registers) + device-side rollout buffers must be
    bit-identical to T one-step launches — final state, every recorded observation / reward / done."""
    import torch
    dev = torch.device("cuda", 0)
    T, ring = 37, 8
    with gpu_pkg.VectorEnv(name, n, seed=SEED, auto_reset=auto) as f, gpu_pkg.VectorEnv(name, n, seed=SEED, auto_reset=auto) as e:
        D = f.ObsDim
        adt = torch.float32 if name == "Pendulum-v1" else torch.int32
        stride = (n + 3) // 4 * 4
        acts = torch.zeros((ring, stride), dtype=adt, device=dev)
        torch.cuda.synchronize()
        for t in range(ring):
            f.SampleActionsDevice(acts[t], seed=SEED + 1, tick=t)
        f.Sync()
        rec_obs = torch.zeros((T, D, n), dtype=torch.float32, device=dev)
        rec_rew = torch.zeros((T, n), dtype=torch.float32, device=dev)
        rec_done = torch.zeros((T, n), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        f.ResetDevice(); e.ResetDevice()
        f.RolloutFusedDevice(acts, T, stride, ring, rec_obs, rec_rew, rec_done)
        f.Sync()
        a_host = acts.cpu().numpy()
        for t in range(T):
            out = e.Step(a_host[t % ring, :n])
            assert np.array_equal(rec_obs[t].cpu().numpy().T, out.Observation), t
            assert np.array_equal(rec_rew[t].cpu().numpy(), out.Reward), t
            assert np.array_equal(rec_done[t].cpu().numpy().astype(bool), out.Done), t
        assert np.array_equal(f.GetState(), e.GetState(), equal_nan=True)
        assert f.Tick == e.Tick == T + 1
        last = f.Read()
        assert np.array_equal(last.Reward, out.Reward) and np.array_equal(last.Done, out.Done)
        assert f.Counters()["lane_steps"] == T * n and f.Counters()["tick"] == T + 1
        # a following ordinary step continues from the same tick on both
        f.Step(a_host[0, :n]); e.Step(a_host[0, :n])
        assert np.array_equal(f.GetState(), e.GetState(), equal_nan=True)
    # bookkeeping handles are fused too since ABI 5 (tests/test_gpu_fused_rollout_ex.py); here: the done list of the LAST step
    with gpu_pkg.VectorEnv(name, 64, seed=SEED, auto_reset=True, done_list=True) as x, gpu_pkg.VectorEnv(name, 64, seed=SEED, auto_reset=True, done_list=True) as y:
        x.ResetDevice(); y.ResetDevice()
        x.RolloutFusedDevice(acts, 9, stride, ring)
        for t in range(9):
            y.StepDevice(acts[t % ring])
        assert np.array_equal(np.sort(x.DoneLanes()), np.sort(y.DoneLanes())) and np.array_equal(x.GetState(), y.GetState(), equal_nan=True)


def test_maximum_size_batch_properties(gpu_pkg, oracle):
    """2^25 lanes (the largest size the CPU side of this test can afford; the bench goes to 2^27): reset draws are
    U(-0.05, 0.05) with the oracle's exact values at both ends of the lane range, a rollout keeps every lane inside
    the thresholds it resets at, and the step count adds up."""
    n = 1 << 25
    with gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED, auto_reset=True) as env:
        obs = env.Reset()
        assert obs.shape == (n, 4) and obs.min() >= -0.05 and obs.max() < 0.05
        assert abs(float(obs.mean())) < 1e-4 and abs(float(obs.std()) - 0.1 / np.sqrt(12)) < 1e-4
        assert np.array_equal(obs[:64].T, oracle.cartpole_reset(SEED, 0, 0, 64))
        assert np.array_equal(obs[-64:].T, oracle.cartpole_reset(SEED, n - 64, 0, 64))
        del obs
        dones = 0
        for t in range(12):
            out = env.Step(1)                                         # IVecEnv.Step(int): scalar broadcast, always push right
            dones += int(out.Done.sum())
        st = env.GetState()
        assert np.isfinite(st).all() and np.abs(st[0]).max() <= 2.5 and np.abs(st[2]).max() <= 0.3
        c = env.Counters()
        assert c["lane_steps"] == 12 * n and c["tick"] == 13 and dones > 0


def test_free_running_autoreset_rollout_is_bit_identical_to_the_cpu_restatement(gpu_pkg, oracle):
    """300 free-running steps with fused auto-reset on 2^16 lanes — device-sampled actions, Philox resets keyed by
    (global lane, tick), one-launch steps and the fused T-step kernel mixed — replayed on the CPU with the oracle's
    float32 kernel-semantics step + its Philox reset: every state bit, every done flag must agree."""
    import torch
    n, ring, off = 1 << 16, 16, 7_000_000
    with gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED, auto_reset=True, lane_offset=off) as env:
        acts = torch.empty((ring, n), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        for t in range(ring):
            env.SampleActionsDevice(acts[t], seed=77, tick=t)
        env.Sync()
        a_host = acts.cpu().numpy()
        for t in range(ring):                                             # the device sampler == the oracle's sampler
            assert np.array_equal(a_host[t], oracle.discrete_sample(77, off, t, 2, 0, n))
        env.ResetDevice()
        env.RolloutDevice(acts, 100, n, ring)                             # ticks 1..100
        env.RolloutFusedDevice(acts, 150, n, ring)                        # ticks 101..250 (slices restart at 0)
        env.RolloutDevice(acts, 50, n, ring)                              # ticks 251..300
        env.Sync()
        got, last = env.GetState(), env.Read()

    s = oracle.cartpole_reset(SEED, off, 0, n)
    tick, total_done = 1, 0
    for seg in (100, 150, 50):
        for t in range(seg):
            s, r, d, _ = oracle.cartpole_step(s, a_host[t % ring], dtype=np.float32)
            fin = d.astype(bool)
            fresh = oracle.cartpole_reset(SEED, off, tick, n)
            s[:, fin] = fresh[:, fin]
            tick += 1; total_done += int(fin.sum())
    assert np.array_equal(got, s)
    assert np.array_equal(last.Done, fin) and np.array_equal(last.Reward, r)
    assert total_done > 10 * n                                            # ~13 episodes per lane on average


@pytest.mark.parametrize("name,auto", [("CartPole-v1", True), ("CartPole-v1", False), ("Acrobot-v1", True)])
def test_checkpoint_resume_is_bit_exact(gpu_pkg, name, auto):
    """Checkpoint() = state + engine tick (+ steps_beyond_done): restoring it into a FRESH handle with the same seed and
    replaying the same actions reproduces the continuation bit for bit — states, rewards, done flags, and the Philox reset
    draws of lanes that finish after the checkpoint."""
    n = 5000
    rng = np.random.default_rng(21)
    nact = 2 if name == "CartPole-v1" else 3
    acts = rng.integers(0, nact, (90, n)).astype(np.int32)
    with gpu_pkg.VectorEnv(name, n, seed=SEED, auto_reset=auto) as a:
        a.Reset()
        for t in range(40):
            a.Step(acts[t])
        ck = a.Checkpoint()
        tail_a = [a.Step(acts[t]) for t in range(40, 90)]
        end_a = a.GetState()
    with gpu_pkg.VectorEnv(name, n, seed=SEED, auto_reset=auto) as b:
        b.Reset()
        b.Restore(ck)
        tail_b = [b.Step(acts[t]) for t in range(40, 90)]
        end_b = b.GetState()
        with pytest.raises(ValueError):
            b.Restore(dict(ck, num_envs=n + 1))
    assert np.array_equal(end_a, end_b, equal_nan=True)
    for x, y in zip(tail_a, tail_b):
        assert np.array_equal(x.Observation, y.Observation) and np.array_equal(x.Reward, y.Reward) and np.array_equal(x.Done, y.Done)
    assert name != "CartPole-v1" or any(x.Done.any() for x in tail_a)       # random-action Acrobot does not finish in 90 steps


@pytest.mark.parametrize("name,kw", [
    ("CartPole-v1", dict(auto_reset=True, episode_stats=True, max_episode_steps=13, done_list=True, final_obs=True)),
    ("CartPole-v1", dict(auto_reset=False, episode_stats=True, max_episode_steps=13)),
    ("Acrobot-v1", dict(auto_reset=True, episode_stats=True, max_episode_steps=7, final_obs=True)),
    ("Pendulum-v1", dict(auto_reset=True, episode_stats=True, max_episode_steps=9, done_list=True)),
    ("CartPole-v1", dict(auto_reset=True, episode_stats=True, max_episode_steps=11, dtype=np.float64)),
])
def test_checkpoint_of_every_configuration_is_bit_exact(gpu_pkg, name, kw):
    """VERDICT r3 #5 / SURVEY §5: Checkpoint() used to refuse episode_stats / max_episode_steps / Seed(int[]) handles.  It now
    carries every array such a handle keeps (gymnet_vecenv_get_array): running episode return / length — which decide WHEN the
    time limit truncates — the done flags ResetWhere(None) consumes, the dense finished-episode views, terminal observations,
    the per-lane Philox keys.  Restore() into a fresh handle, mid-episode, then the same actions: truncation steps, reset
    draws, rewards, episode records and the dense views all continue bit for bit."""
    n = 3000 + 5
    rng = np.random.default_rng(31)
    box = name == "Pendulum-v1"
    nact = 2 if name == "CartPole-v1" else 3
    acts = rng.uniform(-2, 2, (60, n)).astype(np.float32) if box else rng.integers(0, nact, (60, n)).astype(np.int32)
    seeds = (np.arange(n, dtype=np.int64) * 2654435761) % (1 << 40) + 5
    auto = kw["auto_reset"]

    def drive(env, t0, t1):
        out = []
        for t in range(t0, t1):
            o = env.Step(acts[t])
            rec = env.DoneRecords() if kw.get("done_list") else None
            out.append((o.Observation.copy(), o.Reward.copy(), o.Done.copy(), o.Truncated.copy(), rec))
            if not auto:
                env.ResetWhere()                                   # the caller's `if (done) Reset()`: consumes the done flags
        return out

    with gpu_pkg.VectorEnv(name, n, seed=SEED, **kw) as a:
        a.Seed(seeds)                                              # per-lane Philox keys (VecEnv.Seed(int[]), VecEnv.cs:48-53)
        a.Reset()
        drive(a, 0, 25)
        if not auto:
            a.Step(acts[25])                                       # checkpoint BETWEEN a step and its ResetWhere(): done flags pending
        ck = a.Checkpoint()
        assert {"episode_return", "episode_length", "finished_return", "finished_length", "done", "reward", "lane_seeds"} <= set(ck["arrays"])
        assert ("final_obs" in ck["arrays"]) == bool(kw.get("final_obs")) and ("steps_beyond_done" in ck["arrays"]) == (name == "CartPole-v1" and not auto)
        assert 0 < ck["arrays"]["episode_length"].max() < kw["max_episode_steps"] + 1 and ck["arrays"]["episode_length"].min() >= 0   # mid-episode
        if not auto:
            a.ResetWhere()
        tail_a = drive(a, 26 if not auto else 25, 60)
        end_a = (a.GetState(), a.EpisodeStats(), a.FinalObs() if kw.get("final_obs") else None, a.Tick)
    with gpu_pkg.VectorEnv(name, n, seed=999, **kw) as b:          # a different seed: everything must come from the checkpoint
        b.Reset()
        b.Restore(ck)
        assert b.GetSeed() == (SEED, True) and b.Tick == ck["tick"]          # Seed(int[]) keeps the scalar key; per-lane keys active
        if not auto:
            assert np.array_equal(b.Read().Done, ck["arrays"]["done"].astype(bool))
            b.ResetWhere()
        tail_b = drive(b, 26 if not auto else 25, 60)
        end_b = (b.GetState(), b.EpisodeStats(), b.FinalObs() if kw.get("final_obs") else None, b.Tick)
        with pytest.raises(ValueError):
            b.Restore(dict(ck, dtype="float64" if ck["dtype"] == "float32" else "float32"))
    assert np.array_equal(end_a[0], end_b[0], equal_nan=True) and end_a[3] == end_b[3]
    assert np.array_equal(end_a[1][0], end_b[1][0]) and np.array_equal(end_a[1][1], end_b[1][1])
    if end_a[2] is not None:
        assert np.array_equal(end_a[2], end_b[2])
    trunc = 0
    for x, y in zip(tail_a, tail_b):
        assert all(np.array_equal(u, v, equal_nan=True) for u, v in zip(x[:4], y[:4]))
        trunc += int(x[3].sum())
        if x[4] is not None:
            ia, ib = np.argsort(x[4]["lanes"]), np.argsort(y[4]["lanes"])
            for key in x[4]:
                assert np.array_equal(x[4][key][ia], y[4][key][ib]), key
    assert trunc > 0                                               # the time limit fired after the restore, on the same steps
    with gpu_pkg.VectorEnv(name, n, seed=SEED, auto_reset=auto, dtype=kw.get("dtype", np.float32)) as plain:   # a handle without the arrays the checkpoint carries
        plain.Reset()
        with pytest.raises(NotImplementedError):
            plain.Restore(ck)


def test_dense_episode_views_stay_current_without_caller_cooperation(gpu_pkg):
    """with DONE_LIST the step kernel wrote compact records only, and the dense "last finished episode per
    lane" arrays were refreshed by their getters for the most recent step alone — rollout_device(K > 1), graph replays and any
    loop that skipped the getter lost episodes.  The kernel keeps the dense views itself again (compact-only is an explicit
    opt-in): a K-step rollout with done_list + episode_stats + final_obs must leave exactly what a handle without the list
    leaves, read once at the end; and with GYMNET_FLAG_COMPACT_RECORDS_ONLY the documented on-demand behaviour holds."""
    import torch
    n, ring, K = 20_000, 8, 64
    acts = torch.randint(0, 2, (ring, n), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    views = {}
    for label, kw in (("list", dict(done_list=True)), ("nolist", {}), ("compact", dict(done_list=True, compact_records_only=True))):
        for graph in (1, 0):
            with gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED, auto_reset=True, episode_stats=True, final_obs=True,
                                   launch_policy={"graph": graph}, **kw) as env:
                env.ResetDevice()
                env.RolloutDevice(acts, K, n, ring)                 # K steps, no getter in between
                env.Sync()
                dv = env.DeviceView()
                assert dv.d_finished_return and dv.d_finished_length
                ret, ln = env.EpisodeStats()
                views[(label, graph)] = (ret, ln, env.FinalObs(), env.GetArray("finished_length"))
    ref = views[("nolist", 0)]
    assert (ref[1] > 0).mean() > 0.9                                # nearly every lane finished at least one episode in 64 steps
    for key in (("list", 1), ("list", 0), ("nolist", 1)):
        assert all(np.array_equal(u, v) for u, v in zip(ref, views[key])), key
    comp = views[("compact", 0)]
    last_only = comp[1] != ref[1]
    assert last_only.any() and np.array_equal(comp[1][~last_only], ref[1][~last_only])     # opt-in: only the last step's records were applied


def test_launch_policy_through_the_abi(gpu_pkg, monkeypatch):
    """the launch policy is set through gymnet_vecenv_set_launch_policy, and the shipped library no longer reads
    GYMNET_* from the process environment (a host's environment must not change which kernel a library runs)."""
    for var, val in (("GYMNET_VEC", "1"), ("GYMNET_NT", "0"), ("GYMNET_RESET_FORM", "0"), ("GYMNET_BLOCK", "64"), ("GYMNET_GRAPH", "0")):
        monkeypatch.setenv(var, val)
    n = 1 << 20
    with gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED, auto_reset=True) as env:
        assert env.KernelName() == "step_kernel<CartPole,4,true,false,15,1>"            # the environment changed nothing
        assert env.GetLaunchPolicy() == {"vec": 4, "block": 256, "nt": 15, "sequential_lanes": 1, "reset_form": 1, "lds_pipe": 0,
                                         "occupancy_lds_bytes": 0, "graph": -1}
        env.SetLaunchPolicy(vec=1, nt=12)
        assert env.KernelName() == "step_kernel<CartPole,1,true,false,12,0>" and env.LaunchPolicy()["envs_per_thread"] == 1
        env.SetLaunchPolicy(vec=4, reset_form=1, block=128, graph=1)
        assert env.KernelName() == "step_kernel<CartPole,4,true,false,12,1>" and env.GetLaunchPolicy()["block"] == 128
        for bad in (dict(vec=2), dict(vec=3), dict(block=96), dict(nt=7), dict(sequential_lanes=4), dict(lds_pipe=1), dict(graph=5)):
            with pytest.raises(ValueError):
                env.SetLaunchPolicy(**bad)
        assert env.KernelName() == "step_kernel<CartPole,4,true,false,12,1>"             # a refused policy changes nothing
        with pytest.raises(TypeError):
            env.SetLaunchPolicy(lanes=4)
        env.SetLaunchPolicy(graph=-2)
        assert env.GetLaunchPolicy()["graph"] == -1
    with gpu_pkg.VectorEnv("Acrobot-v1", 1 << 20, seed=SEED, auto_reset=True) as env:
        assert env.KernelName() == "step_kernel_pipe<Acrobot,4,true,15>"
        env.SetLaunchPolicy(sequential_lanes=1, vec=2)
        assert env.KernelName() == "step_kernel<Acrobot,2,true,false,15,0>"
        with pytest.raises(ValueError):
            env.SetLaunchPolicy(vec=4)


def test_an_all_equal_seed_vector_is_seed_int_and_keeps_the_lean_kernel(gpu_pkg):
    """VecEnv.Seed(int) reaches a VecEnv-typed C# caller's lanes as N equal seeds (VecEnv.cs:44-46 walks Environments); round 2
    turned that into the per-lane-key kernel variant (slower, no fused rollout) although it computes the same bits.  An
    all-equal vector now IS Seed(int): same kernel, same draws, fused rollout available; a genuinely per-lane vector still
    selects the keyed variant."""
    import torch
    n = 6000
    rng = np.random.default_rng(8)
    acts = rng.integers(0, 2, (30, n)).astype(np.int32)
    with gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED, auto_reset=True) as a, gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED, auto_reset=True) as b:
        lean = a.KernelName()
        assert lean.split(",")[3] == "false"
        a.Seed(7)
        b.Seed([7] * n)                                     # gymnet_vecenv_seed_lanes with N equal keys
        assert b.KernelName() == lean == a.KernelName()
        assert np.array_equal(a.Reset(), b.Reset())
        for t in range(30):
            x, y = a.Step(acts[t]), b.Step(acts[t])
            assert np.array_equal(x.Observation, y.Observation) and np.array_equal(x.Done, y.Done)
        d = torch.from_numpy(acts[:4]).cuda().contiguous()
        b.RolloutFusedDevice(d, 8, n, 4); a.RolloutFusedDevice(d, 8, n, 4)    # GYMNET_ERR_UNSUPPORTED in round 2
        assert np.array_equal(a.GetState(), b.GetState())
        b.Seed(np.arange(n))                                # per-lane keys: the keyed (EXTRAS) variant — fused as well since ABI 5
        assert b.KernelName().split(",")[3] == "true"      # step_kernel<Env, VEC, AUTORESET, EXTRAS, ...>
        b.RolloutFusedDevice(d, 8, n, 4)
        b.Seed(np.full(n, 7))                               # and back
        assert b.KernelName() == lean


@pytest.mark.parametrize("n", [1024 + 5, 64 * 4 * 3 + 1, 7, 64 * 4 + 3])
def test_wave_compacted_reset_when_every_lane_finishes_and_the_last_wave_is_partial(gpu_pkg, oracle, monkeypatch, n):
    """reset_pending_wave hands a wave's finished sub-lanes to the wave's ACTIVE lanes.  Worst cases for that hand-off: every
    sub-lane of every lane finishes in the same step (256 slots per full wave: four rounds of 64), and the batch's last wave
    has only one or two active threads (rounds of one or two slots).  Every lane must come out holding exactly its own Philox
    reset draw, and the compacted form must equal the per-thread drain loop."""
    s = np.zeros((4, n), np.float32)
    s[0] = 2.39; s[1] = 3.0                                        # x' = 2.39 + 0.02 * 3 > x_threshold for every lane
    got = {}
    for rf in (1, 0):
        with gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED, auto_reset=True, lane_offset=12345, launch_policy={"vec": 4, "reset_form": rf}) as env:
            assert env.KernelName() == f"step_kernel<CartPole,4,true,false,15,{rf}>"
            env.Reset(); env.SetState(s)
            tick = env.Tick
            out = env.Step(np.ones(n, np.int32))
            assert out.Done.all()
            got[rf] = env.GetState()
            assert np.array_equal(got[rf], oracle.cartpole_reset(SEED, 12345, tick, n))
            out = env.Step(np.ones(n, np.int32))                      # and the NEXT step sees ordinary states again
            assert not out.Done.any()
    assert np.array_equal(got[0], got[1])


def test_hip_step_against_vectors_evaluated_from_the_reference_text(gpu_pkg, golden):
    """The HIP path against tests/golden/cartpole_reference_text.npz — CartPoleEnv.Step outputs obtained by evaluating the
    reference's own source text (oracle/evaluate_reference_text.py; see tests/test_oracle.py): float32 state within 1e-5 of the
    reference's float64 result; done, reward and steps_beyond_done exact on every instance — including the 200 the fixture puts
    within +-2 float32 ulps of a threshold on purpose (the kernel takes the flag from the reference's float64 sums)."""
    g = golden("cartpole_reference_text")
    n = g["state"].shape[1]
    with gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED, auto_reset=False) as env:
        env.Reset()
        env.SetState(g["state"].astype(np.float32))                 # the fixture's states are binary32 values
        env.SetStepsBeyondDone(g["sbd"])
        out = env.Step(g["action"])
        got = env.GetState().astype(np.float64)
        want = g["next_state"]
        err = np.abs(got - want)
        inr = slice(0, 2400)
        assert err[:, inr].max() <= TOL                                             # in-range block: absolute bar of north_star
        assert (err / np.maximum(1.0, np.abs(want))).max() <= TOL                   # wide block: values up to ~1e2
        near = _near_threshold(want, 1e-6)
        assert 100 <= near.sum() <= 260                                             # the fixture's +-2-ulp block is really there
        # integer outputs: ALL 3200 instances, the +-2-float32-ulp block included (row a3: "must be bit-exact")
        assert np.array_equal(out.Done, g["done"].astype(bool))
        assert np.array_equal(out.Reward, g["reward"])
        assert np.array_equal(env.GetStepsBeyondDone(), g["sbd_out"])


def test_single_instance_facade_with_the_time_limit_extension(gpu_pkg):
    """GpuEnv(max_episode_steps=k): upstream gym's TimeLimit on the single-instance façade (an extension; the reference has none).
    A MountainCar episode under a constant action never reaches the goal, so it ends exactly at the limit, by truncation, with
    Information["TimeLimit.truncated"]; CartPole's natural termination carries no such flag."""
    mc = gpu_pkg.MountainCarEnv(seed=3, max_episode_steps=25)
    try:
        mc.Reset()
        for t in range(1, 26):
            obs, reward, done, info = mc.Step(1)
            assert reward == -1.0 and done == (t == 25)
        assert info == {"TimeLimit.truncated": True}
        mc.Reset()
        obs, reward, done, info = mc.Step(1)
        assert not done and info is None                              # a reset starts the count again
    finally:
        mc.CloseEnvironment()
    cp = gpu_pkg.CartPoleEnv(seed=3, max_episode_steps=500)
    try:
        cp.Reset()
        for t in range(200):
            obs, reward, done, info = cp.Step(1)                          # always push right: falls within a dozen steps
            if done:
                break
        assert done and t < 30 and info is None
    finally:
        cp.CloseEnvironment()
