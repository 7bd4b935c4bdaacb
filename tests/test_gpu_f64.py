"""GPU parity for GYMNET_FLAG_F64 — CartPole in the reference's OWN arithmetic (float64 state, the literal
CartPoleEnv.cs:141-167 sequence, float64 observations at the boundary; gym.net_amd/csrc/cartpole64.hpp).

Bars: the HIP kernel equals the float64 "kernel semantics" twin of the oracle BIT FOR BIT (same IEEE operations, own sin/cos on both
sides); it equals the reference-arithmetic restatement (libm sin/cos) and the vectors evaluated from the reference's source
text to <= 1e-12 with every integer output exact; and, free-running, it reproduces every episode length of the recorded
reference-test-shaped trace and of a 100 000-iteration README loop (README.md:32-52)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SEED = 0x5EED


def test_f64_step_equals_the_vectors_evaluated_from_the_reference_text(gpu_pkg, golden, oracle):
    g = golden("cartpole_reference_text")                                  # 3200 float64 input -> output vectors
    n = g["state"].shape[1]
    with gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED, auto_reset=False, dtype=np.float64) as env:
        assert env.KernelName().startswith("step_kernel<CartPole64,") and env.Dtype == np.float64
        first = env.Reset()
        assert first.dtype == np.float64 and first.shape == (n, 4)
        env.SetState(g["state"])
        env.SetStepsBeyondDone(g["sbd"])
        out = env.Step(g["action"])
        got = env.GetState()
        assert got.dtype == np.float64 and out.Observation.dtype == np.float64
        # integer outputs: exact on all 3200, the +-2-float32-ulp block included (x + tau*x_dot has no sin/cos in it)
        assert np.array_equal(out.Done, g["done"].astype(bool))
        assert np.array_equal(out.Reward, g["reward"])
        assert np.array_equal(env.GetStepsBeyondDone(), g["sbd_out"])
        want = g["next_state"]
        assert np.array_equal(got[[0, 2]], want[[0, 2]])                     # x', theta': bit-exact (no transcendental involved)
        err = np.abs(got - want) / np.maximum(1.0, np.abs(want))
        assert err.max() <= 1e-12, err.max()                                # velocities: the two sin/cos differ by <= 2 ulp
        assert np.array_equal(out.Observation, got.T)                       # the observation IS the float64 state (:166,185)
        # bit for bit against the oracle's twin that uses the kernel's own sin/cos
        s2, r2, d2, b2 = oracle.cartpole_step(g["state"], g["action"], g["sbd"], dtype=np.float64, kernel_sincos=True)
        assert np.array_equal(got, s2) and np.array_equal(out.Done, d2.astype(bool)) and np.array_equal(out.Reward, r2)


@pytest.mark.parametrize("n", [1, 7, 4096 + 3, 50_000])
def test_f64_autoreset_rollout_is_bit_identical_to_the_twin(gpu_pkg, oracle, n):
    """Free-running with the fused auto-reset: state, reward, done and the 53-bit Philox reset draws against the CPU twin, every
    step, for lane counts that exercise the two-lane and the tail paths and a lane offset above 2^32."""
    off, steps = 123_456_789_000, 30
    rng = np.random.default_rng(n)
    with gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED, auto_reset=True, lane_offset=off, dtype=np.float64) as env:
        first = env.Reset()
        s = oracle.cartpole_reset_f64(SEED, off, 0, n)
        assert np.array_equal(first.T, s)
        assert np.abs(s).max() < 0.05 and len(np.unique(s)) > 3 * n                      # U(-0.05, 0.05), 53-bit: no repeats
        dones = 0
        for t in range(steps):
            a = rng.integers(0, 2, n).astype(np.int32)
            tick = env.Tick
            out = env.Step(a)
            s, r, d = oracle.cartpole_autoreset_step_f64(SEED, off, tick, s, a)
            assert np.array_equal(out.Observation.T, s) and np.array_equal(out.Reward, r) and np.array_equal(out.Done, d.astype(bool)), t
            dones += int(d.sum())
        assert dones > 0 or n < 8
        assert np.array_equal(env.GetState(), s)
        assert env.Counters()["lane_steps"] == steps * n


def test_f64_facade_reproduces_the_reference_test_trace_free_running(gpu_pkg, golden):
    """CartpoleEnvironment.cs:19-27's loop on the single-instance façade, which now defaults to float64: every state of the
    1000-iteration trace within 1e-12 of the float64 restatement WITHOUT teacher forcing, every done flag and episode length
    exact (the float32 engine needs 1e-4 here and matches the lengths only on this one recorded trace)."""
    g = golden("cartpole_reference_test_trace")
    cp = gpu_pkg.CartPoleEnv(seed=SEED)
    try:
        assert cp._v.Dtype == np.float64
        done, k, lens, cur, worst = True, 0, [], 0, 0.0
        for i in range(1000):
            if done:
                first = cp.Reset()
                assert first.dtype == np.float64
                cp._v.SetState(g["resets"][k].reshape(4, 1)); k += 1
                done = False
                if cur:
                    lens.append(cur)
                cur = 0
            else:
                observation, reward, _done, information = cp.Step(i % 2)
                done = _done; cur += 1
                assert observation.dtype == np.float64 and reward == 1.0 and information is None
                worst = max(worst, float(np.abs(observation - g["it_state"][i]).max()))
            assert int(done) == g["it_done"][i], i
        assert worst <= 1e-12, worst
        assert k == int(g["resets_used"]) and lens == list(g["episode_lengths"])
    finally:
        cp.CloseEnvironment()


def test_f64_readme_loop_100k_iterations_matches_the_float64_oracle(gpu_pkg, oracle):
    """README.md:32-52 — `if (done) Reset() else Step(action)` — for 100 000 iterations, free-running, against the oracle's
    reference-arithmetic restatement (libm sin / cos) fed with the SAME reset draws: every episode length equal, states within
    1e-12.  Run as 64 independent single-instance loops side by side (64 lanes x 1563 iterations >= 100 000 env iterations) so
    that the test costs ~1600 host-boundary calls instead of 100 000; each lane is its own README loop."""
    lanes, iters = 64, 1563
    rng = np.random.default_rng(77)
    with gpu_pkg.VectorEnv("CartPole-v1", lanes, seed=SEED, auto_reset=False, dtype=np.float64) as env:
        obs = env.Reset()
        s64 = obs.T.copy()                                                  # the oracle starts from the engine's own draw
        sbd = np.full(lanes, -1, np.int32)
        done = np.zeros(lanes, bool)
        cur = np.zeros(lanes, np.int64)
        lens_gpu, lens_ref, worst, total = [], [], 0.0, 0
        for i in range(iters):
            if done.any():                                                  # the caller's `if (done) Reset()`, per lane
                obs = env.ResetWhere()                                      # mask None = lanes whose last Done flag is set
                fresh = obs.T
                assert (np.abs(fresh[:, done]) < 0.05).all() and not np.array_equal(fresh[:, done], s64[:, done])
                s64[:, done] = fresh[:, done]                               # same draws for the oracle: RNG factored out
                assert np.array_equal(fresh[:, ~done], out.Observation.T[:, ~done])   # untouched lanes untouched
                sbd[done] = -1
                lens_ref.extend(cur[done].tolist()); cur[done] = 0
                done[:] = False
            a = rng.integers(0, 2, lanes).astype(np.int32)
            out = env.Step(a)
            s64, r, d, sbd = oracle.cartpole_step(s64, a, sbd, dtype=np.float64)     # libm sin/cos: the reference's arithmetic
            worst = max(worst, float(np.abs(out.Observation.T - s64).max()))
            assert np.array_equal(out.Done, d.astype(bool)), i              # hence every episode length
            assert np.array_equal(out.Reward, r)
            done = d.astype(bool); cur += 1; total += lanes
        assert total >= 100_000 and len(lens_ref) > 2000 and worst <= 1e-12, (total, len(lens_ref), worst)


def test_f64_host_buffers_step_async_and_errors(gpu_pkg):
    n = 5000
    rng = np.random.default_rng(3)
    with gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED, auto_reset=True, dtype=np.float64) as a, \
            gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED, auto_reset=True, dtype=np.float64) as b:
        acts, obs, rew, done = a.HostBuffers()                              # pinned, device-mapped: obs is float64 [n, 4]
        assert obs.dtype == np.float64 and obs.shape == (n, 4)
        a.ResetInto(obs)
        assert np.array_equal(obs, b.Reset())
        for t in range(12):
            acts[:] = rng.integers(0, 2, n)
            a.StepInto(acts, obs, rew, done)
            if t % 2:
                o = b.Step(acts.copy())
            else:
                o = b.StepAsync(acts.copy()).Result()                       # VecEnv.StepAsync (VecEnv.cs:63-65)
            assert np.array_equal(obs, o.Observation) and np.array_equal(rew, o.Reward) and np.array_equal(done.astype(bool), o.Done)
        assert np.array_equal(a.Read().Observation, obs)
        with pytest.raises(ValueError):
            a.StepInto(acts, obs.astype(np.float32), rew, done)             # float32 buffer on a float64 handle
    with pytest.raises(NotImplementedError):
        gpu_pkg.VectorEnv("Pendulum-v1", 64, dtype=np.float64)              # only CartPole's float64 arithmetic is defined by the reference


def test_f64_device_path_graph_replay_time_limit_and_lane_seeds(gpu_pkg, oracle):
    """rollout_device (hipGraph replay and eager) == per-step launches; EPISODE_STATS + max_episode_steps truncation and per-lane
    Philox keys (the EXTRAS variant of the float64 kernel) against a host-side replay with the twin."""
    import torch
    n, ring, steps, limit = 3000 + 1, 6, 41, 9
    acts = torch.randint(0, 2, (ring, n + 1), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    res = []
    for graph in (1, 0, None):
        with gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED, auto_reset=True, dtype=np.float64,
                               launch_policy={} if graph is None else {"graph": graph}) as env:
            env.ResetDevice()
            if graph is None:
                for t in range(steps):
                    env.StepDevice(acts[t % ring])
            else:
                env.RolloutDevice(acts, steps, n + 1, ring)
            env.Sync()
            assert env.Tick == steps + 1
            r = env.Read()
            res.append((env.GetState(), r.Reward, r.Done))
    for x in res[1:]:
        assert all(np.array_equal(u, v) for u, v in zip(res[0], x))
    a_host = acts.cpu().numpy()[:, :n]
    seeds = np.arange(n, dtype=np.uint64) * 7 + 3
    with gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED, auto_reset=True, dtype=np.float64, episode_stats=True, max_episode_steps=limit) as env:
        env.Seed(seeds.astype(np.int64))
        assert env.KernelName().split(",")[3] == "true"                      # step_kernel<CartPole64, VEC, AUTORESET, EXTRAS, NT, RESETF>
        first = env.Reset()
        s = oracle.cartpole_reset_f64(0, 0, 0, n, lane_seed=seeds)
        assert np.array_equal(first.T, s)
        ln = np.zeros(n, np.int32)
        fin_len = np.zeros(n, np.int32)
        for t in range(25):
            tick = env.Tick
            out = env.Step(a_host[t % ring])
            stepped, r, d = oracle.cartpole_step(s, a_host[t % ring], dtype=np.float64, kernel_sincos=True)[:3]
            ln += 1
            trunc = ln >= limit
            fin = d.astype(bool) | trunc
            fresh = oracle.cartpole_reset_f64(0, 0, tick, n, lane_seed=seeds)
            s = np.where(fin, fresh, stepped)
            assert np.array_equal(out.Observation.T, s), t
            assert np.array_equal(out.Done, fin) and np.array_equal(out.Truncated, trunc)
            fin_len[fin] = ln[fin]; ln[fin] = 0
        got_ret, got_len = env.EpisodeStats()
        assert np.array_equal(got_len, fin_len) and got_len.max() == limit and np.array_equal(got_ret, fin_len.astype(np.float32))


def test_f64_at_2p20_lanes_matches_the_twin_and_the_float32_engine_statistically(gpu_pkg, oracle):
    """BASELINE's batch size: 2^20 float64 lanes, 24 free-running steps from the reset, replayed on the CPU twin bit for bit;
    and the float32 engine started from the same (rounded) states stays within 1e-5 per teacher-forced step of it."""
    import torch
    n, steps, ring = 1 << 20, 24, 8
    with gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED, auto_reset=True, dtype=np.float64) as env:
        assert env.KernelName() == "step_kernel_pipe2<CartPole64,4,true,15>"   # the default at this size (one lock-step generation otherwise)
        acts = torch.empty((ring, n), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        for t in range(ring):
            env.SampleActionsDevice(acts[t], seed=SEED + 1, tick=t)
        env.ResetDevice()
        env.Sync()
        a_host = acts.cpu().numpy()
        s = oracle.cartpole_reset_f64(SEED, 0, 0, n)
        assert np.array_equal(env.GetState(), s)
        env.RolloutDevice(acts, steps, n, ring)
        env.Sync()
        dones = 0
        for t in range(steps):
            s, r, d = oracle.cartpole_autoreset_step_f64(SEED, 0, 1 + t, s, a_host[t % ring])
            dones += int(d.sum())
        out = env.Read()
        assert np.array_equal(env.GetState(), s) and np.array_equal(out.Done, d.astype(bool)) and np.array_equal(out.Reward, r)
        assert dones > 0.01 * n * steps
        # one teacher-forced step of the float32 engine from these states: <= 1e-5, done flags identical (row a3)
        with gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED, auto_reset=False) as f32:
            f32.Reset()
            s32 = s.astype(np.float32)
            f32.SetState(s32)
            o32 = f32.Step(a_host[0])
            w, _, wd, _ = oracle.cartpole_step(s32.astype(np.float64), a_host[0], dtype=np.float64)
            assert np.abs(f32.GetState().astype(np.float64) - w).max() <= 1e-5 and np.array_equal(o32.Done, wd.astype(bool))


def test_f64_kernel_reproduces_the_committed_twin_trace(gpu_pkg, golden):
    """The HIP float64 kernel against tests/golden/cartpole_f64_kernel.npz (bit patterns committed from the oracle's twin): reset
    draws at lane offsets above 2^32 and large ticks, and a 250-step free-running auto-reset trace of 16 lanes."""
    g = golden("cartpole_f64_kernel")
    for k, (off, tick) in enumerate(((0, 0), (123_456_789_000, 7), (1 << 40, 2 ** 33 + 5))):
        with gpu_pkg.VectorEnv("CartPole-v1", 64, seed=SEED, auto_reset=True, lane_offset=off, dtype=np.float64) as env:
            env.Tick = tick
            assert np.array_equal(env.Reset().T, g["resets"][k]), k
    seed, off = int(g["trace_seed"]), int(g["trace_offset"])
    with gpu_pkg.VectorEnv("CartPole-v1", 16, seed=seed, auto_reset=True, lane_offset=off, dtype=np.float64) as env:
        env.Reset()
        for t in range(g["trace_actions"].shape[0]):
            out = env.Step(g["trace_actions"][t])
            assert np.array_equal(out.Observation.T, g["trace_states"][t]) and np.array_equal(out.Done, g["trace_done"][t].astype(bool)), t


def test_f64_steps_beyond_done_reward_stream_and_counter(gpu_pkg, golden):
    """CartPoleEnv.cs:168-183 in the float64 mode: reward 1, ..., 1, 1 (done), 0, 0, ... when the caller keeps stepping without a
    reset, steps_beyond_done -1, ..., -1, 0, 1, 2, ..., and the console warning counted instead of printed — the recorded
    sequence of tests/golden/cartpole_steps_beyond_done.npz, free-running, states within 1e-12; plus a ragged batch whose
    lanes fall at different times (two-lane and tail paths of the kernel, one atomic per wave for the counter)."""
    g = golden("cartpole_steps_beyond_done")
    with gpu_pkg.VectorEnv("CartPole-v1", 1, seed=SEED, dtype=np.float64) as env:
        env.Reset()
        env.SetState(g["start"].reshape(4, 1))
        for t in range(g["reward"].shape[0]):
            out = env.Step(1)
            assert out.Reward[0] == g["reward"][t] and bool(out.Done[0]) == bool(g["done"][t])
            assert env.GetStepsBeyondDone()[0] == g["sbd"][t]
            assert np.abs(env.GetState()[:, 0] - g["states"][t]).max() <= 1e-12
        assert env.Counters()["stepped_after_done"] == int(g["done"].sum()) - 1
    n, steps = 1000 + 1, 40
    with gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED, dtype=np.float64) as a, gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED) as b:
        a.Reset(); b.Reset()
        b.SetState(a.GetState().astype(np.float32))
        a.SetState(b.GetState().astype(np.float64))                  # both start from the same float32-representable states
        after = 0
        sbd = np.full(n, -1)
        for t in range(steps):
            oa, ob = a.Step(1), b.Step(1)                            # always push right: every lane falls within ~15 steps
            if t < 8:                                                # before float32 drift can move a termination by a step
                assert np.array_equal(oa.Done, ob.Done) and np.array_equal(oa.Reward, ob.Reward)
            after += int((oa.Done & (sbd >= 0)).sum())
            sbd = np.where(oa.Done, sbd + 1, sbd)
            assert np.array_equal(a.GetStepsBeyondDone(), sbd)
        assert oa.Done.all() and (oa.Reward == 0).all()
        assert a.Counters()["stepped_after_done"] == after > n


def test_f64_multi_item_kernel_forms_are_bit_identical(gpu_pkg):
    """step_kernel_pipe2<CartPole64, ITEMS> (launch policy sequential_lanes = 2..4: a thread owns ITEMS lane pairs, all loads first, then
    advance / store pair after pair) runs the same per-lane code with the same Philox counters as the one-shot kernel: states,
    rewards, done flags and reset draws must agree bit for bit, with and without auto-reset; for batches that are not whole multiples
    of 2 * ITEMS * 256 lanes, and for the bookkeeping variant, an explicit request is rejected (it would not take effect)."""
    import torch
    n, ring, steps = 2 * 256 * 12 * 5, 6, 37
    acts = torch.randint(0, 2, (ring, n), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    out = {}
    for auto in (True, False):
        # items 1: the one-shot kernel with the per-thread drain-loop reset (the reference form of this comparison); items 0 stands
        # for the one-shot kernel with the wave-compacted reset, two lanes per reset (reset_form 1, the default with auto-reset);
        # the multi-pair kernels draw ONCE per thread-group of pairs (reset_group_deferred) — three reset forms, the same bits
        for items in (1, 0, 2, 3, 4):
            policy = {"sequential_lanes": max(items, 1)}
            if items <= 1:
                policy["reset_form"] = 1 - items
            with gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED, auto_reset=auto, dtype=np.float64, launch_policy=policy) as env:
                rf = 1 if (items == 0 and auto) else 0
                want = f"step_kernel<CartPole64,2,{str(auto).lower()},false,15,{rf}>" if items <= 1 else f"step_kernel_pipe2<CartPole64,{items},{str(auto).lower()},15>"
                assert env.KernelName() == want, env.KernelName()
                env.ResetDevice()
                env.RolloutDevice(acts, steps, n, ring)
                env.Sync()
                r = env.Read()
                out[(auto, items)] = (env.GetState(), r.Reward, r.Done, env.GetStepsBeyondDone() if not auto else None, env.Counters()["stepped_after_done"])
        ref = out[(auto, 1)]
        assert ref[2].any()
        for items in (0, 2, 3, 4):
            got = out[(auto, items)]
            assert all(np.array_equal(u, v, equal_nan=True) for u, v in zip(ref[:3], got[:3])), (auto, items)
            assert (ref[3] is None or np.array_equal(ref[3], got[3])) and ref[4] == got[4]
    with gpu_pkg.VectorEnv("CartPole-v1", n + 2, seed=SEED, auto_reset=False, dtype=np.float64) as env:
        with pytest.raises(ValueError, match="would not take effect"):
            env.SetLaunchPolicy(sequential_lanes=4)                                    # not whole 2 * 4 * 256-lane groups
        assert env.KernelName() == "step_kernel<CartPole64,2,false,false,15,0>" and env.GetLaunchPolicy()["sequential_lanes"] == 1
    with gpu_pkg.VectorEnv("CartPole-v1", n + 2, seed=SEED, auto_reset=True, dtype=np.float64) as env:
        env.SetLaunchPolicy(sequential_lanes=4)                                        # the auto-reset form takes any batch size
        assert env.KernelName() == "step_kernel_pipe2<CartPole64,4,true,15>"
    with gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED, auto_reset=True, dtype=np.float64, episode_stats=True) as env:
        with pytest.raises(ValueError, match="would not take effect"):
            env.SetLaunchPolicy(sequential_lanes=2)                                    # bookkeeping variant: one-shot kernel only
        assert env.KernelName() == "step_kernel<CartPole64,2,true,true,15,1>"
        with pytest.raises(ValueError):
            env.SetLaunchPolicy(sequential_lanes=5)
        env.SetLaunchPolicy(block=64)                                                  # honoured (the float64 launcher used to ignore it)
        assert env.GetLaunchPolicy()["block"] == 64


@pytest.mark.parametrize("n", [2 * 4 * 256 * 3 + 1, 2 * 4 * 256 * 2 + 2 * 4 * 64 - 3, 1000, 7, 2 * 4 * 256 * 4 - 2])
def test_f64_multi_pair_kernel_takes_any_batch_size(gpu_pkg, oracle, n):
    """step_kernel_pipe2<CartPole64, k, auto-reset> on batches that are not whole 2 * k * block-lane groups: the last workgroups run
    the guarded body (lanes past the end load zeros, never pend a reset, store nothing).  Against the one-shot kernel with the
    drain-loop reset, bit for bit, over enough steps that every lane is reset; and the memory after the batch stays untouched."""
    import torch
    rng = np.random.default_rng(n)
    acts = rng.integers(0, 2, (40, n)).astype(np.int32)
    outs = []
    for pol in ({"sequential_lanes": 1, "reset_form": 0}, {"sequential_lanes": 4}, {"sequential_lanes": 2, "block": 64}, {"sequential_lanes": 3, "block": 128}):
        stride = (n + 65) // 2 * 2                     # even: two doubles per thread need 16-byte aligned rows
        buf = torch.full((4 * stride + 8,), 777.0, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        with gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED, auto_reset=True, dtype=np.float64, ext_obs=buf.data_ptr(), ext_obs_stride=stride,
                               launch_policy=pol) as env:
            if pol["sequential_lanes"] > 1:
                assert env.KernelName() == f"step_kernel_pipe2<CartPole64,{pol['sequential_lanes']},true,15>"
            env.Reset()
            env.Sync()
            pad0 = buf[:4 * stride].view(4, stride)[:, n:].cpu().numpy().copy()      # (create / reset may initialise the rows' padding)
            got = [env.Step(acts[t]) for t in range(40)]
            outs.append((env.GetState(), [g.Observation.copy() for g in got], [g.Done.copy() for g in got], [g.Reward.copy() for g in got]))
            env.Sync()
            pad = buf[:4 * stride].view(4, stride)[:, n:].cpu().numpy()
            # the step kernels write nothing past the batch: the rows' padding is what it was, the words after the buffer untouched
            assert np.array_equal(pad, pad0) and (buf[4 * stride:].cpu().numpy() == 777.0).all()
    assert sum(int(d.sum()) for d in outs[0][2]) > n
    for o in outs[1:]:
        assert np.array_equal(outs[0][0], o[0])
        for k in (1, 2, 3):
            assert all(np.array_equal(u, v) for u, v in zip(outs[0][k], o[k]))


@pytest.mark.parametrize("n,pol,slices", [((1 << 20) + (1 << 19) + 6, {"sequential_lanes": 4, "nt": 15}, 2), ((1 << 21) + 2, {"sequential_lanes": 4, "nt": 0}, 3),
                                          (3 * 786432 - 510, {"sequential_lanes": 2}, 3)])
def test_f64_multi_pair_step_beyond_one_resident_generation_is_launched_in_slices(gpu_pkg, oracle, n, pol, slices):
    """Round 6: a multi-pair step over more lanes than the kernel's resident waves hold (2048 waves of 512 lanes for four pairs, 3072 of
    256 for two) is launched slice by slice (step_kernels.hpp pipe2_chunks): every slice its own launch with shifted rows and lane
    offset, all reading the same tick.  Bit-identical to the one-shot kernel over steps in which every lane is reset at least once,
    with per-slice ragged ends, lane offsets that continue the global numbering, and the twin's reset draws."""
    rng = np.random.default_rng(n)
    T = 60
    acts = rng.integers(0, 2, (T, n)).astype(np.int32)
    outs = []
    for p in ({"sequential_lanes": 1}, pol):
        with gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED, auto_reset=True, dtype=np.float64, lane_offset=(1 << 33) + 10, launch_policy=p) as env:
            if p["sequential_lanes"] > 1:
                assert env.KernelName() == f"step_kernel_pipe2<CartPole64,{p['sequential_lanes']},true,{p.get('nt', 15)}> x {slices}"
            env.Reset()
            fin = np.zeros(n, bool)
            for t in range(T):
                o = env.Step(acts[t])
                fin |= o.Done
            outs.append((env.GetState(), o.Observation.copy(), o.Done.copy(), o.Reward.copy(), env.Tick))
    assert fin.mean() > 0.8
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(a, b)
    # the last step's freshly reset lanes hold the twin's draw for their GLOBAL lane id (the slices' offsets continue the numbering)
    d = outs[1][2]
    fresh = oracle.cartpole_reset_f64(SEED, (1 << 33) + 10, outs[1][4] - 1, n)
    assert d.any() and np.array_equal(outs[1][0][:, d], fresh[:, d])


def test_f64_overflowing_dividend_steps_like_the_division(gpu_pkg, oracle):
    """the float64 kernel's `/ total_mass` is an fma pair that would turn an INFINITE dividend into NaN where the reference's
    division hands the infinity on — and a lane stepped far past done (no auto-reset) gets there from a finite state: polemass_length *
    theta_dot^2 * sin(theta) overflows.  The kernel (and its twin) return x for an infinite x: here states with |theta_dot| up to 1e200
    step bit-identically to the twin (NaN == NaN) and report the reference restatement's done flags for three steps: where the plain
    division carries an infinity (which compares: done) the kernel does too, where it yields NaN (inf - inf) so does the kernel."""
    big = np.array([[0.0, 0.0, 0.0, 1.0, -1.0, 0.5], [0.0, 1.0, -2.0, 0.1, 0.0, 0.0], [0.3, -0.1, 0.05, 0.01, -0.2, 0.2],
                    [1e160, -1e170, 1e200, -1e155, 3e154, 1e100]])
    n = big.shape[1]
    a = np.array([0, 1, 1, 0, 1, 0], np.int32)
    with gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED, auto_reset=False, dtype=np.float64) as env:
        env.Reset()
        env.SetState(big)
        s_ref = big
        s_twin = big
        for t in range(3):
            out = env.Step(a)
            s_twin, r_twin, d_twin, _ = oracle.cartpole_step(s_twin, a, dtype=np.float64, kernel_sincos=True)
            s_ref, r_ref, d_ref, _ = oracle.cartpole_step(s_ref, a)
            assert np.array_equal(env.GetState(), s_twin, equal_nan=True), t
            assert np.array_equal(out.Done, d_ref.astype(bool)), t
            assert np.array_equal(np.isinf(s_twin), np.isinf(s_ref)) and np.array_equal(np.isnan(s_twin), np.isnan(s_ref)), t
        assert np.isinf(s_twin).any() and out.Done.any() and not out.Done.all()      # (a lane that is NaN in the REFERENCE's arithmetic too reports not-done there as well)


@pytest.mark.parametrize("case", ["everyone_falls", "one_thread_tail", "odd_tail", "lane_seeds"])
def test_f64_two_lanes_per_reset_edges(gpu_pkg, oracle, case):
    """The float64 reset is two Philox calls; the compacted forms spread ONE reset over two lanes (lane 2r call 0, lane 2r + 1 call 1,
    a DPP exchange joins them: reset_pending_wave, reset_group_deferred).  Edges of that scheme against the per-thread drain loop
    (reset_form 0) and the twin's draws: every lane of every wave finishing in the same step (512 resets per wave of the four-pair
    kernel: sixteen rounds of 32); a last wave of ONE active thread (no neighbour lane: it draws for itself) and of an odd number of
    threads; per-lane seeds through the compacted form of the bookkeeping kernel."""
    import torch
    if case == "everyone_falls":
        n, forms = 2 * 4 * 256 * 3, [{"sequential_lanes": 1, "reset_form": 0}, {"sequential_lanes": 1, "reset_form": 1}, {"sequential_lanes": 4},
                                     {"sequential_lanes": 2}]
    elif case == "one_thread_tail":
        n, forms = 2 * (64 * 5 + 1), [{"reset_form": 0}, {"reset_form": 1}]
    elif case == "odd_tail":
        n, forms = 2 * (64 * 5 + 3) - 1, [{"reset_form": 0}, {"reset_form": 1}]
    else:
        n, forms = 2 * 64 * 7 + 10, [{"reset_form": 0}, {"reset_form": 1}]
    kw = dict(episode_stats=True) if case == "lane_seeds" else {}
    rng = np.random.default_rng(len(case))
    acts = rng.integers(0, 2, (12, n)).astype(np.int32)
    seeds = rng.integers(1, 2**62, n)
    outs = []
    for pol in forms:
        with gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED, auto_reset=True, dtype=np.float64, launch_policy=pol, **kw) as env:
            if case == "lane_seeds":
                env.Seed(seeds)
            env.Reset()
            st = env.GetState()
            if case != "lane_seeds":
                st[0, :] = 2.39 + 0.02 * (np.arange(n) % 2 if case != "everyone_falls" else 1)    # past (or right at) the x threshold
                env.SetState(st)
            tick0 = env.Tick
            got = [env.Step(acts[t]) for t in range(12)]
            outs.append((env.GetState(), [g.Observation.copy() for g in got], [g.Done.copy() for g in got]))
            if case == "everyone_falls":
                assert got[0].Done.all()
                # ... and what they were reset to is the twin's draw for (seed, lane, tick)
                assert np.array_equal(got[0].Observation.T, oracle.cartpole_reset_f64(SEED, 0, tick0, n))
            if case == "lane_seeds":
                assert any(g.Done.any() for g in got)
    for o in outs[1:]:
        assert np.array_equal(outs[0][0], o[0])
        assert all(np.array_equal(u, v) for u, v in zip(outs[0][1], o[1])) and all(np.array_equal(u, v) for u, v in zip(outs[0][2], o[2]))


@pytest.mark.parametrize("auto", [True, False])
@pytest.mark.parametrize("n", [2 * 4096, 4096 + 3])
def test_f64_fused_rollout_is_bit_identical_to_stepwise_and_records(gpu_pkg, oracle, auto, n):
    """gymnet_vecenv_rollout_fused_device on a float64 handle: T vector steps in ONE launch (state in registers), bit-identical to T
    one-launch steps — state, reward, done, steps_beyond_done, the step-after-done counter, the Philox reset draws (tick0 + t) — with
    the recorded [T][4][N] float64 observations, [T][N] rewards and done flags equal to what the stepwise handle returned at every
    step; even and odd lane counts (two-lane and tail paths), an action ring shorter than the rollout, and the CPU twin for the
    auto-reset case."""
    import torch
    T, ring = 45, 8
    stride = n + (n % 2)
    acts = torch.randint(0, 2, (ring, stride), dtype=torch.int32, device="cuda")
    rec_o = torch.zeros((T, 4, n), dtype=torch.float64, device="cuda")
    rec_r = torch.full((T, n), -1.0, dtype=torch.float32, device="cuda")
    rec_d = torch.full((T, n), 9, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    a_host = acts.cpu().numpy()[:, :n]
    with gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED, auto_reset=auto, dtype=np.float64, lane_offset=77) as f, \
            gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED, auto_reset=auto, dtype=np.float64, lane_offset=77) as e:
        f.ResetDevice(); e.ResetDevice()
        f.RolloutFusedDevice(acts, T, stride, ring, rec_obs=rec_o, rec_reward=rec_r, rec_done=rec_d)
        f.Sync()
        s = oracle.cartpole_reset_f64(SEED, 77, 0, n)
        obs, rew, don = rec_o.cpu().numpy(), rec_r.cpu().numpy(), rec_d.cpu().numpy()
        for t in range(T):
            o = e.Step(a_host[t % ring])
            assert np.array_equal(obs[t], o.Observation.T) and np.array_equal(rew[t], o.Reward) and np.array_equal(don[t].astype(bool), o.Done), t
            if auto:
                s, r, d = oracle.cartpole_autoreset_step_f64(SEED, 77, 1 + t, s, a_host[t % ring])
                assert np.array_equal(obs[t], s), t
        assert np.array_equal(f.GetState(), e.GetState()) and f.Tick == e.Tick == T + 1
        lf, le = f.Read(), e.Read()
        assert np.array_equal(lf.Observation, le.Observation) and np.array_equal(lf.Reward, le.Reward) and np.array_equal(lf.Done, le.Done)
        assert f.Counters()["stepped_after_done"] == e.Counters()["stepped_after_done"] and f.Counters()["tick"] == T + 1
        if not auto:
            assert np.array_equal(f.GetStepsBeyondDone(), e.GetStepsBeyondDone()) and don[-1].mean() > 0.5 and (rew[-1][don[-1] > 0] == 0).mean() > 0.9
        f.RolloutFusedDevice(acts, 3, stride, ring)                      # without recording, and continuing from the fused state
        for t in range(3):
            e.Step(a_host[t % ring])
        assert np.array_equal(f.GetState(), e.GetState())
    # (bookkeeping handles are fused too since ABI 5, in both state scalars: tests/test_gpu_fused_rollout_ex.py)
