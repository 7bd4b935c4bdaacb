"""GPU tests of the fused rollout that carries its consumer's bookkeeping (gymnet_vecenv_rollout_fused_ex_device):
`steps` vector steps in ONE kernel launch on a BOOKKEEPING handle — episode return / length in registers, max_episode_steps
truncation, per-lane seeds, dense last-finished-episode views, compact (t, lane, return, length) records of every episode that
ends (examples/.../PlaySessions/BasePlaySession.cs:58-69, MemoryTypes/ReplayMemory.cs:53-67) — with the actions read from a ring,
DRAWN IN THE KERNEL (ActionSpace.Sample()) or composed epsilon-greedy over the ring (TrainingPlaySession.cs:46-52).

Bars: bit-identical to `steps` x (SampleActionsDevice / ComposeActionsDevice -> StepDevice) on a twin handle — state, running and
finished episode statistics, terminal observations, the done list of the last step, the recorded streams, the actions taken, the
set of episode records — for float32 CartPole, the float64 mode, Acrobot (derived observation) and Pendulum (Box actions); and at
BASELINE's 2^20 lanes against the oracle's replay."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SEED, ASEED = 0x5EED, 0xAC710


def _episode_buffers(torch, cap):
    return dict(step=torch.full((cap,), -1, dtype=torch.int32, device="cuda"), lane=torch.full((cap,), -1, dtype=torch.int32, device="cuda"),
                ret=torch.zeros(cap, dtype=torch.float32, device="cuda"), length=torch.zeros(cap, dtype=torch.int32, device="cuda"),
                capacity=cap, count=torch.zeros(2, dtype=torch.uint32, device="cuda"))


def _records(ep):
    c = ep["count"].cpu().numpy().astype(np.int64)
    k = int(c[0])
    rec = np.stack([ep["step"].cpu().numpy()[:k], ep["lane"].cpu().numpy()[:k], ep["length"].cpu().numpy()[:k]], axis=1)
    ret = ep["ret"].cpu().numpy()[:k]
    order = np.lexsort((rec[:, 1], rec[:, 0]))
    return rec[order], ret[order], c


@pytest.mark.parametrize("name,dtype", [("CartPole-v1", np.float32), ("CartPole-v1", np.float64), ("Acrobot-v1", np.float32), ("Pendulum-v1", np.float32)])
@pytest.mark.parametrize("actions", ["ring", "sample", "epsilon_greedy"])
@pytest.mark.parametrize("n", [4096 + 6, 1021, 8192])        # 8192: every stream aligned — the float64 rollout's four-lanes-per-thread form
def test_fused_rollout_with_bookkeeping_equals_stepwise(gpu_pkg, name, dtype, actions, n):
    import torch
    if name == "Pendulum-v1" and actions == "epsilon_greedy":
        pytest.skip("epsilon-greedy composition is defined for Discrete action spaces")
    T, ring, limit, tick0, eps = 37, 6, 17, 1000, 0.3
    box = name == "Pendulum-v1"
    adt = torch.float32 if box else torch.int32
    stride = n + (-n % 4)
    kw = dict(seed=SEED, auto_reset=True, dtype=dtype, lane_offset=12_345, done_list=True, episode_stats=True, final_obs=True,
              max_episode_steps=limit)
    with gpu_pkg.VectorEnv(name, n, **kw) as f, gpu_pkg.VectorEnv(name, n, **kw) as e:
        policy = torch.empty((ring, stride), dtype=adt, device="cuda")           # the ring: actions, or the policy's actions
        torch.cuda.synchronize()
        for t in range(ring):
            e.SampleActionsDevice(policy[t], seed=77, tick=t)
        e.Sync()
        seeds = (np.arange(n, dtype=np.int64) * 11 + 5)
        for env in (f, e):
            env.Seed(seeds)                                                        # per-lane Philox keys (VecEnv.Seed(int[]))
            env.ResetDevice()
        D = f.ObsDim
        tdt = torch.float64 if dtype == np.float64 else torch.float32
        rec_o = torch.zeros((T, D, n), dtype=tdt, device="cuda")
        rec_r = torch.zeros((T, n), dtype=torch.float32, device="cuda")
        rec_d = torch.zeros((T, n), dtype=torch.uint8, device="cuda")
        rec_a = torch.zeros((T, n), dtype=adt, device="cuda")
        ep = _episode_buffers(torch, n * T)
        torch.cuda.synchronize()
        f.RolloutFusedDevice(policy, T, stride, ring, rec_obs=rec_o, rec_reward=rec_r, rec_done=rec_d, rec_actions=rec_a, actions=actions,
                             action_seed=ASEED, action_tick0=tick0, epsilon=eps, episodes=ep)
        f.Sync()
        # the twin, one launch per step: draw / compose the action, step, read this step's records
        act = torch.empty(stride, dtype=adt, device="cuda")
        want_rec, want_ret = [], []
        for t in range(T):
            if actions == "ring":
                act.copy_(policy[t % ring]); torch.cuda.synchronize()     # torch's stream is not the handle's: order the copy before the step
            elif actions == "sample":
                e.SampleActionsDevice(act, seed=ASEED, tick=tick0 + t)
            else:
                e.ComposeActionsDevice(policy[t % ring], eps, act, seed=ASEED, tick=tick0 + t)
            e.StepDevice(act)
            e.Sync()
            o = e.Read()
            assert np.array_equal(rec_a[t].cpu().numpy(), act[:n].cpu().numpy()), t
            assert np.array_equal(rec_o[t].cpu().numpy(), o.Observation.T), t
            assert np.array_equal(rec_r[t].cpu().numpy(), o.Reward) and np.array_equal(rec_d[t].cpu().numpy() != 0, o.Done), t
            assert np.array_equal((rec_d[t].cpu().numpy() & 2) != 0, o.Truncated), t
            r = e.DoneRecords()
            for lane, ret, ln in zip(r["lanes"], r["return"], r["length"]):
                want_rec.append((t, int(lane), int(ln))); want_ret.append(float(ret))
        got_rec, got_ret, counts = _records(ep)
        want_rec = np.array(want_rec, dtype=np.int64).reshape(-1, 3)
        order = np.lexsort((want_rec[:, 1], want_rec[:, 0]))
        assert counts[0] == counts[1] == len(want_rec) > 0
        assert np.array_equal(got_rec, want_rec[order]) and np.array_equal(got_ret, np.array(want_ret, np.float32)[order])
        # the handle is in exactly the state T single steps leave
        assert np.array_equal(f.GetState(), e.GetState()) and f.Tick == e.Tick
        for arr in ("reward", "done", "episode_return", "episode_length", "finished_return", "finished_length", "final_obs"):
            assert np.array_equal(f.GetArray(arr), e.GetArray(arr)), arr
        a, b = f.DoneRecords(), e.DoneRecords()                                     # "the most recent step" = the rollout's last step
        oa, ob = np.argsort(a["lanes"]), np.argsort(b["lanes"])
        for k in ("lanes", "return", "length", "final_obs"):
            assert np.array_equal(a[k][oa], b[k][ob]), k
        assert f.Counters()["last_done_count"] == e.Counters()["last_done_count"]
        # and continues identically: stepwise after fused, fused (no records) after stepwise
        f.StepDevice(policy[0]); e.StepDevice(policy[0])
        e.RolloutFusedDevice(policy, 5, stride, ring)
        for t in range(5):
            f.StepDevice(policy[t % ring])
        f.Sync(); e.Sync()
        assert np.array_equal(f.GetState(), e.GetState()) and np.array_equal(f.GetArray("episode_length"), e.GetArray("episode_length"))
        assert np.array_equal(np.sort(f.DoneLanes()), np.sort(e.DoneLanes()))


def test_fused_rollout_ex_argument_errors_and_capacity(gpu_pkg):
    import torch
    n, T = 2048, 20
    acts = torch.randint(0, 2, (4, n), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    with gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED, auto_reset=True) as lean:
        lean.ResetDevice()
        ep = _episode_buffers(torch, 64)
        with pytest.raises(NotImplementedError, match="bookkeeping handle"):
            lean.RolloutFusedDevice(acts, T, n, 4, episodes=ep)
        with pytest.raises(ValueError):
            lean.RolloutFusedDevice(None, T, n, 4)                                  # ring source without a ring
        with pytest.raises(ValueError):
            lean.RolloutFusedDevice(acts, T, n, 4, actions="epsilon_greedy", epsilon=1.5)
        lean.RolloutFusedDevice(None, T, actions="sample", action_seed=3)            # a lean handle may sample: no ring at all
        lean.Sync()
        assert lean.Tick == T + 1
    with gpu_pkg.VectorEnv("Pendulum-v1", n, seed=SEED, auto_reset=True) as p:
        with pytest.raises(NotImplementedError, match="Discrete"):
            p.RolloutFusedDevice(acts, T, n, 4, actions="epsilon_greedy", epsilon=0.1)
    with gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED, auto_reset=True, done_list=True) as d:
        ep = _episode_buffers(torch, 64)
        with pytest.raises(NotImplementedError, match="EPISODE_STATS"):
            d.RolloutFusedDevice(acts, T, n, 4, episodes=ep)                        # return / length records need the statistics
        d.ResetDevice()
        d.RolloutFusedDevice(acts, T, n, 4, episodes={k: ep[k] for k in ("step", "lane", "capacity", "count")})
        d.Sync()
        c = ep["count"].cpu().numpy()
        assert c[0] == 64 and c[1] > 64                                            # more episodes ended than the arrays hold: counted, not kept
        lanes = ep["lane"].cpu().numpy()
        assert ((0 <= lanes) & (lanes < n)).all()
    with gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED, auto_reset=True, validate_actions=True, episode_stats=True) as v:
        with pytest.raises(NotImplementedError, match="VALIDATE_ACTIONS"):
            v.RolloutFusedDevice(acts, T, n, 4)
        v.ResetDevice()
        v.RolloutFusedDevice(None, T, actions="sample")                             # sampled actions are valid by construction
        v.Sync()


def test_episode_records_with_skewed_finishing_lanes_and_a_tight_capacity(gpu_pkg):
    """inside the kernel the episode records live in 256 per-shard segments of 2 * ceil(capacity / 256) + 64 records (a wave
    appends to segment wave-index mod 256).  Lanes that finish EVENLY never fill one; here only the lanes of TWO waves of the same
    shard finish (their poles start at the edge of the angle threshold; every other lane starts upright and an alternating push keeps
    it up for the whole rollout), with a capacity sized to EXACTLY the true episode count.  Round 5 dropped the hot shard's excess;
    now it spills to the shared overflow segment and nothing is lost: count[0] == count[1] == the truth, same records as with a
    roomy capacity.  One record fewer than the truth and exactly one is dropped (counted, not kept; the kept ones are real).  The
    opt-in variant without the overflow path behaves as round 5 did."""
    import torch
    n, T = 64 * 300, 24                                  # 300 waves of 64 lanes (a batch this small runs one lane per thread)
    acts = torch.zeros((2, n), dtype=torch.int32, device="cuda")
    acts[1] = 1                                                              # alternating push: a balanced pole survives 24 steps
    torch.cuda.synchronize()
    s = np.zeros((4, n), np.float32)
    hot_lanes = np.r_[0:64, 64 * 256:64 * 257]                               # waves 0 and 256: the SAME shard (wave index mod 256)
    s[2, hot_lanes] = 0.2                                                    # theta just inside the threshold -> these poles fall at once

    def run(cap, **more):                                                    # a fresh handle per run: same seed, same ticks, same reset draws
        with gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED, auto_reset=True, episode_stats=True) as env:
            env.ResetDevice()
            env.SetState(s)
            ep = dict(_episode_buffers(torch, cap), **more)
            env.RolloutFusedDevice(acts, T, n, 2, episodes=ep)
            env.Sync()
            rec, ret, c = _records(ep)
            return rec, ret, c
    rec, ret, c = run(1 << 16)                                               # roomy: the truth
    truth = int(c[1])
    hot = int(np.isin(rec[:, 1], hot_lanes).sum())
    assert c[0] == c[1] and hot >= 128 and truth < 2 * hot                   # (nearly) every episode ends in the two hot waves
    per_shard = 2 * ((truth + 255) // 256) + 64
    assert hot > per_shard                                                   # the hot shard's own segment cannot hold them ...
    rec_t, ret_t, c_t = run(truth)                                           # ... "sized to the true episode count"
    assert c_t[0] == c_t[1] == truth and np.array_equal(rec_t, rec) and np.array_equal(ret_t, ret)     # ... and nothing is lost
    rec_s, ret_s, c_s = run(truth - 1)                                       # one short: one dropped, counted
    assert c_s[1] == truth and c_s[0] == truth - 1
    keep = {tuple(r) for r in rec.tolist()}
    assert all(tuple(r) in keep for r in rec_s.tolist())
    # GYMNET_RECORDS_NO_OVERFLOW (episodes["no_overflow"]): the 8 % faster kernel variant without the spill path — round 5's behaviour: the
    # hot shard's excess is counted, not kept; a roomy capacity keeps everything
    rec_f, ret_f, c_f = run(truth, no_overflow=True)
    assert c_f[1] == truth and per_shard <= c_f[0] < truth and all(tuple(r) in keep for r in rec_f.tolist())
    rec_g, ret_g, c_g = run(1 << 16, no_overflow=True)
    assert c_g[0] == c_g[1] == truth and np.array_equal(rec_g, rec)


def test_sampled_action_rollout_at_2p20_lanes_equals_the_oracle_replay(gpu_pkg, oracle):
    """BASELINE's batch: 2^20 CartPole lanes, 12 steps in one launch with the actions drawn in the kernel, the episode statistics
    and a 9-step time limit, replayed on the CPU: oracle Discrete.Sample() words -> oracle step -> the bookkeeping in NumPy."""
    import torch
    n, T, limit, tick0 = 1 << 20, 12, 9, 5
    with gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED, auto_reset=True, episode_stats=True, max_episode_steps=limit) as env:
        env.ResetDevice()
        ep = _episode_buffers(torch, n * 3)
        rec_a = torch.zeros((T, n), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        env.RolloutFusedDevice(None, T, actions="sample", action_seed=ASEED, action_tick0=tick0, rec_actions=rec_a, episodes=ep)
        env.Sync()
        s = oracle.env_reset("CartPole-v1", SEED, 0, 0, n)
        ln, ret = np.zeros(n, np.int32), np.zeros(n, np.float32)
        want = []
        got_a = rec_a.cpu().numpy()
        for t in range(T):
            a = oracle.discrete_sample(ASEED, 0, tick0 + t, 2, 0, n)
            assert np.array_equal(got_a[t], a), t
            stepped, _, r, d = oracle.env_step("CartPole-v1", s, a, dtype=np.float32)
            ln += 1; ret += r
            fin = d.astype(bool) | (ln >= limit)
            fresh = oracle.env_reset("CartPole-v1", SEED, 0, 1 + t, n)
            s = np.where(fin, fresh, stepped)
            lanes = np.nonzero(fin)[0]
            want.append(np.stack([np.full(len(lanes), t), lanes, ln[fin]], axis=1))
            ln[fin] = 0; ret[fin] = 0.0
        assert np.array_equal(env.GetState(), s)
        assert np.array_equal(env.GetArray("episode_length"), ln) and np.array_equal(env.GetArray("episode_return"), ret)
        got_rec, got_ret, counts = _records(ep)
        want = np.concatenate(want)
        assert counts[0] == counts[1] == len(want) and np.array_equal(got_rec, want)
        assert np.array_equal(got_ret, want[:, 2].astype(np.float32))              # CartPole: reward 1 per step, return == length


@pytest.mark.parametrize("name,dtype,actions,lane_offset,launch", [
    ("CartPole-v1", np.float32, "epsilon_greedy", 8, {"vec": 4}),    # four lanes per thread sharing ONE call per stream (words A and B)
    ("CartPole-v1", np.float32, "epsilon_greedy", 7, {"vec": 4}),    # a shard that starts INSIDE a group of four lanes: the per-lane word path
    ("CartPole-v1", np.float32, "sample", 3, None),                  # one lane per thread (the default at this size), word (L & 3) of its group's call
    ("CartPole-v1", np.float64, "sample", 0, None),                  # the reference-arithmetic handle: its twin and its two-call reset draws
    ("CartPole-v1", np.float64, "epsilon_greedy", (1 << 20) + 2, None),
    ("Acrobot-v1", np.float32, "sample", 0, None),                   # derived observation, three actions
    ("Acrobot-v1", np.float32, "epsilon_greedy", 6, {"vec": 2}),     # two lanes per thread (the packed form), lane offset 6: word base 2
])
def test_bookkeeping_rollout_variants_equal_the_oracle_replay(gpu_pkg, oracle, name, dtype, actions, lane_offset, launch):
    """the fused bookkeeping rollout's remaining variants replayed on the CPU — not against the stepwise HIP path but
    against the oracle: ActionSpace.Sample() / the epsilon-greedy composer from the oracle's action-stream words (TrainingPlaySession.cs:
    46-52, Discrete.cs:17-28), the step from the oracle's kernel-semantics restatement (float32 twin; the float64 twin for the F64
    handle; Acrobot's RK4 with its derived observation), the fused reset from the oracle's Philox draws (two calls per lane for F64),
    the episode bookkeeping (return, length, a 13-step time limit) in NumPy.  2^16 lanes, 40 steps in ONE launch: the state, the
    running statistics, the actions taken, the last step's observations and every (t, lane, return, length) record must match."""
    import torch
    n, T, limit, tick0, eps, ring = 1 << 16, 40, 13, 77, 0.35, 5
    f64 = dtype == np.float64
    nvals = 2 if name.startswith("CartPole") else 3
    rng = np.random.default_rng(5)
    policy = rng.integers(0, nvals, (ring, n)).astype(np.int32)
    with gpu_pkg.VectorEnv(name, n, seed=SEED, auto_reset=True, dtype=dtype, lane_offset=lane_offset, episode_stats=True, max_episode_steps=limit,
                           launch_policy=launch) as env:
        env.ResetDevice()
        D = env.ObsDim
        ep = _episode_buffers(torch, n * 8)
        rec_a = torch.zeros((T, n), dtype=torch.int32, device="cuda")
        rec_o = torch.zeros((T, D, n), dtype=torch.float64 if f64 else torch.float32, device="cuda")
        pol_d = torch.from_numpy(policy).cuda()
        torch.cuda.synchronize()
        env.RolloutFusedDevice(pol_d if actions == "epsilon_greedy" else None, T, n, ring, actions=actions, action_seed=ASEED, action_tick0=tick0,
                               epsilon=eps, rec_actions=rec_a, rec_obs=rec_o, episodes=ep)
        env.Sync()
        # ---- the replay ----
        if f64:
            s = oracle.cartpole_reset_f64(SEED, lane_offset, 0, n)
        else:
            s = oracle.env_reset(name, SEED, lane_offset, 0, n)
        ln, ret = np.zeros(n, np.int32), np.zeros(n, np.float32)
        want, want_ret = [], []
        got_a = rec_a.cpu().numpy()
        last_obs = None
        for t in range(T):
            if actions == "sample":
                a = oracle.discrete_sample(ASEED, lane_offset, tick0 + t, nvals, 0, n)
            else:
                a = oracle.compose_discrete(ASEED, lane_offset, tick0 + t, nvals, eps, policy[t % ring])
            assert np.array_equal(got_a[t], a), t
            if f64:
                stepped, r, d, _ = oracle.cartpole_step(s, a, dtype=np.float64, kernel_sincos=True)
                obs = stepped
                fresh = oracle.cartpole_reset_f64(SEED, lane_offset, 1 + t, n)
                fresh_obs = fresh
            else:
                stepped, obs, r, d = oracle.env_step(name, s, a, dtype=np.float32)
                fresh, fresh_obs = oracle.env_reset(name, SEED, lane_offset, 1 + t, n, with_obs=True)
            ln += 1
            ret += r.astype(np.float32)
            fin = d.astype(bool) | (ln >= limit)
            s = np.where(fin, fresh, stepped)
            last_obs = np.where(fin, fresh_obs, obs)
            lanes = np.nonzero(fin)[0]
            want.append(np.stack([np.full(len(lanes), t), lanes, ln[fin]], axis=1))
            want_ret.append(ret[fin].copy())
            ln[fin] = 0; ret[fin] = 0.0
        assert np.array_equal(env.GetState(), s)
        assert np.array_equal(rec_o[T - 1].cpu().numpy(), last_obs)                  # the observation AFTER the last step (fresh for reset lanes)
        assert np.array_equal(env.GetArray("episode_length"), ln) and np.array_equal(env.GetArray("episode_return"), ret)
        got_rec, got_ret, counts = _records(ep)
        want = np.concatenate(want)
        assert counts[0] == counts[1] == len(want) > n and np.array_equal(got_rec, want)
        assert np.array_equal(got_ret, np.concatenate(want_ret))
        explored = (got_a != policy[np.arange(T) % ring]).mean() if actions == "epsilon_greedy" else None
        if explored is not None:
            assert abs(explored - eps * (1 - 1 / nvals)) < 0.01                      # explored AND drew a different action
