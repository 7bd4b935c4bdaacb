"""GPU tests for the C-ABI entry points the parity suites do not reach: stand-alone space sampling (all four
Box regimes of src/Gym/Spaces/Box.cs:69-90), device-side observation packing, the device view, per-lane seeds
with fused auto-reset, external observation buffers for envs whose observation is derived."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SEED = 0x5EED


def test_standalone_sampling_regimes(gpu_pkg, oracle):
    import torch
    lib, capi = gpu_pkg.load_library(), gpu_pkg._capi
    n = 400_000
    out = torch.empty(n, dtype=torch.float32, device="cuda")
    iout = torch.empty(n, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    inf = float("inf")

    def box(low, high, tick):
        capi.check(lib.gymnet_sample_box_device(0, None, C.c_void_p(out.data_ptr()), n, low, high, 7, 11, tick))
        torch.cuda.synchronize()
        return out.cpu().numpy().astype(np.float64)

    u = box(-5.0, 5.0, 1)                                               # bounded: uniform(low, high)   (BoxTest.cs:36-41)
    assert u.min() >= -5.0 and u.max() <= 5.0 and abs(u.mean()) < 0.03 and abs(u.std() - 10 / np.sqrt(12)) < 0.03
    assert np.array_equal(u.astype(np.float32), oracle.box_uniform_sample(7, 11, 1, -5.0, 5.0, n))
    e = box(2.0, inf, 2)                                                # low-bounded: low + Exp(1)     (Box.cs:83)
    assert e.min() >= 2.0 and abs(e.mean() - 3.0) < 0.02 and abs(e.std() - 1.0) < 0.03
    h = box(-inf, 7.0, 3)                                               # high-bounded: high + Exp(1)   (Box.cs:84, sic)
    assert h.min() >= 7.0 and abs(h.mean() - 8.0) < 0.02
    g = box(-inf, inf, 4)                                               # unbounded: Normal(0.5, 1)     (Box.cs:82, sic)
    assert abs(g.mean() - 0.5) < 0.01 and abs(g.std() - 1.0) < 0.01 and abs(((g - 0.5) ** 3).mean()) < 0.03
    capi.check(lib.gymnet_sample_discrete_device(0, None, C.c_void_p(iout.data_ptr()), n, 3, 10, 7, 11, 5))
    torch.cuda.synchronize()
    d = iout.cpu().numpy()
    assert np.array_equal(d, oracle.discrete_sample(7, 11, 5, 3, 10, n))     # Start + randint(0, N) (Discrete.cs:27)
    assert set(np.unique(d)) == {10, 11, 12} and abs(np.bincount(d - 10) / n - 1 / 3).max() < 0.005
    with pytest.raises(ValueError):
        capi.check(lib.gymnet_sample_box_device(0, None, C.c_void_p(out.data_ptr()), n, 3.0, 1.0, 0, 0, 0))
    # Box(NDArray low, NDArray high) (Box.cs:25-51): every element picks its own regime — CartPole's ObservationSpace has two
    # bounded and two (float.MaxValue-)bounded components; here one element of each of the four regimes
    m = 100_000
    lo = torch.tensor([-5.0, 2.0, -inf, -inf, -1.0], dtype=torch.float32, device="cuda")
    hi = torch.tensor([5.0, inf, 7.0, inf, 1.0], dtype=torch.float32, device="cuda")
    rows = torch.empty((m, 5), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    capi.check(lib.gymnet_sample_box_elementwise_device(0, None, C.c_void_p(rows.data_ptr()), m, 5, C.c_void_p(lo.data_ptr()),
                                                        C.c_void_p(hi.data_ptr()), 7, 11, 1))
    torch.cuda.synchronize()
    r = rows.cpu().numpy().astype(np.float64)
    assert np.array_equal(r[:, 0].astype(np.float32), u[:m].astype(np.float32))       # element 0 == the scalar sampler's draws (same seed / tick)
    assert r[:, 1].min() >= 2.0 and abs(r[:, 1].mean() - 3.0) < 0.03                  # low + Exp(1)
    assert r[:, 2].min() >= 7.0 and abs(r[:, 2].mean() - 8.0) < 0.03                  # high + Exp(1) (sic)
    assert abs(r[:, 3].mean() - 0.5) < 0.02 and abs(r[:, 3].std() - 1.0) < 0.02       # Normal(0.5, 1) (sic)
    assert r[:, 4].min() >= -1.0 and r[:, 4].max() <= 1.0 and abs(r[:, 4].std() - 2 / np.sqrt(12)) < 0.01
    assert abs(np.corrcoef(r[:, 0], r[:, 4])[0, 1]) < 0.02                            # elements draw from different keys
    with pytest.raises(ValueError):
        capi.check(lib.gymnet_sample_box_elementwise_device(0, None, C.c_void_p(rows.data_ptr()), m, 0, C.c_void_p(lo.data_ptr()),
                                                            C.c_void_p(hi.data_ptr()), 7, 11, 1))


@pytest.mark.parametrize("name", ["CartPole-v1", "Pendulum-v1", "Acrobot-v1"])
def test_pack_obs_device_and_device_view(gpu_pkg, name):
    import torch
    n = 3000
    with gpu_pkg.VectorEnv(name, n, seed=SEED, auto_reset=True) as env:
        obs = env.Reset()
        D = env.ObsDim
        packed = torch.zeros((n, D), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        env.PackObsDevice(packed); env.Sync()
        assert np.array_equal(packed.cpu().numpy(), obs)                 # row-major [N, D] == the NDArray the host API returns
        v = env.DeviceView()
        assert v.num_envs == n and v.obs_dim == D and v.state_dim == env.StateDim
        assert bool(v.obs_aliases_state) == (name == "CartPole-v1") and (v.d_obs == v.d_state) == (name == "CartPole-v1")
        assert v.d_reward and v.d_done and v.stream and v.state_stride >= n and v.state_stride % 64 == 0
        out = env.Step(env.SampleActions(seed=1, tick=0))
        again = env.Read()                                               # gymnet_vecenv_read: the same Step results again
        assert np.array_equal(again.Observation, out.Observation) and np.array_equal(again.Reward, out.Reward)


def test_per_lane_seeds_with_fused_autoreset(gpu_pkg, oracle):
    # VecEnv.Seed(int[]) (VecEnv.cs:48-53): lane i resets from Philox(key = seeds[i], counter = (global lane, tick))
    n, off = 2048, 100
    seeds = (np.arange(n) * 7919 + 13).astype(np.int64)
    rng = np.random.default_rng(3)
    with gpu_pkg.VectorEnv("CartPole-v1", n, seed=1, auto_reset=True, lane_offset=off) as env:
        env.Seed(seeds.tolist())
        first = env.Reset()
        want = np.stack([oracle.cartpole_reset(int(seeds[i]), off + i, 0, 1)[:, 0] for i in range(0, n, 37)])
        assert np.array_equal(first[::37], want)
        for t in range(30):
            tick = env.Tick
            out = env.Step(rng.integers(0, 2, n).astype(np.int32))
            for i in np.nonzero(out.Done)[0][:5]:
                assert np.array_equal(out.Observation[i], oracle.cartpole_reset(int(seeds[i]), off + int(i), tick, 1)[:, 0])


def test_external_obs_buffer_for_derived_observations(gpu_pkg):
    # Pendulum's observation (cos, sin, thdot) is derived: with d_ext_obs the OBSERVATION arrays live in the caller's
    # buffer (the all-gather layout) while the state stays internal
    import torch
    n = 5000
    buf = torch.zeros((3, n), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    with gpu_pkg.VectorEnv("Pendulum-v1", n, seed=SEED, auto_reset=True, ext_obs=buf.data_ptr(), ext_obs_stride=n) as env, \
            gpu_pkg.VectorEnv("Pendulum-v1", n, seed=SEED, auto_reset=True) as ref:
        env.Reset(); ref.Reset()
        a = np.linspace(-2, 2, n).astype(np.float32)
        for t in range(5):
            o1 = env.Step(a); o2 = ref.Step(a)
            assert np.array_equal(o1.Observation, o2.Observation)
        env.Sync()
        assert np.array_equal(buf.cpu().numpy().T, o1.Observation)


def test_create_destroy_does_not_leak_device_memory(gpu_pkg):
    import torch

    def cycle():
        with gpu_pkg.VectorEnv("CartPole-v1", 1 << 18, seed=1, auto_reset=True, done_list=True, episode_stats=True, final_obs=True) as env:
            env.Reset()
            env.Step(1)
            env.Seed(list(range(1 << 18)))
            env.Step(0)
            env.DoneLanes()
        with gpu_pkg.VectorEnv("Acrobot-v1", 1000, seed=1) as env:                   # small batch: host-mapped staging too
            env.Reset(); env.Step(2)
        with gpu_pkg.VectorEnv("CartPole-v1", 1 << 14, seed=1, auto_reset=True) as env:   # graph capture / replay objects
            acts = torch.zeros((4, 1 << 14), dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()
            env.ResetDevice(); env.RolloutDevice(acts, 16, 1 << 14, 4); env.Sync()

    cycle()                                                  # first use loads code objects and warms the runtime's pools
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(30):
        cycle()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < (32 << 20), (free0, free1)      # 30 leaked handles would be ~ 1 GiB


def test_handles_are_independent_across_threads(gpu_pkg):
    """Different handles from different threads at the same time (the reference runs two env instances concurrently in
    tests/Gym.Tests/Envs/Aether/LunarLanderEnvironment.cs:185-190); one handle from two threads is refused, not raced."""
    import threading
    n, steps = 1 << 15, 60
    rng = np.random.default_rng(5)
    acts = rng.integers(0, 2, (steps, n)).astype(np.int32)

    def rollout(seed, out, k):
        with gpu_pkg.VectorEnv("CartPole-v1", n, seed=seed, auto_reset=True) as env:
            env.Reset()
            for t in range(steps):
                env.Step(acts[t])
            out[k] = env.GetState()

    solo = {}
    for k in range(4):
        rollout(100 + k, solo, k)
    conc = {}
    threads = [threading.Thread(target=rollout, args=(100 + k, conc, k)) for k in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    for k in range(4):
        assert np.array_equal(solo[k], conc[k])

    # one handle, two threads: the second caller gets AlreadySteppingError (the guard the reference only declared)
    with gpu_pkg.VectorEnv("CartPole-v1", 1 << 22, seed=1, auto_reset=True) as env:
        env.Reset()
        errors, big = [], np.ones(1 << 22, np.int32)

        def hammer():
            for _ in range(15):
                try:
                    env.Step(big)
                except gpu_pkg.AlreadySteppingError:
                    errors.append(1)

        ts = [threading.Thread(target=hammer) for _ in range(3)]
        for th in ts:
            th.start()
        for th in ts:
            th.join()
        assert env.Counters()["lane_steps"] == (45 - len(errors)) * (1 << 22)    # every accepted call ran exactly once


def test_instance_properties_match_the_reference_constructor(gpu_pkg):
    # what `new CartPoleEnv()` sets up (CartPoleEnv.cs:43-52) and what VecEnv carries (VecEnv.cs:13-27)
    with gpu_pkg.VectorEnv("CartPole-v1", 8) as env:
        assert env.NumberOfEnvironments == 8 and env.Environments == []
        assert isinstance(env.ActionSpace, gpu_pkg.Discrete) and env.ActionSpace.N == 2
        assert isinstance(env.ObservationSpace, gpu_pkg.Box) and env.ObservationSpace.Shape == (4,)
        assert env.ObservationSpace.DType == np.float32
        high = env.ObservationSpace.High
        assert high[0] == np.float32(2.4) * 2 and high[1] == np.finfo(np.float32).max and np.array_equal(env.ObservationSpace.Low, -high)
        assert env.Metadata == {"render.modes": ["human", "rgb_array"], "video.frames_per_second": 50}
        obs = env.Reset()
        assert all(env.ObservationSpace.Contains(o) for o in obs)                    # every reset observation is inside the Box
        assert env.get_attr("state").shape == (8, 4) and np.array_equal(env.get_attr("steps_beyond_done"), np.full(8, -1))
        env.set_attr("state", np.zeros((8, 4), np.float32))
        assert not env.GetState().any()
        with pytest.raises(ValueError):
            env.set_attr("state", None)                                             # ArgumentNullException, VecEnv.cs:88
    cp = gpu_pkg.CartPoleEnv()
    assert cp.ActionSpace.Contains(cp.ActionSpace.Sample()) and cp.Render() is None
    cp.Dispose()


def test_batched_epsilon_greedy_composer(gpu_pkg, oracle):
    # TrainingPlaySession.ComposeAction (TrainingPlaySession.cs:46-52), one decision per lane
    import torch
    n, off = 200_000, 31
    policy = torch.full((n,), 1, dtype=torch.int32, device="cuda")
    out = torch.empty(n, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    with gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED, lane_offset=off) as env:
        for eps in (0.0, 0.25, 1.0):
            env.ComposeActionsDevice(policy, eps, out, seed=5, tick=3); env.Sync()
            got = out.cpu().numpy()
            assert np.array_equal(got, oracle.compose_discrete(5, off, 3, 2, eps, np.ones(n, np.int32)))
            if eps == 0.0:
                assert (got == 1).all()                                           # never explores (u <= 0 has probability 2^-24)
            if eps == 1.0:
                assert np.array_equal(got, env.SampleActions(seed=5, tick=3))     # always ActionSpace.Sample()
            if eps == 0.25:
                assert abs((got == 0).mean() - 0.125) < 0.005                     # explored (25 %) and drew the other action (50 %)
        with pytest.raises(ValueError):
            env.ComposeActionsDevice(policy, 1.5, out)
    with gpu_pkg.VectorEnv("Pendulum-v1", 8) as env, pytest.raises(NotImplementedError):
        env.ComposeActionsDevice(policy, 0.1, out)


@pytest.mark.parametrize("name,n", [("CartPole-v1", 1 << 18), ("CartPole-v1", 1003), ("Pendulum-v1", 40_002), ("Acrobot-v1", 70_001),
                                    ("MountainCar-v0", 4096)])
def test_pinned_host_buffers_give_the_same_results_as_caller_memory(gpu_pkg, name, n):
    """gymnet_vecenv_host_buffers (ABI 3): stepping through the library's page-locked, device-mapped buffers (one export kernel
    writes obs / reward / done across PCIe) returns exactly what the staged path returns into ordinary memory — every env
    (2-, 3-, 4- and 6-wide observations), batch sizes with ragged tails, small (host-mapped) and large batches, reset and async."""
    rng = np.random.default_rng(3)
    with gpu_pkg.VectorEnv(name, n, seed=SEED, auto_reset=True) as a, gpu_pkg.VectorEnv(name, n, seed=SEED, auto_reset=True) as b:
        pa, po, pr, pd = a.HostBuffers()
        assert a.HostBuffers()[1] is po and po.shape == (n, a.ObsDim) and pd.dtype == np.uint8
        a.ResetInto(po)
        want = b.Reset()
        assert np.array_equal(po, want)
        for t in range(6):
            act = rng.uniform(-2, 2, n).astype(np.float32) if name == "Pendulum-v1" else rng.integers(0, 3 if name != "CartPole-v1" else 2, n).astype(np.int32)
            pa[:] = act
            a.StepInto(pa, po, pr, pd)
            out = b.Step(act)
            assert np.array_equal(po, out.Observation) and np.array_equal(pr, out.Reward) and np.array_equal(pd.astype(bool), out.Done), t
        # mixing is allowed: pinned actions, ordinary outputs (staged copies)
        o2, r2, d2 = np.empty_like(po), np.empty_like(pr), np.empty_like(pd)
        a.StepInto(pa, o2, r2, d2)
        out = b.Step(np.array(pa))
        assert np.array_equal(o2, out.Observation) and np.array_equal(r2, out.Reward)
        with pytest.raises(ValueError):
            a.StepInto(pa.astype(np.int64), po, pr, pd)


def test_abi4_entry_points_reject_bad_arguments_without_side_effects(gpu_pkg):
    """Error behaviour of the ABI 4 additions, straight through ctypes: null pointers, a struct_size from another ABI, sizes that do
    not match the array, ids out of range, arrays the configuration lacks — every one a status code (never a crash), the right
    reference-style exception class in the binding, and no change to the handle."""
    import ctypes as C
    capi = gpu_pkg._capi
    lib = capi.load_library()
    n = 1000
    with gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED, auto_reset=True, episode_stats=True) as env:
        env.Reset()
        name0, pol0 = env.KernelName(), env.GetLaunchPolicy()
        assert lib.gymnet_vecenv_set_launch_policy(env._h, None) == capi.ERR_INVALID_ARG
        bad = capi.LaunchPolicy(struct_size=8, vec=1, block=-1, nt=-1, sequential_lanes=-1, reset_form=-1, lds_pipe=-1, occupancy_lds_bytes=-1, graph=-1)
        assert lib.gymnet_vecenv_set_launch_policy(env._h, C.byref(bad)) == capi.ERR_INVALID_ARG and b"ABI mismatch" in lib.gymnet_last_error()
        assert lib.gymnet_vecenv_get_launch_policy(env._h, None) == capi.ERR_INVALID_ARG
        assert lib.gymnet_vecenv_get_launch_policy(None, C.byref(bad)) == capi.ERR_INVALID_ARG
        assert env.KernelName() == name0 and env.GetLaunchPolicy() == pol0
        buf = np.zeros(n, np.float32)
        p = buf.ctypes.data_as(C.c_void_p)
        assert lib.gymnet_vecenv_get_array(env._h, capi.ARRAY_REWARD, None, n * 4) == capi.ERR_INVALID_ARG
        assert lib.gymnet_vecenv_get_array(env._h, capi.ARRAY_REWARD, p, n * 4 - 1) == capi.ERR_INVALID_ARG
        assert lib.gymnet_vecenv_get_array(env._h, 99, p, n * 4) == capi.ERR_INVALID_ARG
        assert lib.gymnet_vecenv_get_array(env._h, -1, p, n * 4) == capi.ERR_INVALID_ARG
        assert lib.gymnet_vecenv_get_array(env._h, capi.ARRAY_STEPS_BEYOND_DONE, p, n * 4) == capi.ERR_UNSUPPORTED      # auto-reset: no sbd
        assert lib.gymnet_vecenv_get_array(env._h, capi.ARRAY_FINAL_OBS, p, n * 16) == capi.ERR_UNSUPPORTED
        assert lib.gymnet_vecenv_get_array(env._h, capi.ARRAY_LANE_SEEDS, p, n * 8) == capi.ERR_UNSUPPORTED             # one key for all lanes
        assert lib.gymnet_vecenv_set_array(env._h, capi.ARRAY_EPISODE_LENGTH, None, n * 4) == capi.ERR_INVALID_ARG
        assert lib.gymnet_vecenv_set_array(env._h, capi.ARRAY_EPISODE_LENGTH, p, n * 8) == capi.ERR_INVALID_ARG
        assert lib.gymnet_vecenv_get_array(None, capi.ARRAY_REWARD, p, n * 4) == capi.ERR_INVALID_ARG
        assert lib.gymnet_vecenv_get_seed(None, None, None) == capi.ERR_INVALID_ARG
        assert lib.gymnet_vecenv_get_seed(env._h, None, None) == capi.OK                                                 # both outs optional
        assert env.GetSeed() == (SEED, False)
        with pytest.raises(NotImplementedError):
            env.GetArray("final_obs")
        assert env.GetArray("episode_length").max() == 0 and env.GetArray("done").dtype == np.uint8
        # set(LANE_SEEDS) on a handle without per-lane keys INSTALLS them, without rewinding the tick
        tick = env.Tick
        env.SetArray("lane_seeds", np.arange(n, dtype=np.uint64) + 5)
        assert env.GetSeed() == (SEED, True) and env.Tick == tick and env.KernelName().split(",")[3] == "true"
        assert np.array_equal(env.GetArray("lane_seeds"), np.arange(n, dtype=np.uint64) + 5)
    # float64 handles (ABI 5): the flag combines with every other one; CartPole only; buffers are the caller's responsibility (typed void*)
    cfg = capi.Config(struct_size=C.sizeof(capi.Config), env_id=0, num_envs=64, lane_offset=0, device=0,
                      flags=capi.FLAG_F64 | capi.FLAG_AUTORESET | capi.FLAG_DONE_LIST | capi.FLAG_FINAL_OBS | capi.FLAG_DOUBLE_BUFFER, seed=1)
    h = C.c_void_p()
    assert lib.gymnet_vecenv_create(C.byref(cfg), C.byref(h)) == capi.OK and h.value
    assert lib.gymnet_vecenv_get_array(h, capi.ARRAY_FINAL_OBS, (C.c_double * (4 * 64))(), 4 * 64 * 4) == capi.ERR_INVALID_ARG   # doubles: 4 x 64 x 8 bytes
    assert lib.gymnet_vecenv_get_array(h, capi.ARRAY_FINAL_OBS, (C.c_double * (4 * 64))(), 4 * 64 * 8) == capi.OK
    assert lib.gymnet_vecenv_destroy(h) == capi.OK
    h = C.c_void_p()
    cfg.flags = capi.FLAG_COMPACT_RECORDS_ONLY | capi.FLAG_AUTORESET
    assert lib.gymnet_vecenv_create(C.byref(cfg), C.byref(h)) == capi.ERR_INVALID_ARG and b"DONE_LIST" in lib.gymnet_last_error()
    cfg.flags, cfg.env_id = capi.FLAG_F64, 3
    assert lib.gymnet_vecenv_create(C.byref(cfg), C.byref(h)) == capi.ERR_UNSUPPORTED
    cfg.flags, cfg.env_id = capi.FLAG_AUTORESET | 0x4000, 0
    assert lib.gymnet_vecenv_create(C.byref(cfg), C.byref(h)) == capi.ERR_INVALID_ARG and b"unknown flag bits 0x4000" in lib.gymnet_last_error()
    with gpu_pkg.VectorEnv("CartPole-v1", 64, seed=1, dtype=np.float64) as e64:
        v = e64.DeviceView()
        assert v.state_dtype == capi.DTYPE_F64 and v.d_state == v.d_obs and v.d_state and not v.d_obs_alt and v.obs_aliases_state == 1
        with gpu_pkg.VectorEnv("CartPole-v1", 64, seed=1) as e32:
            assert e32.DeviceView().state_dtype == capi.DTYPE_F32
    gcfg = capi.GroupConfig(struct_size=C.sizeof(capi.GroupConfig), env_id=0, global_num_envs=128, num_members=2,
                            flags=capi.FLAG_AUTORESET | capi.FLAG_F64, seed=1, devices=(C.c_int32 * 2)(0, 0), gather=capi.GATHER_NONE, max_episode_steps=0)
    g = C.c_void_p()
    assert lib.gymnet_group_create(C.byref(gcfg), C.byref(g)) == capi.OK and g.value                     # ABI 5: float64 members
    m = C.c_void_p()
    assert lib.gymnet_group_member(g, 1, C.byref(m)) == capi.OK
    v = capi.DeviceView()
    assert lib.gymnet_vecenv_device_view(m, C.byref(v)) == capi.OK and v.state_dtype == capi.DTYPE_F64 and v.num_envs == 64
    assert lib.gymnet_group_destroy(g) == capi.OK
    gcfg.env_id = 1
    g = C.c_void_p()
    assert lib.gymnet_group_create(C.byref(gcfg), C.byref(g)) == capi.ERR_UNSUPPORTED and not g.value    # float64 arithmetic is CartPole's only
