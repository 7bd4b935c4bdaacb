"""GPU tests of GYMNET_FLAG_F64 as a FIRST-CLASS mode: the reference's own arithmetic — float64 state that is the
float64 observation, CartPoleEnv.cs:141-166,185 — through every path the float32 engine has, because both are instantiations of
ONE kernel skeleton (gym.net_amd/csrc/step_kernels.hpp): DONE_LIST / EPISODE_STATS / FINAL_OBS / COMPACT_RECORDS_ONLY records,
DOUBLE_BUFFER, external observation buffers (aligned and not), the wave-compacted fused reset, groups of logical members with the
direct all-gather (incl. BASELINE config 5's shape, 8 x 2^20 lanes, overlapped), ShardedVectorEnv.  Bars: bit-identical to the
float64 twin of the oracle / to a plain single handle; the sharded batch equals the one-handle batch bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SEED = 0x5EED


def _twin_step(oracle, seed, off, tick, s, a, ln, ret, limit=0, lane_seed=None):
    """One vector step of the float64 twin with the engine's bookkeeping: returns the new state and this step's records."""
    stepped, r, d = oracle.cartpole_step(s, a, dtype=np.float64, kernel_sincos=True)[:3]
    ln += 1
    ret += r
    trunc = (ln >= limit) if limit else np.zeros_like(d, bool)
    fin = d.astype(bool) | trunc
    fresh = oracle.cartpole_reset_f64(seed, off, tick, s.shape[1], lane_seed=lane_seed)
    new = np.where(fin, fresh, stepped)
    rec = {"lanes": np.nonzero(fin)[0], "final_obs": stepped[:, fin].T.copy(), "return": ret[fin].copy(), "length": ln[fin].copy(),
           "done": d.astype(np.uint8) | (trunc.astype(np.uint8) << 1)}
    ln[fin] = 0
    ret[fin] = 0.0
    return new, r, rec


@pytest.mark.parametrize("compact_only", [False, True])
@pytest.mark.parametrize("n,vec", [(4096 + 5, 2), (3001, 1)])
def test_f64_done_list_records_and_final_obs_equal_the_twin(gpu_pkg, oracle, n, vec, compact_only):
    """Wave-ballot done compaction, compact (lane, return, length, terminal observation) records and the dense views of a float64
    handle, every step, against the twin: the records' terminal observations are the PRE-reset float64 states bit for bit."""
    off, limit = 1_000_000_007, 11
    rng = np.random.default_rng(n)
    with gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED, auto_reset=True, lane_offset=off, dtype=np.float64, done_list=True, episode_stats=True,
                           final_obs=True, max_episode_steps=limit, compact_records_only=compact_only, launch_policy={"vec": vec}) as env:
        assert env.KernelName() == f"step_kernel<CartPole64,{vec},true,true,15,{1 if vec == 2 else 0}>"   # wide lanes: wave-compacted reset
        s = env.Reset().T.copy()
        assert np.array_equal(s, oracle.cartpole_reset_f64(SEED, off, 0, n))
        ln, ret = np.zeros(n, np.int32), np.zeros(n, np.float32)
        dense_len, dense_ret, dense_obs = np.zeros(n, np.int32), np.zeros(n, np.float32), np.zeros((n, 4))
        total = 0
        for t in range(30):
            a = rng.integers(0, 2, n).astype(np.int32)
            tick = env.Tick
            out = env.Step(a)
            s, r, rec = _twin_step(oracle, SEED, off, tick, s, a, ln, ret, limit)
            assert np.array_equal(out.Observation.T, s) and np.array_equal(out.Reward, r), t
            assert np.array_equal(out.Done, rec["done"] != 0) and np.array_equal(out.Truncated, (rec["done"] & 2) != 0)
            got = env.DoneRecords()
            order = np.argsort(got["lanes"])
            assert np.array_equal(got["lanes"][order], rec["lanes"]), t
            assert got["final_obs"].dtype == np.float64 and np.array_equal(got["final_obs"][order], rec["final_obs"]), t
            assert np.array_equal(got["return"][order], rec["return"]) and np.array_equal(got["length"][order], rec["length"])
            assert env.Counters()["last_done_count"] == len(rec["lanes"])
            assert np.array_equal(np.sort(env.DoneLanes()), rec["lanes"])
            dense_len[rec["lanes"]] = rec["length"]; dense_ret[rec["lanes"]] = rec["return"]; dense_obs[rec["lanes"]] = rec["final_obs"]
            total += len(rec["lanes"])
            if compact_only or t % 7 == 6:          # compact-only: the dense views follow only the steps whose getters are called
                fr, fl = env.EpisodeStats()
                fo = env.FinalObs()
                assert fo.dtype == np.float64
                assert np.array_equal(fl, dense_len) and np.array_equal(fr, dense_ret) and np.array_equal(fo, dense_obs), t
        assert total > n // 2
        assert np.array_equal(env.GetArray("final_obs"), dense_obs.T) and env.GetArray("final_obs").dtype == np.float64
        # checkpoint / restore of the float64 configuration with every array
        ck = env.Checkpoint()
        acts = rng.integers(0, 2, (5, n)).astype(np.int32)
        want = [env.Step(acts[t]).Observation.copy() for t in range(5)]
        env.Restore(ck)
        for t in range(5):
            assert np.array_equal(env.Step(acts[t]).Observation, want[t])
        # float64 state through VecEnv.get_attr / set_attr is the identity (set_attr used to round to float32)
        st = env.get_attr("state")
        assert st.dtype == np.float64 and st.shape == (n, 4)
        env.set_attr("state", st)
        assert np.array_equal(env.GetState(), st.T)


def test_f64_double_buffer_wave_reset_and_launch_forms_are_bit_identical(gpu_pkg):
    """The float64 instantiations of every launch form — lanes per thread 1 / 2, workgroups of 64 / 256, wave-compacted fused
    reset, ping-ponged state buffers (one-launch steps, graph replay, fused rollout) — against the plain handle."""
    import torch
    n, ring = 2 * 256 * 9 + 6, 6
    acts = torch.randint(0, 2, (ring, n), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    res = {}
    # "plain" runs the default: two lanes per thread, wave-compacted reset with two lanes per reset (reset_form 1)
    forms = {"plain": {}, "vec1": {"launch_policy": {"vec": 1}}, "block64": {"launch_policy": {"block": 64, "nt": 12, "reset_form": 0}},
             "drain_reset": {"launch_policy": {"reset_form": 0}}, "db": {"double_buffer": True},
             "db_graph": {"double_buffer": True, "launch_policy": {"graph": 1, "reset_form": 0}}}
    for name, kw in forms.items():
        with gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED, auto_reset=True, dtype=np.float64, **kw) as env:
            if name == "plain":
                assert env.KernelName() == "step_kernel<CartPole64,2,true,false,15,1>"
            if name == "drain_reset":
                assert env.KernelName() == "step_kernel<CartPole64,2,true,false,15,0>"
            if name == "block64":
                assert env.GetLaunchPolicy()["block"] == 64 and env.KernelName() == "step_kernel<CartPole64,2,true,false,12,0>"
            env.ResetDevice()
            env.StepDevice(acts[0])
            if name.startswith("db"):
                assert env.ObsBufferIndex() == 1 and env.DeviceView().d_obs_alt
            env.RolloutDevice(acts, 4 * ring, n, ring)
            env.RolloutFusedDevice(acts, 9, n, ring)
            env.StepDevice(acts[1])
            env.Sync()
            r = env.Read()
            assert r.Observation.dtype == np.float64
            res[name] = (env.GetState(), r.Observation, r.Reward, r.Done, env.Tick)
    assert res["plain"][3].any()
    for name in forms:
        assert all(np.array_equal(u, v) for u, v in zip(res["plain"], res[name])), name


@pytest.mark.parametrize("misalign", [0, 1])
def test_f64_external_observation_buffers(gpu_pkg, oracle, misalign):
    """gymnet_config.d_ext_obs on a float64 handle: the LIVE state lives in the caller's buffer of doubles (zero-copy for a GPU
    policy / an all-gather slice); a buffer that is only 8-byte aligned runs one lane per thread and computes the same bits."""
    import torch
    n, stride = 5000, 5008
    buf = torch.zeros(4 * stride + 2, dtype=torch.float64, device="cuda")
    alt = torch.zeros(4 * stride + 2, dtype=torch.float64, device="cuda")
    ext, ext2 = buf[misalign:misalign + 4 * stride].view(4, stride), alt[misalign:misalign + 4 * stride].view(4, stride)
    torch.cuda.synchronize()
    rng = np.random.default_rng(8)
    with gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED, auto_reset=True, dtype=np.float64, ext_obs=ext.data_ptr(), ext_obs_stride=stride,
                           double_buffer=True, ext_obs_alt=ext2.data_ptr(), stream=torch.cuda.current_stream().cuda_stream) as env:
        assert env.KernelName() == f"step_kernel<CartPole64,{1 if misalign else 2},true,false,15,{0 if misalign else 1}>"
        if misalign:
            with pytest.raises(ValueError):
                env.SetLaunchPolicy(vec=2)                      # 16-byte accesses on an 8-byte aligned row
        env.ResetDevice()
        env.Sync()
        s = oracle.cartpole_reset_f64(SEED, 0, 0, n)
        assert np.array_equal(ext[:, :n].cpu().numpy(), s)
        bufs = (ext, ext2)
        for t in range(9):
            a = rng.integers(0, 2, n).astype(np.int32)
            tick = env.Tick
            out = env.Step(a)
            s, r, d = oracle.cartpole_autoreset_step_f64(SEED, 0, tick, s, a)
            assert np.array_equal(out.Observation.T, s) and np.array_equal(out.Done, d.astype(bool)), t
            assert np.array_equal(bufs[env.ObsBufferIndex()][:, :n].cpu().numpy(), s), t        # the caller's buffer IS the state
    with pytest.raises(ValueError):
        gpu_pkg.VectorEnv("CartPole-v1", n, dtype=np.float64, ext_obs=ext.data_ptr() + 4, ext_obs_stride=stride)   # not even element-aligned


@pytest.mark.parametrize("overlap", [False, True])
def test_f64_group_sharding_invariance_with_direct_allgather(gpu_pkg, overlap):
    """G logical members in float64 (replicas of doubles, rank-major [G][4][N/G]) == ONE float64 handle, bit for bit, after the
    reset and after every step; with the double-buffered overlap a gather stays in flight across the next step."""
    import torch
    G, n = 4, 4 * 1024 + 8
    rng = np.random.default_rng(5)
    with gpu_pkg.GroupVectorEnv("CartPole-v1", n, G, devices=[0] * G, seed=SEED, auto_reset=True, gather="direct", overlap=overlap,
                                dtype=np.float64) as grp, \
            gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED, auto_reset=True, dtype=np.float64) as one:
        nl = grp.LanesPerMember
        assert all(m.Dtype == np.float64 and m.KernelName().startswith("step_kernel<CartPole64,") for m in grp.Members)
        one.Reset()
        grp.ResetDevice()
        grp.AllGatherObs(); grp.WaitGather(); grp.Sync()
        want = one.Read().Observation
        for m in range(G):
            rep = grp.ReadReplica(m)
            assert rep.dtype == np.float64 and np.array_equal(np.concatenate(list(rep), axis=1).T, want), ("reset", m)
        acts = torch.empty(n, dtype=torch.int32, device="cuda")
        pending = None
        for t in range(14):
            a = rng.integers(0, 2, n).astype(np.int32)
            acts.copy_(torch.from_numpy(a)); torch.cuda.synchronize()
            grp.StepDevice([acts[m * nl:(m + 1) * nl] for m in range(G)])
            grp.AllGatherObs()
            out = one.Step(a)
            if overlap and t % 2 == 0:
                continue                                                 # this gather is still in flight during the next step
            grp.WaitGather(); grp.Sync()
            for m in range(G):
                rep = grp.ReadReplica(m)
                assert np.array_equal(np.concatenate(list(rep), axis=1).T, out.Observation), (t, m)
            r = grp.Members[G - 1].Read()
            assert np.array_equal(r.Reward, out.Reward[(G - 1) * nl:]) and np.array_equal(r.Done, out.Done[(G - 1) * nl:])
    # the host-boundary forms of a float64 group carry doubles
    with gpu_pkg.GroupVectorEnv("CartPole-v1", 10_000, 2, devices=[0, 0], seed=SEED, auto_reset=True, gather="none", dtype=np.float64) as grp, \
            gpu_pkg.VectorEnv("CartPole-v1", 10_000, seed=SEED, auto_reset=True, dtype=np.float64) as one:
        a0, b0 = grp.Reset(), one.Reset()
        assert a0.dtype == np.float64 and np.array_equal(a0, b0)
        for t in range(8):
            a = rng.integers(0, 2, 10_000).astype(np.int32)
            g, o = grp.Step(a), one.Step(a)
            assert np.array_equal(g.Observation, o.Observation) and np.array_equal(g.Reward, o.Reward) and np.array_equal(g.Done, o.Done)
    # RCCL variant at one member: ncclDouble in place
    with gpu_pkg.GroupVectorEnv("CartPole-v1", 4096, 1, devices=[0], seed=SEED, auto_reset=True, gather="rccl", dtype=np.float64) as grp, \
            gpu_pkg.VectorEnv("CartPole-v1", 4096, seed=SEED, auto_reset=True, dtype=np.float64) as one:
        grp.ResetDevice(); one.Reset()
        grp.AllGatherObs(); grp.WaitGather(); grp.Sync()
        assert np.array_equal(grp.ReadReplica(0)[0].T, one.Read().Observation)


def test_f64_config5_shape_eight_members_of_2p20_lanes_overlapped_gather(gpu_pkg, oracle):
    """BASELINE config 5's shape in the reference's arithmetic: 8 members x 2^20 float64 lanes (2^23 lanes, 32 MiB slice per
    member, 256 MiB replica per member and buffer), direct all-gather overlapped with the next step — on ONE device here (the
    pool has no second GPU).  Member 7's replica after the rollout equals the float64 twin replayed on the CPU for a sample of
    lanes of every member, and every member's replica holds the same bytes."""
    import torch
    G, nl, steps, ring = 8, 1 << 20, 6, 4
    n = G * nl
    free, _ = torch.cuda.mem_get_info()
    if free < 12 * (1 << 30):
        pytest.skip("needs ~6 GiB of device memory")
    with gpu_pkg.GroupVectorEnv("CartPole-v1", n, G, devices=[0] * G, seed=SEED, auto_reset=True, gather="direct", overlap=True,
                                dtype=np.float64) as grp:
        assert grp.Members[0].KernelName() == "step_kernel_pipe2<CartPole64,4,true,15>"
        acts = torch.empty((ring, n), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        for t in range(ring):
            for m in range(G):
                grp.Members[m].SampleActionsDevice(acts[t, m * nl:(m + 1) * nl], seed=SEED + 1, tick=t)
        grp.ResetDevice()
        for t in range(steps):
            grp.StepDevice([acts[t % ring, m * nl:(m + 1) * nl] for m in range(G)])
            grp.AllGatherObs()                                   # overlapped: in flight during the next step
        grp.WaitGather(); grp.Sync()
        a_host = acts.cpu().numpy()
        rep7 = grp.ReadReplica(7)                                # [8, 4, 2^20] doubles
        sample = 4096
        for m in range(G):
            off = m * nl
            s = oracle.cartpole_reset_f64(SEED, off, 0, sample)
            for t in range(steps):
                s, _, _ = oracle.cartpole_autoreset_step_f64(SEED, off, 1 + t, s, a_host[t % ring, off:off + sample])
            assert np.array_equal(rep7[m][:, :sample], s), m
        rep0 = grp.ReadReplica(0)
        assert np.array_equal(rep0, rep7)


SHARDED_F64_CHILD = r"""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import torch, torch.distributed as dist
import __graft_entry__ as ge
pkg = ge.load_package()
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "%d")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)       # RCCL, world of one
stream = torch.cuda.Stream(dev); torch.cuda.set_stream(stream)
n = 1 << 14
for overlap in (False, True):
    env = pkg.ShardedVectorEnv("CartPole-v1", n, rank=0, world_size=1, device=0, seed=0x5EED, auto_reset=True, gather_obs=True,
                               tensor_device=dev, force_gather=True, overlap=overlap, dtype="float64")
    one = pkg.VectorEnv("CartPole-v1", n, seed=0x5EED, auto_reset=True, dtype=np.float64)
    assert env.obs_bufs.dtype == torch.float64 and env.local.Dtype == np.float64 and env.overlap == overlap
    rng = np.random.default_rng(1)
    env.ResetDevice(); one.Reset()
    acts = torch.empty(n, dtype=torch.int32, device=dev)
    for t in range(16):
        a = rng.integers(0, 2, n).astype(np.int32)
        acts.copy_(torch.from_numpy(a)); torch.cuda.synchronize()
        env.StepDevice(acts)
        env.AllGatherObs(overlap=overlap)
        want = one.Step(a).Observation
        if overlap and t %% 2 == 0 and t + 1 < 16:
            continue                                   # gather stays in flight across the next step
        env.WaitGather(); env.Sync(); torch.cuda.synchronize()
        got = env.GlobalObs().cpu().numpy()            # [G=1, 4, n] doubles
        assert got.dtype == np.float64 and np.array_equal(got[0].T, want), (overlap, t)
    env.Close(); one.Close()
dist.destroy_process_group()
print("SHARDED_F64_OK")
"""


def test_f64_sharded_vector_env_on_the_hip_engine(gpu_pkg):
    """ShardedVectorEnv(dtype="float64") on the HIP engine: gather buffers of doubles the rank's state lives in, RCCL all-gather
    (world of one on this pool), serial and overlapped — equal to a plain float64 handle bit for bit.  Child process: it
    initialises torch.distributed."""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", SHARDED_F64_CHILD % port], cwd=root, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "SHARDED_F64_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_f64_pair_kernel_workgroup_sizes_and_reset_forms_are_bit_identical(gpu_pkg, oracle):
    """The float64 multi-item kernel (k lane pairs per thread) at every workgroup size, and the one-shot kernel with the
    wave-compacted fused reset, against the float64 twin at a lane offset above 2^32.  (Lane quads, a per-item compacted reset and a
    deferred single reset pass were measured in round 5 and removed: profiles/f64_forms_r05.txt.)"""
    import torch
    n, ring, steps, off = 2 * 256 * 8 * 3, 6, 40, (1 << 33) + 12_288
    acts = torch.randint(0, 2, (ring, n), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    a_host = acts.cpu().numpy()
    s = oracle.cartpole_reset_f64(SEED, off, 0, n)
    for t in range(steps):
        s, r, d = oracle.cartpole_autoreset_step_f64(SEED, off, 1 + t, s, a_host[t % ring])
    forms = [(1, 1, 256, "step_kernel<CartPole64,2,true,false,15,1>"), (1, 0, 64, "step_kernel<CartPole64,2,true,false,15,0>"),
             (2, 0, 256, "step_kernel_pipe2<CartPole64,2,true,15>"), (3, 0, 128, "step_kernel_pipe2<CartPole64,3,true,15>"),
             (4, 0, 64, "step_kernel_pipe2<CartPole64,4,true,15>"), (4, 1, 256, "step_kernel_pipe2<CartPole64,4,true,15>")]
    for items, rf, block, want in forms:
        pol = {"sequential_lanes": items, "reset_form": rf, "block": block}
        with gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED, auto_reset=True, dtype=np.float64, lane_offset=off, launch_policy=pol) as env:
            assert env.KernelName() == want, env.KernelName()
            env.ResetDevice()
            env.RolloutDevice(acts, steps, n, ring)
            env.Sync()
            out = env.Read()
            assert np.array_equal(env.GetState(), s) and np.array_equal(out.Reward, r) and np.array_equal(out.Done, d.astype(bool)), want
    with gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED, auto_reset=True, dtype=np.float64) as env:
        for bad in (dict(vec=4), dict(sequential_lanes=8), dict(sequential_lanes=5)):
            with pytest.raises(ValueError):
                env.SetLaunchPolicy(**bad)


@pytest.mark.parametrize("env_id,dtype,n,want", [
    ("MountainCar-v0", "float32", 1 << 19, "step_kernel<MountainCar,1,true,false,15,0>"),          # one resident generation: scalar lanes
    ("MountainCar-v0", "float32", (1 << 19) + 64, "step_kernel<MountainCar,4,true,false,15,1>"),   # beyond it: 16-byte lanes (was scalar up to 24 MiB)
    ("MountainCar-v0", "float32", 1_000_000, "step_kernel<MountainCar,4,true,false,15,1>"),
    ("CartPole-v1", "float32", 1 << 19, "step_kernel<CartPole,1,true,false,15,0>"),
    ("CartPole-v1", "float32", 3 << 18, "step_kernel<CartPole,4,true,false,15,1>"),
    ("Acrobot-v1", "float32", 1 << 20, "step_kernel_pipe<Acrobot,4,true,15>"),
    ("Acrobot-v1", "float32", (1 << 20) + 1, "step_kernel_pipe<Acrobot,4,true,15>"),               # rounded, not rounded up (was 5 lanes per thread)
    ("Acrobot-v1", "float32", 5 << 18, "step_kernel_pipe<Acrobot,5,true,15>"),
    ("CartPole-v1", "float64", 1 << 20, "step_kernel_pipe2<CartPole64,4,true,15>"),
    ("CartPole-v1", "float64", 1_000_000, "step_kernel_pipe2<CartPole64,4,true,15>"),               # any batch size just under two waves per SIMD
    ("CartPole-v1", "float64", (1 << 20) + 2, "step_kernel<CartPole64,2,true,false,15,1>"),         # one wave more would be a second round
    ("CartPole-v1", "float64", 3 << 18, "step_kernel_pipe2<CartPole64,2,true,15>"),
    ("CartPole-v1", "float64", 1 << 19, "step_kernel<CartPole64,2,true,false,15,1>"),
    # round 6 (profiles/trough_r06.txt, f64_sizes_r06.txt): no stream non-temporal while a step's footprint fits the Infinity Cache
    ("CartPole-v1", "float32", 5 << 18, "step_kernel<CartPole,4,true,false,0,1>"),
    ("CartPole-v1", "float32", 1 << 22, "step_kernel<CartPole,4,true,false,0,1>"),
    ("CartPole-v1", "float32", 1 << 23, "step_kernel<CartPole,4,true,false,12,1>"),                 # 328 MiB per step: the state alone stays cacheable
    ("MountainCar-v0", "float32", 3 << 19, "step_kernel<MountainCar,4,true,false,0,1>"),
    ("Pendulum-v1", "float32", 1 << 21, "step_kernel<Pendulum,4,true,false,12,0>"),                 # its write-only rows make it start later
    ("Pendulum-v1", "float32", 1 << 22, "step_kernel<Pendulum,4,true,false,0,0>"),
    ("CartPole-v1", "float64", 3 << 19, "step_kernel<CartPole64,2,true,false,0,1>"),                # (round 5 let FOUR pairs through at three waves per SIMD)
    ("CartPole-v1", "float64", 1 << 21, "step_kernel<CartPole64,2,true,false,0,1>"),
    ("CartPole-v1", "float64", 1 << 23, "step_kernel<CartPole64,2,true,false,15,1>"),               # 584 MiB per step: nothing can stay
])
def test_default_launch_policy_by_batch_size(gpu_pkg, env_id, dtype, n, want):
    """The launch policy's rules, as the kernel the library itself names (profiles/small_batches_r05.txt, ragged_r05.txt, f64_sizes_r05.txt;
    round 6: trough_r06.txt, f64_sizes_r06.txt)."""
    kw = {"dtype": np.float64} if dtype == "float64" else {}
    with gpu_pkg.VectorEnv(env_id, n, seed=SEED, auto_reset=True, **kw) as env:
        assert env.KernelName() == want
