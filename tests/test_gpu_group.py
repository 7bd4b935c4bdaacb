"""GPU tests of the multi-GPU surface on a 1-GPU box (BASELINE config 5):

  - gymnet_group_* (one process, G members): G logical members on device 0, hand-written direct all-gather, with and
    without the double-buffered overlap; every member's replica [G][D][N/G] must equal the single-handle batch bit for bit;
  - GYMNET_FLAG_DOUBLE_BUFFER on a single handle: ping-ponged observation / state buffers give bit-identical results to the
    in-place handle through one-launch steps, hipGraph replay and the fused rollout;
  - ShardedVectorEnv on the HIP engine with RCCL at world size 1 (force_gather) and `python bench.py --gpus 2` started
    plainly (it spawns its own ranks; on a 1-GPU box they share the GPU over gloo) — each in a child process;
  - the small ABI debts of round 1: Seed(int[]) reuses its buffer, reset_where(NULL) under AUTORESET is a no-op,
    the graph cache is bounded, library calls restore the caller's current device, masked Discrete.Sample on the device.
"""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED = 0x5EED
NACT = {"CartPole-v1": 2, "Pendulum-v1": 0, "MountainCar-v0": 3, "Acrobot-v1": 3}


def _actions(rng, name, n):
    if name == "Pendulum-v1":
        return rng.uniform(-2, 2, n).astype(np.float32)
    return rng.integers(0, NACT[name], n).astype(np.int32)


def _replica(gpu_pkg, grp, member):
    """Member's gathered replica as a numpy array [G, D, n]."""
    return grp.ReadReplica(member)


@pytest.mark.parametrize("name", ["CartPole-v1", "Acrobot-v1"])
@pytest.mark.parametrize("overlap", [False, True])
def test_group_of_logical_members_direct_allgather_equals_single_handle(gpu_pkg, name, overlap):
    import torch
    G, n = 4, 3 * 1024                   # 768 lanes per member
    rng = np.random.default_rng(5)
    with gpu_pkg.GroupVectorEnv(name, n, G, devices=[0] * G, seed=SEED, auto_reset=True, gather="direct", overlap=overlap) as grp, \
            gpu_pkg.VectorEnv(name, n, seed=SEED, auto_reset=True) as one:
        assert grp.LanesPerMember == n // G and len(grp.Members) == G
        nl = grp.LanesPerMember
        one.Reset()
        grp.ResetDevice()
        grp.AllGatherObs(); grp.WaitGather(); grp.Sync()
        want = one.Read().Observation                                   # [n, D]
        for m in range(G):
            rep = _replica(gpu_pkg, grp, m)                             # [G, D, nl]
            assert np.array_equal(np.concatenate(list(rep), axis=1).T, want), ("reset", m)
        acts = torch.empty(n, dtype=torch.float32 if name == "Pendulum-v1" else torch.int32, device="cuda")
        for t in range(12):
            a = _actions(rng, name, n)
            acts.copy_(torch.from_numpy(a)); torch.cuda.synchronize()
            grp.StepDevice([acts[m * nl:(m + 1) * nl] for m in range(G)])
            grp.AllGatherObs()
            if overlap and t % 2 == 0:                                  # leave the gather in flight across the next step
                pending = (t, one.Step(a))
                continue
            grp.WaitGather(); grp.Sync()
            out = one.Step(a)
            for m in range(G):
                rep = _replica(gpu_pkg, grp, m)
                assert np.array_equal(np.concatenate(list(rep), axis=1).T, out.Observation), (t, m)
            # reward / done of every member equal the corresponding slice
            for m in (0, G - 1):
                r = grp.Members[m].Read()
                assert np.array_equal(r.Reward, out.Reward[m * nl:(m + 1) * nl]) and np.array_equal(r.Done, out.Done[m * nl:(m + 1) * nl])


def test_group_host_boundary_step_equals_single_handle(gpu_pkg):
    G, n = 2, 10_000                      # > 4096 lanes per member: the memcpy path; 5000 % 4 == 0
    rng = np.random.default_rng(9)
    with gpu_pkg.GroupVectorEnv("CartPole-v1", n, G, devices=[0, 0], seed=SEED, auto_reset=True, gather="none") as grp, \
            gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED, auto_reset=True) as one:
        assert np.array_equal(grp.Reset(), one.Reset())
        for t in range(20):
            a = rng.integers(0, 2, n).astype(np.int32)
            g, o = grp.Step(a), one.Step(a)
            assert np.array_equal(g.Observation, o.Observation) and np.array_equal(g.Reward, o.Reward) and np.array_equal(g.Done, o.Done)
        g, o = grp.Step(1), one.Step(1)                                   # IVecEnv.Step(int): scalar broadcast
        assert np.array_equal(g.Observation, o.Observation)
        with pytest.raises(NotImplementedError):
            grp.AllGatherObs()                                            # created with gather="none"
    # small members take the host-mapped latency path
    with gpu_pkg.GroupVectorEnv("Pendulum-v1", 64, 4, devices=[0] * 4, seed=3, gather="direct") as grp, \
            gpu_pkg.VectorEnv("Pendulum-v1", 64, seed=3) as one:
        assert np.array_equal(grp.Reset(), one.Reset())
        a = rng.uniform(-2, 2, 64).astype(np.float32)
        g, o = grp.Step(a), one.Step(a)
        assert np.array_equal(g.Observation, o.Observation) and np.array_equal(g.Reward, o.Reward)


def test_group_argument_errors_and_rccl_variant(gpu_pkg):
    with pytest.raises(ValueError):
        gpu_pkg.GroupVectorEnv("CartPole-v1", 1001, 4, devices=[0] * 4)          # N not a multiple of G
    with pytest.raises(ValueError):
        gpu_pkg.GroupVectorEnv("CartPole-v1", 1024, 2, devices=[0, 99])         # no such device
    with pytest.raises(NotImplementedError, match="distinct GPU"):
        gpu_pkg.GroupVectorEnv("CartPole-v1", 1024, 2, devices=[0, 0], gather="rccl")
    # one member on one GPU: the RCCL code path end to end (dlopen librccl, ncclCommInitAll, in-place ncclAllGather)
    with gpu_pkg.GroupVectorEnv("CartPole-v1", 4096, 1, devices=[0], seed=SEED, auto_reset=True, gather="rccl") as grp, \
            gpu_pkg.VectorEnv("CartPole-v1", 4096, seed=SEED, auto_reset=True) as one:
        grp.ResetDevice(); one.Reset()
        grp.AllGatherObs(); grp.WaitGather(); grp.Sync()
        rep = _replica(gpu_pkg, grp, 0)
        assert np.array_equal(rep[0].T, one.Read().Observation)


@pytest.mark.parametrize("name", ["CartPole-v1", "Pendulum-v1", "MountainCar-v0", "Acrobot-v1"])
def test_double_buffered_handle_is_bit_identical_to_in_place(gpu_pkg, name, monkeypatch):
    import torch
    n, ring = 4096 + 64, 6
    # rollout_device through hipGraph replay (even-length graph)
    with gpu_pkg.VectorEnv(name, n, seed=SEED, auto_reset=True, double_buffer=True, launch_policy={"graph": 1}) as db, \
            gpu_pkg.VectorEnv(name, n, seed=SEED, auto_reset=True) as ip:
        adt = torch.float32 if name == "Pendulum-v1" else torch.int32
        acts = torch.empty((ring, n), dtype=adt, device="cuda")
        torch.cuda.synchronize()
        for t in range(ring):
            ip.SampleActionsDevice(acts[t], seed=2, tick=t)
        ip.Sync()
        assert db.ObsBufferIndex() == 0 and db.DeviceView().d_obs_alt and not ip.DeviceView().d_obs_alt
        for e in (db, ip):
            e.ResetDevice()
            e.StepDevice(acts[0])                             # one eager step: buffer index flips
        assert db.ObsBufferIndex() == 1 and ip.ObsBufferIndex() == 0
        for e in (db, ip):
            e.RolloutDevice(acts, 3 * ring, n, ring)          # graph replay (ring is even)
            e.RolloutFusedDevice(acts, 7, n, ring)            # one launch = one flip
            e.StepDevice(acts[1])
            e.Sync()
        a, b = db.Read(), ip.Read()
        assert np.array_equal(a.Observation, b.Observation) and np.array_equal(a.Reward, b.Reward) and np.array_equal(a.Done, b.Done)
        assert np.array_equal(db.GetState(), ip.GetState()) and db.Tick == ip.Tick
        # host-boundary calls keep working on whichever buffer is current
        s = ip.GetState()
        db.SetState(s)
        act = ip.SampleActions(seed=4, tick=0)
        x, y = db.Step(act), ip.Step(act)
        assert np.array_equal(x.Observation, y.Observation)
    with pytest.raises(ValueError):                           # the second external buffer needs the flag and the first buffer
        gpu_pkg.VectorEnv(name, 256, ext_obs_alt=torch.zeros(4096, device="cuda"))


def test_round1_abi_debts(gpu_pkg, oracle):
    import torch
    n = 1 << 16
    # (a) Seed(int[]) in a loop no longer grows device memory (it used to hipMalloc N*8 bytes per call)
    with gpu_pkg.VectorEnv("CartPole-v1", n, seed=1, auto_reset=True) as env:
        seeds = np.arange(n, dtype=np.int64)
        env.Seed(seeds); env.Reset()
        free0 = torch.cuda.mem_get_info()[0]
        for k in range(40):
            env.Seed(seeds + k)
        env.Sync()
        assert free0 - torch.cuda.mem_get_info()[0] < 2 * n * 8
        env.Seed(7)                                                      # back to one key; then per-lane again: same buffer
        env.Seed(seeds); first = env.Reset()
        assert np.array_equal(first[5], oracle.cartpole_reset(5, 5, 0, 1)[:, 0])
    # (b) reset_where(NULL) under AUTORESET: the step already re-drew finished lanes; the idiomatic `if done: Reset()` is a no-op
    with gpu_pkg.VectorEnv("CartPole-v1", 4096, seed=SEED, auto_reset=True) as env:
        env.Reset()
        rng = np.random.default_rng(0)
        for t in range(40):
            out = env.Step(rng.integers(0, 2, 4096).astype(np.int32))
        assert out.Done.any()
        tick = env.Tick
        again = env.ResetWhere(None)
        assert np.array_equal(again, out.Observation) and env.Tick == tick
        env.ResetWhereDevice(None); env.Sync()
        assert env.Tick == tick and np.array_equal(env.Read().Observation, out.Observation)
        m = np.zeros(4096, np.uint8); m[3] = 1
        o2 = env.ResetWhere(m)                                           # an explicit mask still resets
        assert env.Tick == tick + 1 and not np.array_equal(o2[3], out.Observation[3]) and np.array_equal(o2[4], out.Observation[4])
    # (c) the graph cache is bounded: 20 distinct action buffers, results unchanged, memory flat afterwards
    try:
        with gpu_pkg.VectorEnv("CartPole-v1", 2048, seed=SEED, auto_reset=True, launch_policy={"graph": 1}) as a, \
                gpu_pkg.VectorEnv("CartPole-v1", 2048, seed=SEED, auto_reset=True) as b:
            bufs = [torch.randint(0, 2, (4, 2048), dtype=torch.int32, device="cuda") for _ in range(20)]
            torch.cuda.synchronize()
            a.ResetDevice(); b.ResetDevice()
            for rep in range(2):
                for buf in bufs:
                    a.RolloutDevice(buf, 8, 2048, 4)
                    for t in range(8):
                        b.StepDevice(buf[t % 4])
            a.Sync(); b.Sync()
            assert np.array_equal(a.GetState(), b.GetState())
    finally:
        pass
    # (d) masked Discrete.Sample on the device (Discrete.cs:18-26) equals the oracle: per-lane masks and one shared row
    with gpu_pkg.VectorEnv("Acrobot-v1", 5000, seed=SEED, lane_offset=17) as env:
        rng = np.random.default_rng(2)
        mask = rng.integers(0, 2, (5000, 3)).astype(np.uint8)
        mask[:7] = 0                                                     # no valid action -> Start
        mask[7:11] = 2                                                   # only ==1 counts (bmask = mask == 1)
        dm = torch.from_numpy(mask).cuda()
        out = torch.empty(5000, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        env.SampleActionsMaskedDevice(out, dm, per_lane=True, seed=11, tick=3); env.Sync()
        got = out.cpu().numpy()
        assert np.array_equal(got, oracle.discrete_sample_masked(11, 17, 3, 3, 0, mask, 5000))
        assert (got[:11] == 0).all() and all(mask[i, got[i]] == 1 for i in range(11, 5000) if mask[i].any())
        row = torch.tensor([0, 1, 1], dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        env.SampleActionsMaskedDevice(out, row, per_lane=False, seed=11, tick=4); env.Sync()
        got = out.cpu().numpy()
        assert np.array_equal(got, oracle.discrete_sample_masked(11, 17, 4, 3, 0, np.array([0, 1, 1], np.uint8), 5000))
        assert set(np.unique(got)) == {1, 2}
    with gpu_pkg.VectorEnv("Pendulum-v1", 64) as env, pytest.raises(NotImplementedError):
        env.SampleActionsMaskedDevice(out, row, per_lane=False)          # Box.sample cannot be provided a mask (Box.cs:70)
    # (e) sampling with the env's own (seed, tick) no longer replays the reset words (action == 1 iff x0 >= 0)
    with gpu_pkg.VectorEnv("CartPole-v1", 8192, seed=0, auto_reset=True) as env:
        x0 = env.Reset()[:, 0]
        act = env.SampleActions(seed=0, tick=0)
        agree = float(((act == 1) == (x0 >= 0)).mean())
        assert 0.45 < agree < 0.55, agree


def test_library_calls_restore_the_current_device(gpu_pkg):
    import torch
    before = torch.cuda.current_device()
    with gpu_pkg.VectorEnv("CartPole-v1", 1024, device=0, seed=1) as env:
        env.Reset(); env.Step(1)
    assert torch.cuda.current_device() == before


def _run_child(code, timeout=600, env=None):
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if env:
        e.update(env)
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=timeout, env=e)
    assert r.returncode == 0, r.stdout[-3000:] + "\n---\n" + r.stderr[-3000:]
    return r.stdout


SHARDED_CHILD = r"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.getcwd())
import torch, torch.distributed as dist
import __graft_entry__ as ge
pkg = ge.load_package()
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "%d")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)       # RCCL, world of one
n, steps = 1 << 14, 24
stream = torch.cuda.Stream(dev); torch.cuda.set_stream(stream)
res = {}
for overlap in (False, True):
    env = pkg.ShardedVectorEnv("CartPole-v1", n, rank=0, world_size=1, device=0, seed=0x5EED, auto_reset=True,
                               gather_obs=True, tensor_device=dev, force_gather=True, overlap=overlap)
    assert env.gather_obs and env.overlap == overlap
    one = pkg.VectorEnv("CartPole-v1", n, seed=0x5EED, auto_reset=True)
    rng = np.random.default_rng(1)
    env.ResetDevice(); one.Reset()
    acts = torch.empty(n, dtype=torch.int32, device=dev)
    ok = True
    for t in range(steps):
        a = rng.integers(0, 2, n).astype(np.int32)
        acts.copy_(torch.from_numpy(a)); torch.cuda.synchronize()
        env.StepDevice(acts)
        env.AllGatherObs(overlap=overlap)
        want = one.Step(a).Observation
        if overlap and t %% 2 == 0 and t + 1 < steps:
            continue                                   # gather stays in flight across the next step
        env.WaitGather(); env.Sync(); torch.cuda.synchronize()
        got = env.GlobalObs().cpu().numpy()            # [G=1, D, n]
        ok = ok and np.array_equal(got[0].T, want)
    res["overlap" if overlap else "serial"] = bool(ok)
    env.Close(); one.Close()
dist.destroy_process_group()
print("RESULT " + json.dumps(res))
"""


def test_sharded_vector_env_on_hip_with_rccl_world_of_one(gpu_pkg):
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = _run_child(SHARDED_CHILD % port)
    line = [l for l in out.splitlines() if l.startswith("RESULT ")][-1]
    assert json.loads(line[7:]) == {"serial": True, "overlap": True}


def test_plain_bench_gpus_2_spawns_its_own_ranks(gpu_pkg):
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE: the parent spawns two ranks before touching the GPU;
    on this 1-GPU box they share the device over gloo (labelled as such).  This is the driver's SCALE command shape."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--ring", "8",
                        "--num-envs", str(1 << 18)], cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 20 and j["warmup"] == 5 and j["repeats"] >= 3 and j["scaling"] == "weak"
    assert j["config"]["global_num_envs"] == 2 << 18 and j["value"] > 1e8
    import torch
    if torch.cuda.device_count() < 2:
        assert "gloo" in j["config"]["backend"] and "SHARE" in j["config"]["backend"]
    assert "shared-memory" in j["config"]["barrier"]
    g = j["with_obs_allgather"]                       # two ranks, HIP IPC peer buffers, hand-written push: the direct gather ran
    for k in ("direct_ipc", "direct_ipc_overlapped"):
        assert g[k].get("gathered_obs_finite_and_nonzero") is True and g[k]["value"] > 1e8, g
    _check_multi_rank_evidence(j, 2)
    assert j["cpu_baseline"]["kind"] == "port" and j["cpu_baseline"]["value"] > 1e6 and "after the other ranks had exited" in j["cpu_baseline"]["when"]


def _check_multi_rank_evidence(j, world):
    """What the first real multi-GPU lease must carry in ONE line: who ran where, proof that the collective
    backend spans `world` ranks, each rank's own events-based step time, and the single-process gymnet_group_* leg."""
    import torch
    ranks = j["ranks"]
    assert [e["rank"] for e in ranks] == list(range(world)) and len({e["pid"] for e in ranks}) == world
    for e in ranks:
        assert e["kernel"].startswith("step_kernel<CartPole") and 0.5 < e["events_us_per_step"] < 1e4 and e["name"]
    c = j["collective"]
    assert c["world_size"] == world and c["allreduce_sum_of_rank_plus_1"] == c["expected"] == world * (world + 1) // 2
    assert c["is_rccl"] == (torch.cuda.device_count() >= world) and c["distinct_devices"] == min(world, torch.cuda.device_count())
    g = j["group_single_process"]
    assert g["members"] == world and g["real_multi_gpu"] == (torch.cuda.device_count() >= world)
    assert g["step_only"]["value"] > 1e8
    for k in ("direct", "direct_overlapped"):
        assert g[k]["every_member_slice_arrived"] is True and g[k]["value"] > 1e8, g
    assert ("skipped" in g["rccl"]) == (torch.cuda.device_count() < world)


@pytest.mark.gpu
def test_bench_under_the_launcher_the_driver_uses(gpu_pkg):
    """`python -m torch.distributed.run --nproc-per-node 2 … bench.py --gpus 2 …` — the contract's launch line for N > 1.
    With fewer GPUs than ranks the ranks share the device over gloo (chosen automatically, labelled); with one GPU per
    rank the same command runs over RCCL."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "GYMNET_BENCH_BACKEND")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29617", os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "20", "--warmup", "5", "--ring", "8", "--num-envs", str(1 << 18),
                        "--no-cpu-baseline"], cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 20 and j["warmup"] == 5 and j["scaling"] == "weak" and j["value"] > 1e8
    import torch
    assert ("gloo" in j["config"]["backend"]) == (torch.cuda.device_count() < 2)
    _check_multi_rank_evidence(j, 2)                 # the launcher form carries the same evidence (rank 0 starts the group leg)


DIRECT_CHILD = r"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.getcwd())
import torch, torch.distributed as dist
import __graft_entry__ as ge
pkg = ge.load_package()
rank, world, out_dir = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), sys.argv[1]
dist.init_process_group("gloo", rank=rank, world_size=world)        # control plane only: handle exchange + barrier
dev = torch.device("cuda", 0); torch.cuda.set_device(0)              # both ranks share the one GPU of the box
stream = torch.cuda.Stream(dev); torch.cuda.set_stream(stream)
n, steps = 1 << 14, 16
for overlap in (False, True):
    env = pkg.ShardedVectorEnv("CartPole-v1", n, rank=rank, world_size=world, device=0, seed=0x5EED, auto_reset=True,
                               gather_obs=True, tensor_device=dev, overlap=overlap, gather="direct")
    assert env.gather == "direct" and env.overlap == overlap
    rng = np.random.default_rng(1)
    lo, hi = env.lane_offset, env.lane_offset + env.local_num_envs
    env.ResetDevice()
    acts = torch.empty(hi - lo, dtype=torch.int32, device=dev)
    snaps = []
    for t in range(steps):
        a = rng.integers(0, 2, n).astype(np.int32)
        acts.copy_(torch.from_numpy(a[lo:hi])); torch.cuda.synchronize()
        env.StepDevice(acts)
        env.AllGatherObs(overlap=overlap)
        env.WaitGather(); env.Sync(); torch.cuda.synchronize()
        snaps.append(env.GlobalObs().cpu().numpy().copy())           # [G, D, n/G]
    np.save(os.path.join(out_dir, f"direct_{int(overlap)}_rank{rank}.npy"), np.stack(snaps))
    env.Close()
dist.barrier()
dist.destroy_process_group()
"""


def test_direct_allgather_between_two_processes_over_hip_ipc(gpu_pkg, tmp_path):
    """One process per GPU with the hand-written gather: two ranks (sharing this box's one GPU) export their replica
    buffers as HIP IPC peer buffers, map each other's, push their slices with gymnet_push_obs_device, and synchronise with
    stream-sync + barrier.  Every rank's gathered [G][D][N/G] buffer must equal the single-handle batch bit for bit."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "child.py"
    script.write_text(DIRECT_CHILD)
    procs = []
    for r in range(2):
        e = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script), str(tmp_path)], cwd=ROOT, env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n---\n".join(o[-2500:] for o in outs)
    n, steps = 1 << 14, 16
    with gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED, auto_reset=True) as one:
        one.Reset()
        rng = np.random.default_rng(1)
        want = [one.Step(rng.integers(0, 2, n).astype(np.int32)).Observation for _ in range(steps)]
    for overlap in (0, 1):
        got = [np.load(tmp_path / f"direct_{overlap}_rank{r}.npy") for r in range(2)]
        assert np.array_equal(got[0], got[1])
        for t in range(steps):
            assert np.array_equal(np.concatenate(list(got[0][t]), axis=1).T, want[t]), (overlap, t)


def test_group_create_destroy_does_not_leak(gpu_pkg):
    import torch
    for _ in range(3):                                        # settle allocator / RCCL one-time allocations
        gpu_pkg.GroupVectorEnv("CartPole-v1", 1 << 16, 4, devices=[0] * 4, seed=1, auto_reset=True, overlap=True).Close()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(20):
        with gpu_pkg.GroupVectorEnv("CartPole-v1", 1 << 16, 4, devices=[0] * 4, seed=1, auto_reset=True, overlap=True) as g:
            g.ResetDevice(); g.AllGatherObs(); g.WaitGather(); g.Sync()
    torch.cuda.synchronize()
    assert free0 - torch.cuda.mem_get_info()[0] < (4 << 20)


@pytest.mark.parametrize("overlap", [False, True])
def test_group_rollout_device_equals_single_handle(gpu_pkg, overlap):
    """gymnet_group_rollout_device: with G > 1 every member replays a captured graph of `ring` steps (one host thread cannot
    feed G GPUs with eager launches); the result must equal the single-handle rollout bit for bit, also when the observation
    buffers ping-pong (even-length graphs keep the buffer parity)."""
    import torch
    G, n, ring, steps = 4, 4 * 2048, 6, 45                  # 45 = 7 graph replays of 6 + 3 eager steps
    with gpu_pkg.GroupVectorEnv("CartPole-v1", n, G, devices=[0] * G, seed=SEED, auto_reset=True, gather="direct", overlap=overlap) as grp, \
            gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED, auto_reset=True) as one:
        nl = grp.LanesPerMember
        acts = torch.empty((ring, n), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        for t in range(ring):
            one.SampleActionsDevice(acts[t], seed=8, tick=t)
        one.Sync()
        macts = [acts[:, m * nl:(m + 1) * nl].contiguous() for m in range(G)]      # [ring][nl] per member, stride nl
        torch.cuda.synchronize()
        one.ResetDevice(); grp.ResetDevice()
        one.RolloutDevice(acts, steps, n, ring)
        grp.RolloutDevice(macts, steps, nl, ring)
        grp.AllGatherObs(); grp.WaitGather(); grp.Sync(); one.Sync()
        want = one.Read()
        rep = grp.ReadReplica(G - 1)
        assert np.array_equal(np.concatenate(list(rep), axis=1).T, want.Observation)
        for m in range(G):
            r = grp.Members[m].Read()
            assert np.array_equal(r.Reward, want.Reward[m * nl:(m + 1) * nl]) and np.array_equal(r.Done, want.Done[m * nl:(m + 1) * nl])
            assert grp.Members[m].Tick == one.Tick


def test_baseline_config5_shape_on_one_device(gpu_pkg):
    """BASELINE config 5 — CartPole, 2^23 lanes sharded 8 ways with an observation all-gather — in the only form a 1-GPU box
    allows: eight LOGICAL members of 2^20 lanes each on device 0, double-buffered, hand-written direct gather overlapped with
    the next step.  After every step each member's replica [8][4][2^20] must equal the single 2^23-lane batch bit for bit
    (global-lane Philox keys, rank-major layout, zero-copy send side, buffer ping-pong)."""
    import torch
    G, n, steps = 8, 1 << 23, 4
    nl = n // G
    with gpu_pkg.GroupVectorEnv("CartPole-v1", n, G, devices=[0] * G, seed=SEED, auto_reset=True, gather="direct", overlap=True) as grp, \
            gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED, auto_reset=True) as one:
        acts = torch.empty((steps, n), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        for t in range(steps):
            one.SampleActionsDevice(acts[t], seed=3, tick=t)
        one.Sync()
        one.ResetDevice(); grp.ResetDevice()
        for t in range(steps):
            one.StepDevice(acts[t])
            grp.StepDevice([acts[t, m * nl:(m + 1) * nl] for m in range(G)])
            grp.AllGatherObs()
            if t % 2 == 0 and t + 1 < steps:
                continue                                   # this gather stays in flight across the next step
            grp.WaitGather(); grp.Sync(); one.Sync()
            want = one.GetState()                          # [4, n] == the observation for CartPole
            for m in (0, G - 1):
                rep = grp.ReadReplica(m)                   # [G, 4, nl]
                assert np.array_equal(np.concatenate(list(rep), axis=1), want), (t, m)
        assert one.Counters()["lane_steps"] == steps * n


def test_group_flags_validation_and_bookkeeping(gpu_pkg, oracle):
    import torch
    G, n = 4, 4 * 1024
    nl = n // G
    rng = np.random.default_rng(17)
    # VALIDATE_ACTIONS: one bad action anywhere in the batch rejects the step before ANY member has advanced
    with gpu_pkg.GroupVectorEnv("CartPole-v1", n, G, devices=[0] * G, seed=SEED, auto_reset=True, gather="none", validate_actions=True) as grp:
        grp.Reset()
        ticks = [m.Tick for m in grp.Members]
        a = rng.integers(0, 2, n).astype(np.int32)
        a[2 * nl + 5] = 7                                               # member 2
        with pytest.raises(gpu_pkg.InvalidActionError):
            grp.Step(a)
        assert [m.Tick for m in grp.Members] == ticks
        da = torch.from_numpy(a).cuda(); torch.cuda.synchronize()
        with pytest.raises(gpu_pkg.InvalidActionError):
            grp.StepDevice([da[m * nl:(m + 1) * nl] for m in range(G)])
        assert [m.Tick for m in grp.Members] == ticks
        a[2 * nl + 5] = 1
        grp.Step(a)
        assert [m.Tick for m in grp.Members] == [t + 1 for t in ticks]
    # EPISODE_STATS on every member == the slices of the single handle's statistics
    with gpu_pkg.GroupVectorEnv("CartPole-v1", n, G, devices=[0] * G, seed=SEED, auto_reset=True, gather="none", episode_stats=True) as grp, \
            gpu_pkg.VectorEnv("CartPole-v1", n, seed=SEED, auto_reset=True, episode_stats=True) as one:
        grp.Reset(); one.Reset()
        for t in range(60):
            a = rng.integers(0, 2, n).astype(np.int32)
            grp.Step(a); one.Step(a)
        ret, ln = one.EpisodeStats()
        assert ln.max() > 0
        for m in range(G):
            r, l = grp.Members[m].EpisodeStats()
            assert np.array_equal(r, ret[m * nl:(m + 1) * nl]) and np.array_equal(l, ln[m * nl:(m + 1) * nl])
    # the stand-alone masked sampler (no handle): Discrete.Sample(mask) for an arbitrary lane range
    lib, capi = gpu_pkg.load_library(), gpu_pkg._capi
    cnt = 3000
    mask = rng.integers(0, 2, (cnt, 5)).astype(np.uint8)
    dm = torch.from_numpy(mask).cuda()
    out = torch.empty(cnt, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    capi.check(lib.gymnet_sample_discrete_masked_device(0, None, C.c_void_p(out.data_ptr()), cnt, 5, 10, C.c_void_p(dm.data_ptr()), 5, 99, 1234, 6))
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), oracle.discrete_sample_masked(99, 1234, 6, 5, 10, mask, cnt))
    capi.check(lib.gymnet_sample_discrete_masked_device(0, None, C.c_void_p(out.data_ptr()), cnt, 5, 10, None, 0, 99, 1234, 6))   # no mask: Discrete.cs:27
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), oracle.discrete_sample(99, 1234, 6, 5, 10, cnt))


@pytest.mark.gpu
def test_driver_shaped_bench_line_carries_the_contract(gpu_pkg):
    """`python bench.py --gpus 1 --steps 20 --warmup 5` — the driver's N = 1 command — prints ONE JSON line with the contract's
    keys, the `roofline` and `cpu_baseline` objects, and the secondary figures (none of which may ever be `value`)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--cpu-seconds", "1"],
                       cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["metric"] == "env-steps/sec" and j["unit"] == "env-steps/s" and j["n_gpus"] == 1 and j["steps"] == 20 and j["warmup"] == 5
    assert j["higher_is_better"] is True and j["scaling"] == "weak" and j["vs_baseline"] is None and j["dtype"] == "f32" and j["data"] == "synthetic"
    assert j["value"] > 5e10 and abs(j["value"] - (1 << 20) / (j["ms_per_step"] * 1e-3)) < 1e-3 * j["value"]
    assert "CartPole-v1 batched, batch=1048576" in j["config"]["workload"] and "model" not in j["config"]
    rf = j["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert rf["kernel"] == "step_kernel<CartPole,4,true,false,15,1>" and rf["algorithmic_bytes_per_launch"] == 41 << 20
    # `achieved` / `frac`: the wall-clock figures (bytes per launch / ms_per_step — what the driver can recompute from the line);
    # the HIP-event (kernel-side) figures beside them, never below them
    assert abs(rf["achieved"] - 41 * (1 << 20) / (j["ms_per_step"] * 1e-3) / 1e9) < 1e-6 * rf["achieved"] and 0.4 < rf["frac"] < 1.0
    assert abs(rf["achieved_by_events"] - 41 * (1 << 20) / (rf["launch_us"] * 1e-6) / 1e9) < 1e-6 * rf["achieved_by_events"]
    assert rf["frac"] == rf["frac_by_wall"] <= rf["frac_by_events"] * 1.02 < 1.0 and rf["bytes_per_env_step"] == 41
    # roofline.traffic is MEASURED in the run (two rocprofv3 --pmc child passes, FETCH_SIZE / WRITE_SIZE apart), per launch like
    # `achieved`; it must sit within a few percent of the 41 B x 2^20 the kernel moves — more would mean wasted re-reads
    # (the strict form of this check — the measurement must WORK — is tests/test_zz_gpu_traffic_measurement.py, last in the run, so
    # that a box where a nested profiler cannot start does not cut the remaining parity tests short under `pytest -x`)
    assert 0.97 < rf["traffic"] / (41 << 20) < 1.06
    if rf["traffic_source"].startswith("measured in this run"):
        assert abs(rf["traffic_over_moved_bytes"] - rf["traffic"] / (41 << 20)) < 1e-9
        assert 0.9 < rf["traffic"] / rf["traffic_constant_from_profiles"] < 1.1          # and agrees with the committed profile
    else:
        assert "traffic_measurement_error" in rf and "NOT measured in this run" in rf["traffic_source"]   # labelled fallback, never silent
    cb = j["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 1e6 and cb["unit"] == "env-steps/s" and cb["sample"]
    assert j["fused_rollout"]["us_per_step"] < j["ms_per_step"] * 1e3
    hb = j["host_boundary"]
    assert hb["pinned_library_buffers"]["ms_per_step"] <= hb["pageable_caller_buffers"]["ms_per_step"] * 1.1
    assert hb["pinned_library_buffers"]["env_steps_per_sec"] < 0.1 * j["value"]           # PCIe-inclusive: never the headline
    oc = j["other_configs_2p20"]
    assert oc["Pendulum-v1"]["kernel"].startswith("step_kernel<Pendulum,4,true") and oc["Acrobot-v1"]["kernel"] == "step_kernel_pipe<Acrobot,4,true,15>"
    assert all(0.3 < oc[k]["frac_of_peak"] <= oc[k]["frac_of_peak_algorithmic_bytes"] < 1.0 for k in oc)       # priced on MOVED bytes
    assert oc["Acrobot-v1"]["moved_bytes_per_step"] == 57 and oc["Pendulum-v1"]["moved_bytes_per_step"] == 33
    f64 = j["cartpole_f64_2p20"]
    assert f64["kernel"] == "step_kernel_pipe2<CartPole64,4,true,15>" and f64["bytes_per_env_step"] == 73 and 0.3 < f64["frac_of_peak"] < 1.0
    assert f64["env_steps_per_sec"] < j["value"]                                          # beside, never as, `value`
    assert j["hbm_resident_2p27"]["num_envs"] == 1 << 27 and 0.5 < j["hbm_resident_2p27"]["frac_of_peak"] < 1.0
