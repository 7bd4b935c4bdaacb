"""world_size-2 and world_size-8 tests of the multi-GPU path on CPU (gloo): one process per rank, each owning a contiguous
lane block whose observations live inside the rank-major all-gather buffer; after every step the gathered
[G][D][N/G] buffer on EVERY rank must equal the single-shard batch bit for bit (sharding invariance,
SURVEY.md §8(e)).  Compute is the oracle-backed stand-in from tests/_oracle_local_env.py — the HIP engine
needs a GPU; what is under test here is gym.net_amd/sharding.py."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, STEPS, SEED = 4096, 25, 0x5EED


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir, overlap=False, dtype="float32"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    from _oracle_local_env import OracleLocalEnv
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = ge.load_package()
    env = pkg.ShardedVectorEnv("CartPole-v1", N, rank=rank, world_size=world, seed=SEED, auto_reset=True,
                               gather_obs=True, local_env_factory=OracleLocalEnv, tensor_device="cpu", overlap=overlap, dtype=dtype)
    assert env.overlap == overlap and env.obs_bufs.shape[0] == (2 if overlap else 1)
    assert env.obs_bufs.dtype == (torch.float64 if dtype == "float64" else torch.float32)
    assert env.local_num_envs == N // world and env.lane_offset == rank * N // world
    rng = np.random.default_rng(99)                       # every rank derives the same GLOBAL action array
    acts = rng.integers(0, 2, (STEPS, N)).astype(np.int32)
    lo, hi = env.lane_offset, env.lane_offset + env.local_num_envs
    env.ResetDevice()
    snaps = []
    for t in range(STEPS):
        env.StepDevice(torch.from_numpy(acts[t, lo:hi].copy()))
        if overlap:                                       # double-buffered: the gather of step t is only waited for
            env.AllGatherObs(overlap=True)                # when its buffer is read here / about to be overwritten
            if t % 3 != 0:
                env.WaitGather()
                snaps.append(env.GlobalObs().clone().numpy())
            else:                                         # leave it in flight across the next step, then read it
                last = env.GlobalObs()
                snaps.append(None)
                if t + 1 == STEPS:
                    env.WaitGather()
                    snaps[-1] = last.clone().numpy()
            if t > 0 and snaps[t - 1] is None:
                env._finish(env._last ^ 1)
                snaps[t - 1] = env.obs_bufs[env._last ^ 1].clone().numpy()
        else:
            env.AllGatherObs(async_op=(t % 2 == 0))       # both the blocking and the async form
            env.Wait()
            snaps.append(env.GlobalObs().clone().numpy())
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), np.stack(snaps))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world,overlap,dtype", [(2, False, "float32"), (2, True, "float32"), (8, False, "float32"), (8, True, "float32"),
                                                 (2, True, "float64"), (8, False, "float64"), (8, True, "float64")])
def test_sharded_rollout_equals_single_shard(tmp_path, oracle, world, overlap, dtype):
    """World sizes 2 and 8 (BASELINE config 5's shape: 8 ranks, rank-major [8][4][N/8] gather buffer), with and without the
    double-buffered overlap, in float32 and in the reference-arithmetic float64 mode (gather buffers of doubles): ShardPlan's lane
    blocks, the per-rank Philox lane offsets, the overlap bookkeeping and the all-gather — every rank must end up holding the
    single-shard batch, bit for bit, after every step."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), overlap, dtype), nprocs=world, join=True)
    got = [np.load(tmp_path / f"rank{r}.npy") for r in range(world)]
    assert got[0].shape == (STEPS, world, 4, N // world) and got[0].dtype == np.dtype(dtype)
    for r in range(1, world):
        assert np.array_equal(got[0], got[r]), r          # every rank holds the same gathered observations
    # single shard, same seed, same global actions, same kernel-semantics oracle
    rng = np.random.default_rng(99)
    acts = rng.integers(0, 2, (STEPS, N)).astype(np.int32)
    f64 = dtype == "float64"
    s = oracle.cartpole_reset_f64(SEED, 0, 0, N) if f64 else oracle.cartpole_reset(SEED, 0, 0, N)
    for t in range(STEPS):
        if f64:
            s, _, _ = oracle.cartpole_autoreset_step_f64(SEED, 0, t + 1, s, acts[t])
            gathered = got[0][t]
            assert np.array_equal(np.concatenate(list(gathered), axis=1), s), t
            continue
        s2, _, d, _ = oracle.cartpole_step(s, acts[t], dtype=np.float32)
        fresh = oracle.cartpole_reset(SEED, 0, t + 1, N)
        fin = d.astype(bool)
        s2[:, fin] = fresh[:, fin]
        s = s2
        gathered = got[0][t]                               # [G, 4, N/G] rank-major
        assert np.array_equal(np.concatenate(list(gathered), axis=1), s), t


def _failing_setup_worker(rank, world, port, out_dir, fail_stage):
    """Peer-buffer set-up with a library whose export (stage "create") or import (stage "open") fails on rank 1 only."""
    sys.path.insert(0, ROOT)
    import ctypes as C
    import torch.distributed as dist
    import __graft_entry__ as ge
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = ge.load_package()
    import importlib
    capi = importlib.import_module(pkg.__name__ + "._capi")
    sharding = importlib.import_module(pkg.__name__ + ".sharding")
    calls = []

    class FakeLib:
        def gymnet_peer_buffer_create(self, device, nbytes, base, handle):
            calls.append("create")
            if fail_stage == "create" and rank == 1:
                return capi.ERR_HIP
            C.cast(base, C.POINTER(C.c_void_p))[0] = C.c_void_p(0x1000 * (rank + 1))
            return 0

        def gymnet_peer_buffer_open(self, device, handle, out):
            calls.append("open")
            if fail_stage == "open" and rank == 1:
                return capi.ERR_HIP
            C.cast(out, C.POINTER(C.c_void_p))[0] = C.c_void_p(0x9000)
            return 0

        def gymnet_peer_buffer_close(self, device, p):
            calls.append("close")
            return 0

        def gymnet_peer_buffer_destroy(self, device, p):
            calls.append("destroy")
            return 0

        def gymnet_last_error(self):
            return b"injected failure"

    fake = FakeLib()
    capi.load_library = lambda: fake
    env = sharding.ShardedVectorEnv.__new__(sharding.ShardedVectorEnv)
    import torch
    env._torch, env._dist, env.group = torch, dist, None
    env.rank, env.world_size, env.tensor_device = rank, world, torch.device("cpu")
    env._esz = 4
    msg = ""
    try:
        env._make_peer_buffers((1, world, 4, 16), 0)
    except RuntimeError as e:
        msg = str(e)
    with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as f:
        f.write(msg + "\n" + " ".join(calls))
    dist.barrier()                                        # both ranks are still in step: nobody is stuck in a collective
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("fail_stage", ["create", "open"])
def test_peer_buffer_setup_failure_on_one_rank_is_raised_on_every_rank(tmp_path, fail_stage):
    """The fault the 8-rank exercise found: one rank's HIP-IPC export failed and the others waited in a collective for
    ever.  Now every rank reports its set-up result; a failure anywhere raises everywhere and what was created / opened
    is released in order (importers close, then exporters free)."""
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    mp.spawn(_failing_setup_worker, args=(world, port, str(tmp_path), fail_stage), nprocs=world, join=True)
    for r in range(world):
        msg, calls = open(tmp_path / f"rank{r}.txt").read().split("\n")
        assert "rank(s) [1]" in msg and ("creation failed" if fail_stage == "create" else "opening peer buffers failed") in msg, msg
        calls = calls.split()
        if fail_stage == "create":
            assert calls == (["create", "destroy"] if r == 0 else ["create"]), calls
        else:
            assert calls == (["create", "open", "close", "destroy"] if r == 0 else ["create", "open", "destroy"]), calls
