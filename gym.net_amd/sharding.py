"""Lane-block sharding of one logical batch over the GPUs of a node — one process per GPU.

The reference has no distributed code at all (its `Distributed*` classes are an in-process thread
pool, src/Gym/Internal/Threading/*); what shards here is the independence the reference's own
vector wrapper already has: VecEnvWrapper.Step maps envs independently
(src/Gym/Envs/VecEnvWrapper.cs:22-24), and nothing in CartPoleEnv.Step couples two instances
(src/Gym.Environments/Envs/Classic/CartPoleEnv.cs:137-186).

  - rank r of G owns the contiguous global lanes [r*N/G, (r+1)*N/G);
  - reset draws are keyed by the GLOBAL lane id (gymnet_config.lane_offset), so the results do not
    depend on G (sharding invariance: concat of the shards == the one-GPU batch, bitwise);
  - the data path needs NO collective.  The one exchange north_star names — an all-gather of the
    observations so every rank sees [N, D] (e.g. for a replicated policy) — is optional, runs through
    torch.distributed (backend "nccl" = RCCL over xGMI), and is zero-copy on the send side: the
    rank's observation arrays live INSIDE the gather buffer (gymnet_config.d_ext_obs), laid out
    rank-major [G][D][N/G] so one all-gather moves everything with no repack.

torch is used here for device memory and the collective only (plumbing, not compute).
"""
from .vector_env import VectorEnv


class ShardPlan:
    """Contiguous lane blocks.  N must divide evenly when observations are gathered (equal-size blocks)."""

    def __init__(self, global_num_envs, world_size):
        if world_size < 1 or global_num_envs < world_size:
            raise ValueError("need world_size >= 1 and at least one env per rank")
        self.N, self.G = int(global_num_envs), int(world_size)

    def offset(self, rank):
        return self.N * rank // self.G

    def count(self, rank):
        return self.N * (rank + 1) // self.G - self.N * rank // self.G

    def shard(self, rank):
        if not 0 <= rank < self.G:
            raise ValueError("rank out of range")
        return self.offset(rank), self.count(rank)

    @property
    def even(self):
        return self.N % self.G == 0

    def owner(self, global_lane):
        """(rank, local lane) of a global lane id."""
        if not 0 <= global_lane < self.N:
            raise ValueError("lane out of range")
        r = min(self.G - 1, (global_lane * self.G + self.G - 1) // self.N)
        while self.offset(r) > global_lane:
            r -= 1
        while self.offset(r) + self.count(r) <= global_lane:
            r += 1
        return r, global_lane - self.offset(r)


def _hip_local_env(env, num_envs, lane_offset, seed, auto_reset, ext_obs, ext_obs_stride, device, stream, ext_obs_alt=None, dtype="float32"):
    return VectorEnv(env, num_envs, device=device, seed=seed, auto_reset=auto_reset, lane_offset=lane_offset,
                     ext_obs=ext_obs, ext_obs_stride=ext_obs_stride, stream=stream,
                     double_buffer=ext_obs_alt is not None, ext_obs_alt=ext_obs_alt, dtype=dtype)


class ShardedVectorEnv:
    """This rank's shard of a global batch + the optional observation all-gather.

    gather="direct" pushes this rank's slice into every peer's replica (HIP IPC peer buffers).  Its cross-process ordering is
    stream synchronize -> `barrier` -> push -> stream synchronize -> `barrier`: `barrier` is any HOST barrier over the ranks
    (default: the process group's) and is only ever called right after the main stream has been drained, so "every rank
    passed the barrier" means every rank's queued readers of the old replicas have finished.

    overlap=True double-buffers the observation arrays (GYMNET_FLAG_DOUBLE_BUFFER): there are two gather buffers, step
    t+1 writes the other one while the all-gather of step t's buffer is still in flight on a side stream, so a
    consumer that wants every rank to see all observations pays max(step, gather) per step instead of step + gather.

    dtype="float64" (CartPole) shards the reference-arithmetic mode (GYMNET_FLAG_F64, CartPoleEnv.cs:141-166,185): the gather
    buffers hold doubles — 32 MiB per rank at N = 2^23, G = 8 instead of 16 — and everything else is unchanged.

    local_env_factory exists so the host-side sharding logic can be exercised without a GPU (the
    CPU tests inject an oracle-backed stand-in); the default — and the only thing the product ever
    uses — is the HIP VectorEnv.
    """

    def __init__(self, env, global_num_envs, rank=None, world_size=None, device=None, seed=0, auto_reset=True,
                 gather_obs=True, process_group=None, local_env_factory=None, tensor_device=None, force_gather=False,
                 overlap=False, gather="rccl", barrier=None, dtype="float32"):
        import torch
        import torch.distributed as dist
        self._torch, self._dist = torch, dist
        import numpy as np
        self.dtype = np.dtype(dtype)
        if self.dtype not in (np.dtype(np.float32), np.dtype(np.float64)):
            raise ValueError("dtype must be float32 or float64")
        self._tdtype = torch.float64 if self.dtype == np.float64 else torch.float32
        self._esz = self.dtype.itemsize
        self.group = process_group
        if world_size is None:
            world_size = dist.get_world_size(process_group) if dist.is_initialized() else 1
        if rank is None:
            rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self.rank, self.world_size = int(rank), int(world_size)
        self.plan = ShardPlan(global_num_envs, world_size)
        self.gather_obs = bool(gather_obs) and (world_size > 1 or force_gather)
        self.overlap = bool(overlap) and self.gather_obs
        if gather_obs and not self.plan.even:
            raise ValueError("global_num_envs must be a multiple of world_size to all-gather observations")
        self.lane_offset, self.local_num_envs = self.plan.shard(self.rank)
        factory = local_env_factory or _hip_local_env
        if tensor_device is None:
            tensor_device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self.tensor_device = torch.device(tensor_device)
        if device is None:
            device = self.tensor_device.index if self.tensor_device.type == "cuda" else 0
        from . import _capi
        env_id = _capi.ENV_IDS[env] if isinstance(env, str) else int(env)
        self.obs_dim = {0: 4, 1: 3, 2: 2, 3: 6}[env_id]
        n_local = self.local_num_envs
        # rank-major gather buffers [B][G][D][N/G] (B = 2 when overlapping); this rank's observation arrays ARE slice
        # [b][rank] of whichever buffer b the last step wrote
        nbuf = 2 if self.overlap else 1
        self._cuda = self.tensor_device.type == "cuda"
        # gather = "rccl": the collective of the process group (RCCL over xGMI for backend "nccl").
        # gather = "direct": the hand-written push — every rank's buffers are peer buffers (HIP IPC), each rank maps all the
        # others' and stores its slice straight into them, one xGMI link per peer (gymnet_push_obs_device); cross-process
        # ordering is a stream synchronize + `barrier` (a callable; default: the process group's barrier).
        self.gather = gather if (self.gather_obs and self._cuda) else "rccl"
        self._barrier = barrier or (lambda: dist.barrier(group=process_group))
        self._peers, self._own_base = None, None
        shape = (nbuf, self.world_size, self.obs_dim, n_local)
        if self.gather == "direct":
            self.obs_bufs = self._make_peer_buffers(shape, device)
        else:
            self.obs_bufs = torch.zeros(shape, dtype=self._tdtype, device=self.tensor_device)
        self.obs_all = self.obs_bufs[0]
        stream = None
        if self._cuda:
            stream = torch.cuda.current_stream(self.tensor_device).cuda_stream
        kw = {"ext_obs_alt": self.obs_bufs[1][self.rank].data_ptr()} if self.overlap else {}
        if self.dtype == np.float64:
            kw["dtype"] = "float64"            # (only passed when asked for: stand-in factories of older tests take no dtype)
        self.local = factory(env, n_local, self.lane_offset, seed, auto_reset, self.obs_bufs[0][self.rank].data_ptr(),
                             n_local, device, stream, **kw)
        self._work = None
        self._cur = 0                      # buffer the latest observation lives in (mirror of the handle's obs_buffer)
        self._last = 0                     # buffer gathered last
        self._pending = [None, None]       # per buffer: an unfinished overlapped gather (cuda event or Work)
        self._gstream = torch.cuda.Stream(self.tensor_device) if (self._cuda and self.overlap) else None
        # events are created once and re-recorded: a fresh torch.cuda.Event per step costs more host time than the step kernel
        self._ev_step = torch.cuda.Event() if self._gstream is not None else None
        self._ev_done = [torch.cuda.Event(), torch.cuda.Event()] if self._gstream is not None else None

    # ---- direct gather: peer buffers over HIP IPC --------------------------------------------------------------
    def _make_peer_buffers(self, shape, device):
        import ctypes as C
        import numpy as np
        from . import _capi
        torch, dist = self._torch, self._dist
        lib = _capi.load_library()
        nbytes = int(np.prod(shape)) * self._esz
        base, handle = C.c_void_p(), _capi.IpcHandle()
        self._peer_dev, self._lib, self._peers = int(device), lib, {}

        def agree(err, payload=None):
            """Every rank reports (error or None, payload); a failure anywhere is raised EVERYWHERE, so no rank goes on to a
            collective or a barrier its peers will never reach."""
            got = [None] * self.world_size
            dist.all_gather_object(got, (err, payload), group=self.group)
            bad = {r: e for r, (e, _) in enumerate(got) if e is not None}
            return bad, [pl for _, pl in got]

        err = None
        try:
            _capi.check(lib.gymnet_peer_buffer_create(int(device), nbytes, C.byref(base), C.byref(handle)))
        except Exception as e:                                      # noqa: BLE001 - reported to every rank below
            err = repr(e)
        # all 64 bytes of the struct (a c_char array FIELD reads back as a C string, cut at the first NUL)
        bad, handles = agree(err, None if err else C.string_at(C.byref(handle), 64))
        if bad:
            if err is None:
                lib.gymnet_peer_buffer_destroy(int(device), base)
            raise RuntimeError(f"peer buffer creation failed on rank(s) {sorted(bad)}: {next(iter(bad.values()))}")
        self._own_base = base.value
        err = None
        try:
            for r, hb in enumerate(handles):
                if r == self.rank:
                    continue
                h, p = _capi.IpcHandle(), C.c_void_p()
                assert len(hb) == 64
                C.memmove(C.byref(h), hb, 64)
                _capi.check(lib.gymnet_peer_buffer_open(int(device), C.byref(h), C.byref(p)))
                self._peers[r] = p.value
        except Exception as e:                                      # noqa: BLE001
            err = repr(e)
        bad, _ = agree(err)
        if bad:
            for p in self._peers.values():
                lib.gymnet_peer_buffer_close(int(device), p)
            self._peers = None
            agree(None)                                             # every importer has unmapped before any exporter frees
            lib.gymnet_peer_buffer_destroy(int(device), self._own_base)
            self._own_base = None
            raise RuntimeError(f"opening peer buffers failed on rank(s) {sorted(bad)}: {next(iter(bad.values()))}")

        class _Raw:                                   # zero-copy torch view of the library's allocation
            __cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<f8" if self._esz == 8 else "<f4", "data": (self._own_base, False), "version": 2}
        t = torch.as_tensor(_Raw(), device=self.tensor_device)
        return t

    def _push(self, b, stream_ptr):
        import ctypes as C
        from . import _capi
        G, D, n = self.world_size, self.obs_dim, self.local_num_envs
        off = ((b * G + self.rank) * D * n) * self._esz              # byte offset of slice [b][rank] in every replica
        peers = [self._peers[r] for r in sorted(self._peers)]
        dst = (C.c_void_p * max(1, len(peers)))(*[C.c_void_p(p + off) for p in peers])
        _capi.check(self._lib.gymnet_push_obs_device(self._peer_dev, C.c_void_p(stream_ptr), C.c_void_p(self._own_base + off), dst,
                                                     len(peers), D * n * (self._esz // 4)))      # count in 4-byte words

    # ---- stepping -----------------------------------------------------------------------------------
    def _current_buffer(self):
        if not self.overlap:
            return 0
        if hasattr(self.local, "ObsBufferIndex"):
            self._cur = self.local.ObsBufferIndex()
        return self._cur

    def _finish(self, b):
        """Orders later work on the main stream behind the overlapped gather of buffer b."""
        p = self._pending[b]
        if p is None:
            return
        if self._cuda:
            self._torch.cuda.current_stream(self.tensor_device).wait_event(p)
        else:
            p.wait()
        self._pending[b] = None

    def ResetDevice(self):
        self._finish(self._current_buffer())               # a reset rewrites the current buffer in place
        self.local.ResetDevice()

    def StepDevice(self, d_actions):
        """Step this rank's lanes; d_actions: this rank's slice of the global action array (device pointer/tensor)."""
        if self.overlap:
            self._finish(self._current_buffer() ^ 1)       # the step writes the OTHER buffer: its last gather must be done
            self._cur ^= 1
        self.local.StepDevice(d_actions)

    def AllGatherObs(self, async_op=False, overlap=False):
        """Everyone's observations into the current gather buffer [G][D][N/G].  The send buffer is slice [rank] of that
        buffer itself (in place).  overlap=True (needs overlap=True at construction): the collective runs on a side
        stream and only WaitGather() / the step that will overwrite this buffer wait for it."""
        if not self.gather_obs:
            return None
        dist, b = self._dist, self._current_buffer()
        t = self.obs_bufs[b]
        flat_out = t.view(-1)
        flat_in = t[self.rank].reshape(-1)
        self._last = b
        if self.gather == "direct":
            tc = self._torch.cuda
            main = tc.current_stream(self.tensor_device)
            # The barrier is a HOST barrier (`barrier` may be a shared-memory spin barrier or gloo): it says nothing about work
            # still queued on this rank's stream — e.g. a policy kernel reading the replica a peer is about to push into.  So
            # the stream is drained first: once every rank has passed the barrier, every rank's readers of the old slices have
            # FINISHED (a cross-process write-after-read race otherwise).  Contract for a custom `barrier`: it is
            # always called right after a stream synchronize.
            main.synchronize()
            self._barrier()                                   # every rank is done reading the replicas about to be overwritten
            if overlap and self.overlap:
                self._ev_step.record(main)
                self._gstream.wait_event(self._ev_step)       # the step that produced this buffer
                self._push(b, self._gstream.cuda_stream)
                self._ev_done[b].record(self._gstream)
                self._pending[b] = self._ev_done[b]
                self._direct_inflight = True
            else:
                self._push(b, main.cuda_stream)
                main.synchronize()
                self._barrier()                               # every rank's pushes have landed
            return None

        def gather(async_flag):
            try:
                return dist.all_gather_into_tensor(flat_out, flat_in, group=self.group, async_op=async_flag)
            except (RuntimeError, NotImplementedError):      # backends without the flat form (older gloo)
                outs = [t[r].view(-1) for r in range(self.world_size)]
                return dist.all_gather(outs, flat_in.clone(), group=self.group, async_op=async_flag)

        if overlap and self.overlap:
            if self._cuda:
                tc = self._torch.cuda
                main = tc.current_stream(self.tensor_device)
                self._ev_step.record(main)
                tc.set_stream(self._gstream)
                try:
                    self._gstream.wait_event(self._ev_step)   # the step that produced this buffer
                    work = gather(True)
                    work.wait()                               # side stream waits for RCCL's stream; the host does not
                    self._ev_done[b].record(self._gstream)
                finally:
                    tc.set_stream(main)
                self._pending[b] = self._ev_done[b]
            else:
                self._pending[b] = gather(True)
            return None
        work = gather(async_op)
        self._work = work if async_op else None
        return work

    def Wait(self):
        if self._work is not None:
            self._work.wait()
            self._work = None

    def WaitGather(self):
        """After this, work queued on the main stream sees the last gathered buffer complete."""
        self.Wait()
        if self.gather == "direct" and getattr(self, "_direct_inflight", False):
            self._gstream.synchronize()
            self._barrier()                                   # every rank's pushes have landed
            self._direct_inflight = False
        self._finish(self._last)

    def GlobalObs(self):
        """[N, D]-shaped logical view of the gathered observations as a [G, D, N/G] tensor:
        global lane g = r * (N/G) + i  ->  GlobalObs()[r, :, i].  (The buffer gathered last.)"""
        return self.obs_bufs[self._last]

    LastGatheredObs = GlobalObs

    def Sync(self):
        self.local.Sync()
        if self._gstream is not None:
            self._gstream.synchronize()

    def Close(self):
        self.local.Close()
        if self._peers is not None:
            if self._cuda:
                self._torch.cuda.synchronize(self.tensor_device)
            self._barrier()                                   # nobody still pushes into a buffer that is about to go away
            import sys
            for r, p in self._peers.items():
                if self._lib.gymnet_peer_buffer_close(self._peer_dev, p) != 0:
                    sys.stderr.write(f"gym.net_amd: closing rank {r}'s peer buffer failed on rank {self.rank}\n")
            self._peers = None
            self.obs_bufs = self.obs_all = None
            self._barrier()                                   # every importer has unmapped before any exporter frees
            if self._lib.gymnet_peer_buffer_destroy(self._peer_dev, self._own_base) != 0:
                sys.stderr.write(f"gym.net_amd: freeing rank {self.rank}'s peer buffer failed\n")
