#!/usr/bin/env python3
"""GPU probe (round 4): does it matter WHERE a handle's observation arrays live?  Acrobot at 2^20 lanes, library-allocated
arrays vs a torch-allocated external observation buffer (what ShardedVectorEnv / bench.py's headline handle uses), each in
turn, twice, in one process; optionally after a CartPole batch of the same size has been created and stepped (the situation
of bench.py's in-line per-configuration figures).   python tools/acrobot_alloc_probe.py [--after-cartpole]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
dev = torch.device("cuda", 0)
n, ring = 1 << 20, 32
stream = torch.cuda.Stream(dev)
torch.cuda.set_stream(stream)
keep = []
if "--after-cartpole" in sys.argv:
    cp = pkg.VectorEnv("CartPole-v1", n, seed=1, auto_reset=True, stream=stream.cuda_stream)
    ca = torch.empty((256, n), dtype=torch.int32, device=dev)
    for t in range(256):
        cp.SampleActionsDevice(ca[t].data_ptr(), seed=2, tick=t)
    cp.ResetDevice(); cp.RolloutDevice(ca.data_ptr(), 4096, n, 256); cp.Sync()
    keep += [cp, ca]


def timed(e, a):
    e.ResetDevice(); e.RolloutDevice(a.data_ptr(), 128, n, ring); e.Sync(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); e.RolloutDevice(a.data_ptr(), 1024, n, ring); e.Sync(); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 1024 * 1e6)
    return " ".join(f"{x:.2f}" for x in ts)


for rep in range(2):
    for mode in ("library", "torch-ext-obs"):
        ext = torch.zeros((6, n), dtype=torch.float32, device=dev) if mode != "library" else None
        kw = {} if ext is None else {"ext_obs": ext.data_ptr(), "ext_obs_stride": n}
        with pkg.VectorEnv("Acrobot-v1", n, seed=1, auto_reset=True, stream=stream.cuda_stream, **kw) as e:
            a = torch.empty((ring, n), dtype=torch.int32, device=dev)
            for t in range(ring):
                e.SampleActionsDevice(a[t].data_ptr(), seed=2, tick=t)
            v = e.DeviceView()
            print(f"{mode:14s} d_obs %#x d_state %#x  {e.KernelName()}  us/step: {timed(e, a)}" % (v.d_obs, v.d_state), flush=True)
            del a
