#!/usr/bin/env python3
"""GPU probe: the grid-stride, software-pipelined step kernel (GYMNET_ITEMS lanes per thread) against the one-shot kernel,
per env at 2^20 lanes: us per step (HIP events, best of 5 x 2048 steps) and a hash of the final state (must not change)."""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(dev)
torch.cuda.set_stream(stream)
n, ring, steps = 1 << 20, 64, 2048
envs = sys.argv[1:] or ["Acrobot-v1", "CartPole-v1", "Pendulum-v1", "MountainCar-v0"]
for name in envs:
    adt = torch.float32 if name == "Pendulum-v1" else torch.int32
    acts = torch.empty((ring, n), dtype=adt, device=dev)
    ref = None
    for vec, items in ((None, 1), (1, 1), (1, 2), (1, 3), (1, 4), (1, 6), (1, 8), (1, 12), (1, 16)):
        os.environ.pop("GYMNET_VEC", None)
        if vec is not None:
            os.environ["GYMNET_VEC"] = str(vec)
        os.environ["GYMNET_ITEMS"] = str(items)
        with pkg.VectorEnv(name, n, seed=1, auto_reset=True, stream=stream.cuda_stream) as e:
            for t in range(ring):
                e.SampleActionsDevice(acts[t], seed=3, tick=t)
            e.ResetDevice()
            e.RolloutDevice(acts, 256, n, ring)
            e.Sync()
            best = 1e9
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                e.RolloutDevice(acts, steps, n, ring)
                e1.record(stream)
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1e3 / steps)
            h = hashlib.sha256(e.GetState().tobytes()).hexdigest()[:12]
            ref = ref or h
            pol = e.LaunchPolicy()
            print(f"{name:15s} policy {pol}  {best:7.3f} us/step  state {h} {'OK' if h == ref else 'MISMATCH'}", flush=True)
