#!/usr/bin/env python3
"""GPU probe: the non-temporal stream mask (GYMNET_NT: 0 none, 12 action + reward/done, 15 every stream) x lanes per thread,
per env at 2^20 lanes.  us per step, HIP events, best of 5 x 2048 steps."""
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
for name in (sys.argv[1:] or ["Acrobot-v1", "Pendulum-v1", "MountainCar-v0", "CartPole-v1"]):
    adt = torch.float32 if name == "Pendulum-v1" else torch.int32
    acts = torch.empty((ring, n), dtype=adt, device=dev)
    for vec in ((1, 2) if name == "Acrobot-v1" else (1, 4)):
        for nt in (0, 12, 15):
            os.environ["GYMNET_VEC"] = str(vec)
            os.environ["GYMNET_NT"] = str(nt)
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
                print(f"{name:15s} {e.LaunchPolicy()}  {best:7.3f} us/step", flush=True)
