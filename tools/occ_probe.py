#!/usr/bin/env python3
"""GPU probe: does capping the one-shot step kernel's occupancy (more, smaller wave generations -> shorter exposed load / store
bursts at the head and tail of the launch) help?  GYMNET_LDS requests unused dynamic LDS per workgroup to cap occupancy."""
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
for name in (sys.argv[1:] or ["Acrobot-v1", "CartPole-v1"]):
    adt = torch.float32 if name == "Pendulum-v1" else torch.int32
    acts = torch.empty((ring, n), dtype=adt, device=dev)
    for block in (256, 128, 64):
        for lds_kb in (0, 24, 27, 32, 40, 54, 80):
            os.environ["GYMNET_ITEMS"] = "1"
            os.environ["GYMNET_BLOCK"] = str(block)
            os.environ["GYMNET_LDS"] = str(lds_kb * 1024)
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
                wg_per_cu = min(2048 // block, (160 // lds_kb) if lds_kb else 99)
                print(f"{name:12s} block {block:3d} lds {lds_kb:3d} KiB -> <= {wg_per_cu * block // 256:2d} waves/SIMD   {best:7.3f} us/step", flush=True)
