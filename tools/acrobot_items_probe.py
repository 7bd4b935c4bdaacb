#!/usr/bin/env python3
"""GPU probe: Acrobot one-shot kernel vs the multi-lane kernel (GYMNET_ITEMS = lanes per thread: all loads first, then
compute / store lane after lane), several batch sizes.  us per 2^20 lanes (HIP events, best of 5), state hash."""
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
ring = 64
os.environ["GYMNET_VEC"] = "1"
for n in (1 << 19, 3 << 18, 1 << 20, 5 << 18, 3 << 19, 1 << 21, 1 << 22):
    logn = n / (1 << 20)
    acts = torch.empty((ring, n), dtype=torch.int32, device=dev)
    for items in (1, 2, 3, 4, 5, 6, 8):
        os.environ["GYMNET_ITEMS"] = str(items)
        steps = max(256, (1 << 31) // n)
        with pkg.VectorEnv("Acrobot-v1", n, seed=1, auto_reset=True, stream=stream.cuda_stream) as e:
            for t in range(ring):
                e.SampleActionsDevice(acts[t], seed=3, tick=t)
            e.ResetDevice()
            e.RolloutDevice(acts, 128, n, ring)
            e.Sync()
            best = 1e9
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                e.RolloutDevice(acts, steps, n, ring)
                e1.record(stream)
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1e3 / steps)
            h = hashlib.sha256(e.GetState().tobytes()).hexdigest()[:10]
        print(f"n={logn:5.2f}*2^20 lanes/thread={items}  {best:8.3f} us/step = {best * (1 << 20) / n:7.3f} us per 2^20 lanes  ({65 * n / best / 1e6:5.2f} TB/s)  state {h}", flush=True)
