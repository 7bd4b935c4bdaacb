#!/usr/bin/env python3
"""GPU probe for the ALU-bound env: one lane per thread (scalar FP32) vs two lanes per thread on packed FP32, across batch
sizes (more wave generations = more overlap of memory and arithmetic) and occupancy caps (GYMNET_LDS: unused LDS per block).
Prints us per 2^20 lanes (HIP events, best of 5) and a state hash (must not depend on the variant)."""
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
name = "Acrobot-v1"
ring = 16


def run(n, vec, lds_kb=0, block=256, steps=None):
    os.environ["GYMNET_VEC"] = str(vec)
    os.environ["GYMNET_LDS"] = str(lds_kb * 1024)
    os.environ["GYMNET_BLOCK"] = str(block)
    steps = steps or max(64, (1 << 31) // n)
    acts = torch.empty((ring, n), dtype=torch.int32, device=dev)
    with pkg.VectorEnv(name, n, seed=1, auto_reset=True, stream=stream.cuda_stream) as e:
        for t in range(ring):
            e.SampleActionsDevice(acts[t], seed=3, tick=t)
        e.ResetDevice()
        e.RolloutDevice(acts, 64, n, ring)
        e.Sync()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            e.RolloutDevice(acts, steps, n, ring)
            e1.record(stream)
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / steps)
        h = hashlib.sha256(e.GetState()[:, :4096].tobytes()).hexdigest()[:10]
    return best, h


for logn in (18, 19, 20, 21, 22, 23):
    n = 1 << logn
    for vec in (1, 2):
        t, h = run(n, vec)
        print(f"n=2^{logn} vec={vec}  {t:9.3f} us/step  = {t * (1 << 20) / n:7.3f} us per 2^20 lanes  {65 * n / t / 1e6:6.2f} TB/s  state {h}", flush=True)
for vec in (1, 2):
    for block in (256, 128, 64):
        for lds in (0, 20, 27, 40, 54):
            t, h = run(1 << 20, vec, lds, block)
            print(f"n=2^20 vec={vec} block={block:3d} lds={lds:2d}KiB  {t:7.3f} us/step  state {h}", flush=True)
