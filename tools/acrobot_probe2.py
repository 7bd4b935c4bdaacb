#!/usr/bin/env python3
"""Which part of the two-lanes-per-thread Acrobot kernel is slow: the dwordx2 accesses or the packed arithmetic?
Usage: GYMNET_LIB_PATH=... python tools/acrobot_probe2.py <label>

The probe builds of the library it is pointed at (never shipped; same sources, one macro each):
  cd gym.net_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared \
      kernels.hip capi.hip group.hip -ldl -DGYMNET_PROBE_NO_PACK -o ../lib/libgymnet_amd_nopack.so          # two lanes per thread, scalar math
  ... -DGYMNET_PROBE_NO_PACK -DGYMNET_PROBE_ACROBOT_NOMATH -o ../lib/libgymnet_amd_nomath.so                  # Acrobot's traffic, no arithmetic
"""
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
label = sys.argv[1] if len(sys.argv) > 1 else ""
ring = 16
for logn in (20, 22):
    n = 1 << logn
    for vec in (1, 2):
        for nt in (15, 0):
            os.environ["GYMNET_VEC"] = str(vec)
            os.environ["GYMNET_NT"] = str(nt)
            steps = (1 << 31) // n
            acts = torch.empty((ring, n), dtype=torch.int32, device=dev)
            with pkg.VectorEnv("Acrobot-v1", n, seed=1, auto_reset=True, stream=stream.cuda_stream) as e:
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
            print(f"{label:8s} n=2^{logn} vec={vec} nt={nt:2d}  {best * (1 << 20) / n:7.3f} us per 2^20 lanes", flush=True)
