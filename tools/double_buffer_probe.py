#!/usr/bin/env python3
"""Round 5 probe (runs ON THE GPU BOX): does writing the new state to the OTHER half of a ping-pong pair (GYMNET_FLAG_DOUBLE_BUFFER) change the
step kernel's time against the in-place update?  The step kernels are write-bound (profiles/write_path_probe_r02.txt); us per vector step,
2^20 lanes, eager rollout of 2048 steps, median of 5."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
n, ring, steps = 1 << 20, 64, 2048
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(dev)
torch.cuda.set_stream(stream)
for env_id, dtype in (("CartPole-v1", "float32"), ("CartPole-v1", "float64"), ("MountainCar-v0", "float32"), ("Pendulum-v1", "float32"), ("Acrobot-v1", "float32")):
    box = env_id.startswith("Pendulum")
    acts = (torch.rand((ring, n), device=dev) * 4 - 2) if box else torch.randint(0, 2, (ring, n), dtype=torch.int32, device=dev)
    row = []
    for db in (False, True):
        kw = dict(dtype=dtype) if dtype == "float64" else {}
        with pkg.VectorEnv(env_id, n, seed=0x5EED, auto_reset=True, stream=stream.cuda_stream, double_buffer=db, **kw) as e:
            e.ResetDevice()
            e.RolloutDevice(acts, 256, n, ring); torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(stream); e.RolloutDevice(acts, steps, n, ring); b.record(stream); torch.cuda.synchronize()
                ts.append(a.elapsed_time(b) * 1e3 / steps)
            row.append((sorted(ts)[2], e.KernelName()))
    print(f"{env_id:16s} {dtype:8s} in place {row[0][0]:7.3f} us   ping-pong {row[1][0]:7.3f} us   {row[0][1]}", flush=True)
