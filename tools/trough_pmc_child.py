#!/usr/bin/env python3
"""Child of tools/gpu_trough_r06.sh (runs under rocprofv3 --pmc ON THE GPU BOX): CartPole float32, default launch policy, 30 eager
one-launch steps at each of the batch sizes given (default 2^20, 2^21, 2^22, 2^24 lanes) — the dispatches are told apart by grid size."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1 << 20, 1 << 21, 1 << 22, 1 << 24]
pol = dict(kv.split("=") for kv in sys.argv[2].split(",")) if len(sys.argv) > 2 and sys.argv[2] else {}
for n in sizes:
    acts = torch.empty((2, n), dtype=torch.int32, device="cuda")
    with pkg.VectorEnv("CartPole-v1", n, seed=7, auto_reset=True) as e:
        if pol:
            e.SetLaunchPolicy(**{k: int(v) for k, v in pol.items()})
        for t in range(2):
            e.SampleActionsDevice(acts[t], seed=8, tick=t)
        e.ResetDevice()
        for t in range(30):
            e.StepDevice(acts[t % 2])
        e.Sync()
        print(n, e.KernelName(), e.LaunchPolicy(), flush=True)
    del acts
