#!/usr/bin/env python3
"""Round 6 probe (runs ON THE GPU BOX): float64 CartPole by batch size — the one-shot kernel against the multi-pair kernel, whose form
with the deferred reset now LOOPS over generations (step_kernels.hpp pipe2_shape) instead of launching more workgroups than the chip
holds.  us per vector step by HIP events over back-to-back launches, per 2^20 lanes beside it, and the state after the same number of
steps compared bit for bit with the one-shot kernel's.   python tools/f64_sizes_probe.py [f32]  (f32: the float32 sizes table)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(dev)
torch.cuda.set_stream(stream)
f32 = len(sys.argv) > 1 and sys.argv[1] != "f64"
env_name = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] not in ("f32", "f64") else "CartPole-v1"     # an env name: its float32 table
sizes = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1 << 20, 3 << 19, 1 << 21, 3 << 20, 1 << 22, 1 << 23, (1 << 21) + 6, 5 << 20]
ring = 4


def timed(e, acts, n, launches, reps=5):
    e.RolloutDevice(acts.data_ptr(), launches, n, ring)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        e.RolloutDevice(acts.data_ptr(), launches, n, ring)
        b.record(stream)
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / launches)
    return sorted(ts)[len(ts) // 2]


for n in sizes:
    acts = torch.empty((ring, n), dtype=torch.float32 if env_name == "Pendulum-v1" else torch.int32, device=dev)
    launches = max(64, min(1024, (1 << 28) // n))
    ref = None
    if f32:
        policies = [("default", {}), ("nt=15", dict(nt=15)), ("nt=12", dict(nt=12)), ("nt=0", dict(nt=0))]
    else:
        policies = [("default", {}), ("one-shot nt=12", dict(sequential_lanes=1, nt=12)), ("one-shot nt=15", dict(sequential_lanes=1, nt=15)), ("one-shot nt=0", dict(sequential_lanes=1, nt=0)),
                    ("one-shot nt=0 rf=0", dict(sequential_lanes=1, nt=0, reset_form=0)), ("one-shot nt=15 rf=0", dict(sequential_lanes=1, nt=15, reset_form=0)),
                    ("2 pairs", dict(sequential_lanes=2)), ("3 pairs", dict(sequential_lanes=3)), ("4 pairs nt=15", dict(sequential_lanes=4, nt=15)),
                    ("4 pairs nt=12", dict(sequential_lanes=4, nt=12)), ("4 pairs nt=0", dict(sequential_lanes=4, nt=0))]
    for label, pol in policies:
        with pkg.VectorEnv(env_name, n, seed=7, auto_reset=True, dtype="float32" if f32 else "float64", stream=stream.cuda_stream) as e:
            for t in range(ring):
                e.SampleActionsDevice(acts[t], seed=8, tick=t)
            try:
                if pol:
                    e.SetLaunchPolicy(**pol)
            except Exception as ex:                                   # noqa: BLE001
                print(f"n = {n:9d}  {label:16s} refused: {ex}")
                continue
            e.ResetDevice()
            e.RolloutDevice(acts.data_ptr(), 37, n, ring)
            e.Sync()
            st = e.GetState()
            same = "reference" if ref is None else ("bit-identical" if np.array_equal(st, ref) else "DIFFERS")
            if ref is None:
                ref = st
            us = timed(e, acts, n, launches)
            print(f"n = {n:9d}  {label:16s} {e.KernelName():44s} {us:9.3f} us/step  {us * (1 << 20) / n:7.3f} per 2^20 lanes   {same}", flush=True)
    del acts
    torch.cuda.empty_cache()
