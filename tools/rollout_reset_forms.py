#!/usr/bin/env python3
"""Round 5 probe (runs ON THE GPU BOX): the fused rollout's per-step reset — the per-thread drain loop (launch policy reset_form = 0)
against the wave-compacted reset (reset_form = 1: float32 one Philox call per reset, float64 two lanes per reset) — at 2^20 CartPole
lanes, 256 steps per launch, us per vector step; float32 and float64, lean and bookkeeping handles, ring and sampled actions.
Every pair is checked to leave bit-identical state behind."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
n, ring, seed = 1 << 20, 256, 0x5EED
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(dev)
torch.cuda.set_stream(stream)


def timed(fn, launches=6, reps=3):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for _ in range(launches):
            fn()
        b.record(stream)
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / (launches * ring))
    return sorted(ts)[len(ts) // 2]


acts = torch.empty((ring, n), dtype=torch.int32, device=dev)
with pkg.VectorEnv("CartPole-v1", n, seed=seed, auto_reset=True, stream=stream.cuda_stream) as e:
    for t in range(ring):
        e.SampleActionsDevice(acts[t], seed=seed + 1, tick=t)
torch.cuda.synchronize()

for dtype in ("float32", "float64"):
    for label, kw in (("lean", {}), ("EPISODE_STATS, time limit 500", dict(episode_stats=True, max_episode_steps=500))):
        row, states = [], []
        for rf in (0, 1):
            with pkg.VectorEnv("CartPole-v1", n, seed=seed, auto_reset=True, stream=stream.cuda_stream, dtype=dtype, **kw) as e:
                e.SetLaunchPolicy(reset_form=rf)
                e.ResetDevice()
                e.RolloutFusedDevice(acts, ring, n, ring)
                e.RolloutFusedDevice(None, 64, actions="sample", action_seed=7)
                e.Sync()
                states.append(e.GetState().copy())
                e.ResetDevice()
                t_ring = timed(lambda: e.RolloutFusedDevice(acts, ring, n, ring))
                t_samp = timed(lambda: e.RolloutFusedDevice(None, ring, actions="sample", action_seed=7))
                row.append((t_ring, t_samp))
        same = np.array_equal(states[0].view(np.uint8), states[1].view(np.uint8))
        print(f"{dtype:8s} {label:32s} ring actions: drain {row[0][0]:6.3f}  compacted {row[1][0]:6.3f}   sampled actions: drain {row[0][1]:6.3f}  "
              f"compacted {row[1][1]:6.3f} us/step   state after both: {'bit-identical' if same else 'DIFFERENT'}", flush=True)
