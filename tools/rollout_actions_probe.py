#!/usr/bin/env python3
"""Round 6 probe (runs ON THE GPU BOX): what the action source costs the fused rollout after action stream v2 (one Philox call per
group of four lanes) — us per vector step at 2^20 CartPole lanes, 256 steps per launch, for ring / ActionSpace.Sample() drawn in the
kernel / epsilon-greedy over the ring, float32 and float64, lean and bookkeeping handles; and the stand-alone samplers' launch time.
Also prints the four full-size checksum pins of tests/test_gpu_other_envs.py (their action rings are device-sampled)."""
import hashlib
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


def timed(fn, per=ring, launches=8, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for _ in range(launches):
            fn()
        b.record(stream)
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / (launches * per))
    return sorted(ts)[len(ts) // 2]


acts = torch.empty((ring, n), dtype=torch.int32, device=dev)
for dt in ("float32", "float64"):
    for label, kw in (("lean", {}), ("EPISODE_STATS, time limit 500", dict(episode_stats=True, max_episode_steps=500))):
        with pkg.VectorEnv("CartPole-v1", n, seed=seed, auto_reset=True, dtype=dt, stream=stream.cuda_stream, **kw) as e:
            for t in range(ring):
                e.SampleActionsDevice(acts[t], seed=seed + 1, tick=t)
            e.ResetDevice()
            r = timed(lambda: e.RolloutFusedDevice(acts, ring, n, ring))
            s = timed(lambda: e.RolloutFusedDevice(None, ring, actions="sample", action_seed=7))
            g = timed(lambda: e.RolloutFusedDevice(acts, ring, n, ring, actions="epsilon_greedy", action_seed=7, epsilon=0.1))
            print(f"{dt}  {label:32s} ring {r:6.3f}   sampled {s:6.3f}   epsilon-greedy {g:6.3f} us/step   (eps-greedy / sampled {g / s:.3f})")
with pkg.VectorEnv("CartPole-v1", n, seed=seed, auto_reset=True, stream=stream.cuda_stream) as e:
    out = torch.empty(n, dtype=torch.int32, device=dev)
    s = timed(lambda: e.SampleActionsDevice(out, seed=3, tick=5), per=1, launches=64)
    c = timed(lambda: e.ComposeActionsDevice(acts[0], 0.1, out, seed=3, tick=5), per=1, launches=64)
    print(f"stand-alone at 2^20 lanes: SampleActionsDevice {s:6.2f} us, ComposeActionsDevice {c:6.2f} us per launch")
for name in ("CartPole-v1", "Pendulum-v1", "MountainCar-v0", "Acrobot-v1"):
    with pkg.VectorEnv(name, n, seed=seed, auto_reset=True) as env:
        adt = torch.float32 if name == "Pendulum-v1" else torch.int32
        a32 = torch.empty((32, n), dtype=adt, device="cuda")
        torch.cuda.synchronize()
        for t in range(32):
            env.SampleActionsDevice(a32[t], seed=1, tick=t)
        env.ResetDevice()
        env.RolloutDevice(a32, 1000, n, 32)
        env.RolloutFusedDevice(a32, 1000, n, 32)
        env.Sync()
        print(f'    "{name}": "{hashlib.sha256(env.GetState().tobytes()).hexdigest()[:24]}",')
