"""Times the fused T-step rollout kernel against the one-launch-per-step path (same workload as bench.py)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

pkg = ge.load_package()
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(dev)
torch.cuda.set_stream(st)
out = {}
ENVS = sys.argv[1:] or ["CartPole-v1", "Pendulum-v1", "MountainCar-v0", "Acrobot-v1"]
for name in ENVS:
    n, ring, T = 1 << 20, 64, 1024
    env = pkg.VectorEnv(name, n, seed=1, auto_reset=True, stream=st.cuda_stream)
    adt = torch.float32 if name == "Pendulum-v1" else torch.int32
    acts = torch.empty((ring, n), dtype=adt, device=dev)
    for t in range(ring):
        env.SampleActionsDevice(acts[t], seed=2, tick=t)
    D = env.ObsDim
    rec_obs = torch.empty((ring, D, n), dtype=torch.float32, device=dev)      # recording ring (T x N x D x 4 B would be 16 GiB)
    rec_rew = torch.empty((ring, n), dtype=torch.float32, device=dev)
    rec_done = torch.empty((ring, n), dtype=torch.uint8, device=dev)
    res = {}
    for mode in ("stepwise", "fused", "fused+record"):
        env.ResetDevice()
        def run(steps):
            if mode == "stepwise":
                env.RolloutDevice(acts, steps, n, ring)
            elif mode == "fused":
                for c in range(steps // ring):                       # `ring` steps per launch
                    env.RolloutFusedDevice(acts, ring, n, ring)
            else:
                for c in range(steps // ring):
                    env.RolloutFusedDevice(acts, ring, n, ring, rec_obs, rec_rew, rec_done)
        run(128)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st); run(T); e1.record(st); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / T
        res[mode] = {"us_per_step": us, "env_steps_per_sec": n / (us * 1e-6)}
    out[name] = res
    env.Close()
print(json.dumps(out))
