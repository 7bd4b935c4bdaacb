"""Prints a SHA-256 of the state after a long fused + stepwise rollout: identical across runs and boxes."""
import hashlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
n, ring = 1 << 20, 32
for name in ("CartPole-v1", "Pendulum-v1", "MountainCar-v0", "Acrobot-v1"):
    with pkg.VectorEnv(name, n, seed=0x5EED, auto_reset=True) as env:
        adt = torch.float32 if name == "Pendulum-v1" else torch.int32
        acts = torch.empty((ring, n), dtype=adt, device="cuda"); torch.cuda.synchronize()
        for t in range(ring):
            env.SampleActionsDevice(acts[t], seed=1, tick=t)
        env.ResetDevice()
        env.RolloutDevice(acts, 1000, n, ring)
        env.RolloutFusedDevice(acts, 1000, n, ring)
        env.Sync()
        s = env.GetState()
        print(name, hashlib.sha256(s.tobytes()).hexdigest()[:24], "tick", env.Tick)
