#!/usr/bin/env python3
"""GPU probe: what does GYMNET_FLAG_DOUBLE_BUFFER cost the one-step kernel, and does it depend on how far apart the two
buffers are (same-channel / same-bank aliasing of the read stream and the write stream)?  us per step, HIP events."""
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
name = sys.argv[1] if len(sys.argv) > 1 else "CartPole-v1"
n, ring, steps = 1 << 20, 64, 4096
D = {"CartPole-v1": 4, "Pendulum-v1": 3, "MountainCar-v0": 2, "Acrobot-v1": 6}[name]
adt = torch.float32 if name == "Pendulum-v1" else torch.int32
acts = torch.empty((ring, n), dtype=adt, device=dev)


def timed(env):
    for t in range(ring):
        env.SampleActionsDevice(acts[t], seed=3, tick=t)
    env.ResetDevice()
    env.RolloutDevice(acts, 512, n, ring)
    env.Sync()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        env.RolloutDevice(acts, steps, n, ring)
        e1.record(stream)
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / steps)
    return best


with pkg.VectorEnv(name, n, seed=1, auto_reset=True, stream=stream.cuda_stream) as e:
    print(f"{name} in place (library buffers)            {timed(e):7.3f} us/step", flush=True)
with pkg.VectorEnv(name, n, seed=1, auto_reset=True, stream=stream.cuda_stream, double_buffer=True) as e:
    v = e.DeviceView()
    print(f"{name} double buffer (library buffers, delta {abs(v.d_obs_alt - v.d_obs)} B)   {timed(e):7.3f} us/step", flush=True)
for skew in (0, 256, 4096, 4096 + 256, 65536, 65536 + 4096, (1 << 20) + 4096, (2 << 20) + 8192 + 256):
    words = D * n + skew // 4
    buf = torch.zeros(2 * words + 64, dtype=torch.float32, device=dev)
    a, b = buf[:D * n], buf[words:words + D * n]
    with pkg.VectorEnv(name, n, seed=1, auto_reset=True, stream=stream.cuda_stream, double_buffer=True,
                       ext_obs=a, ext_obs_stride=n, ext_obs_alt=b) as e:
        print(f"{name} double buffer, external, skew {skew:8d} B  {timed(e):7.3f} us/step", flush=True)
    with pkg.VectorEnv(name, n, seed=1, auto_reset=True, stream=stream.cuda_stream, ext_obs=a, ext_obs_stride=n) as e:
        if skew == 0:
            print(f"{name} in place, external buffer              {timed(e):7.3f} us/step", flush=True)
for nt in ("0", "12", "15"):
    os.environ["GYMNET_NT"] = nt
    with pkg.VectorEnv(name, n, seed=1, auto_reset=True, stream=stream.cuda_stream, double_buffer=True) as e:
        print(f"{name} double buffer, GYMNET_NT={nt:2s}               {timed(e):7.3f} us/step", flush=True)
    with pkg.VectorEnv(name, n, seed=1, auto_reset=True, stream=stream.cuda_stream) as e:
        print(f"{name} in place,      GYMNET_NT={nt:2s}               {timed(e):7.3f} us/step", flush=True)
