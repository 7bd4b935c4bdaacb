#!/usr/bin/env python3
"""GPU probe: what a short bracketed region costs beyond its kernels.  wall(K) for K = 1..64 back-to-back CartPole steps at
2^20 lanes with three ways of waiting for completion; the K -> 0 intercept is the fixed cost of a bracket."""
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_INTERRUPT", os.environ.get("GYMNET_BENCH_HSA_INTERRUPT", "0"))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(dev)
torch.cuda.set_stream(stream)
n, ring = 1 << 20, 64
env = pkg.VectorEnv("CartPole-v1", n, seed=1, auto_reset=True, stream=stream.cuda_stream)
acts = torch.empty((ring, n), dtype=torch.int32, device=dev)
for t in range(ring):
    env.SampleActionsDevice(acts[t], seed=3, tick=t)
env.ResetDevice()
env.RolloutDevice(acts, 512, n, ring)
env.Sync()


def med(xs):
    s = sorted(xs)
    return s[len(s) // 2]


def wait_device():
    torch.cuda.synchronize(dev)


def wait_stream():
    stream.synchronize()


def wait_engine():
    env.Sync()


ev = torch.cuda.Event()


def wait_event_poll():
    ev.record(stream)
    while not ev.query():
        pass


for name, wait in (("torch.cuda.synchronize", wait_device), ("stream.synchronize", wait_stream), ("gymnet_vecenv_sync", wait_engine),
                   ("event record + query poll", wait_event_poll)):
    row = []
    for K in (1, 2, 5, 10, 20, 40, 64):
        ts = []
        for _ in range(300):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            env.RolloutDevice(acts, K, n, ring)
            wait()
            ts.append(time.perf_counter() - t0)
        row.append((K, med(ts) * 1e6))
    # least-squares line through K >= 5
    pts = [(k, t) for k, t in row if k >= 5]
    mk = sum(k for k, _ in pts) / len(pts)
    mt = sum(t for _, t in pts) / len(pts)
    slope = sum((k - mk) * (t - mt) for k, t in pts) / sum((k - mk) ** 2 for k, _ in pts)
    print(f"{name:28s} " + "  ".join(f"K={k}: {t:6.1f}us" for k, t in row) + f"   slope {slope:.2f} us/step, intercept {mt - slope * mk:.1f} us", flush=True)
