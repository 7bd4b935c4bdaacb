#!/usr/bin/env python3
"""GPU probe (ONE GPU, logical members): the group path's costs that do not need xGMI — one vector step of a 2^23-lane batch
as 8 members of 2^20 (8 launches from one host thread), the direct all-gather's push kernels writing 8 x 7 x 16 MiB inside
the device (what the push kernel does when HBM, not a link, is the limit), and the overlapped form.  Wall clock per step
over 200 steps, group synchronised at both ends."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
G, n, ring, steps = 8, 1 << 23, 8, 200
nl = n // G
acts = torch.empty((ring, n), dtype=torch.int32, device="cuda")
with pkg.VectorEnv("CartPole-v1", n, seed=1, auto_reset=True) as one:
    for t in range(ring):
        one.SampleActionsDevice(acts[t], seed=3, tick=t)
    one.ResetDevice()
    one.RolloutDevice(acts, 32, n, ring); one.Sync()
    t0 = time.perf_counter(); one.RolloutDevice(acts, steps, n, ring); one.Sync()
    print(f"single handle, 2^23 lanes:                       {(time.perf_counter() - t0) / steps * 1e6:8.1f} us/step", flush=True)
slices = [[acts[t, m * nl:(m + 1) * nl] for m in range(G)] for t in range(ring)]
for overlap in (False, True):
    with pkg.GroupVectorEnv("CartPole-v1", n, G, devices=[0] * G, seed=1, auto_reset=True, gather="direct", overlap=overlap) as grp:
        grp.ResetDevice(); grp.Sync()
        for mode in ("step only", "step + gather"):
            for warm in (True, False):
                t0 = time.perf_counter()
                for t in range(20 if warm else steps):
                    grp.StepDevice(slices[t % ring])
                    if mode != "step only":
                        grp.AllGatherObs()
                grp.WaitGather(); grp.Sync()
                dt = (time.perf_counter() - t0) / steps * 1e6
            print(f"group of 8 logical members, overlap={overlap!s:5}, {mode:14s} {dt:8.1f} us/step"
                  + (f"   (the pushes alone move 8 x 7 x 16 MiB = 896 MiB per step inside the device)" if mode != "step only" else ""), flush=True)
