#!/usr/bin/env python3
"""Round 5 probe (runs ON THE GPU BOX): host-boundary latency of ONE CartPole instance, us per Step() call through the C ABI
(ctypes; the C# P/Invoke cost is of the same order): the launch path (2 kernel launches + 1 stream synchronize per step) against
GYMNET_FLAG_RESIDENT (mailbox in pinned host memory, no launch), float32 and float64, plus the raw ABI call without Python's
array handling."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
for dtype in (np.float32, np.float64):
    for resident in (False, True):
        with pkg.VectorEnv("CartPole-v1", 1, seed=1, dtype=dtype, resident=resident) as e:
            e.Reset()
            a = np.zeros(1, np.int32)
            obs, rew, done = np.empty((1, 4), dtype), np.empty(1, np.float32), np.empty(1, np.uint8)
            for _ in range(2000):
                e.StepInto(a, obs, rew, done)
                if done[0]:
                    e.ResetInto(obs)
            t0 = time.perf_counter()
            K = 20000
            lib, h = e._lib, e._h
            pa, po, pr, pd = (x.ctypes.data_as(C.c_void_p) for x in (a, obs, rew, done))
            for i in range(K):
                a[0] = i & 1
                lib.gymnet_vecenv_step(h, pa, po, pr, pd)
                if done[0]:
                    lib.gymnet_vecenv_reset(h, po)
            dt = time.perf_counter() - t0
            print(f"CartPole-v1 N=1 {np.dtype(dtype).name:8s} {'resident mailbox' if resident else 'launch path     '}: {dt / K * 1e6:7.2f} us per iteration "
                  f"(raw ABI calls from ctypes, reset-on-done included) = {K / dt:9.0f} steps/s")
