"""PCIe-inclusive rates of the HOST-boundary path (numpy in, numpy out), for DESIGN.md — never bench.py's `value`.
  - VectorEnv.Step(actions[N]) -> (obs[N,4], reward[N], done[N]) at N = 2^20 (H2D 4 MiB + D2H 21 MiB per step)
  - the reference's own loop shape on the 1-lane Env facade (README.md:34-47): 1000 x Reset-if-done else Step(i%2)
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

pkg = ge.load_package()
out = {}
for lg in (10, 16, 20):
    n = 1 << lg
    rng = np.random.default_rng(0)
    acts = rng.integers(0, 2, (8, n)).astype(np.int32)
    with pkg.VectorEnv("CartPole-v1", n, seed=1, auto_reset=True) as env:
        env.Reset()
        for t in range(5):
            env.Step(acts[t % 8])
        reps = 200 if lg < 20 else 30
        t0 = time.perf_counter()
        for t in range(reps):
            env.Step(acts[t % 8])
        dt = (time.perf_counter() - t0) / reps
    out[f"host_step_2^{lg}"] = {"ms_per_step": dt * 1e3, "env_steps_per_sec": n / dt}
cp = pkg.CartPoleEnv(seed=1)
done, t0 = True, time.perf_counter()
for i in range(1000):
    if done:
        cp.Reset(); done = False
    else:
        _, _, done, _ = cp.Step(i % 2)
dt = time.perf_counter() - t0
cp.Close()
out["single_instance_reference_loop_1000_iters"] = {"seconds": dt, "iters_per_sec": 1000 / dt}
print(json.dumps(out))
