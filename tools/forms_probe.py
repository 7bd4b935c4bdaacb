"""Times the reset forms of the CartPole one-step kernel at 2^20 lanes (bench workload) — GYMNET_RESET_FORM 0 (per-thread drain
loop) / 1 (wave-compacted reset through LDS) — and the fused rollout kernel, and checks that everything agrees bitwise (forms
with each other, fused with stepwise).  The fused kernel's code-shape variants of round 3 (GYMNET_ROLLOUT_FORM 0..3: bit 0 = the
sub-lane loop written out, bit 1 = wave-compacted reset) were measured with this script at commit "Full-size (2^20) auto-reset
parity tests ..." and then reduced to the winning shape; the numbers are in profiles/forms_probe_r03.txt.
Usage: python tools/forms_probe.py [env] [n]"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

pkg = ge.load_package()
name = sys.argv[1] if len(sys.argv) > 1 else "CartPole-v1"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(dev)
torch.cuda.set_stream(st)
ring, T = 64, 2048
adt = torch.float32 if name == "Pendulum-v1" else torch.int32
acts = torch.empty((ring, n), dtype=adt, device=dev)
out = {"env": name, "n": n}
ref = {}


def timed(fn, reps=5):
    best = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st); fn(); e1.record(st); torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) * 1e3 / T)
    best.sort()
    return best[len(best) // 2], best[0]


for kind, var, forms in (("step", "GYMNET_RESET_FORM", (0, 1, 0, 1)), ("fused", "GYMNET_ROLLOUT_FORM", (0, 0))):
    for f in forms:
        os.environ[var] = str(f)
        env = pkg.VectorEnv(name, n, seed=1, auto_reset=True, stream=st.cuda_stream)
        for t in range(ring):
            env.SampleActionsDevice(acts[t], seed=2, tick=t)
        env.ResetDevice()
        run = (lambda: env.RolloutDevice(acts, T, n, ring)) if kind == "step" else (lambda: env.RolloutFusedDevice(acts, T, n, ring))
        run(); torch.cuda.synchronize()
        state = env.GetState()                     # after exactly T steps from the reset: comparable across forms
        key = kind
        if key in ref:
            assert np.array_equal(ref[key], state), (kind, f)
        else:
            ref[key] = state
        med, lo = timed(run)
        out.setdefault(kind, {}).setdefault(str(f), []).append({"us_per_step_median": round(med, 3), "min": round(lo, 3)})
        env.Close()
        del os.environ[var]
assert np.array_equal(ref["step"], ref["fused"])
out["bitwise"] = "all forms agree; fused == stepwise"
print(json.dumps(out))
