"""Times the bookkeeping (EXTRAS) variants of the step kernel at 2^20 CartPole lanes (HIP events over T back-to-back launches).
   python tools/extras_probe.py            all variants, one JSON object
   python tools/extras_probe.py all 200    ONE variant, 200 launches (the shape the rocprofv3 passes of tools/gpu_extras_r03.sh use)"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(dev); torch.cuda.set_stream(st)
n, ring = 1 << 20, 64
VARIANTS = {"lean": {}, "done_list": dict(done_list=True), "episode_stats": dict(episode_stats=True),
            "done_list+stats": dict(done_list=True, episode_stats=True),
            "all": dict(done_list=True, episode_stats=True, final_obs=True),
            "stats+final_obs(dense)": dict(episode_stats=True, final_obs=True),
            "no_autoreset(sbd)": dict(auto_reset=False)}
only = sys.argv[1] if len(sys.argv) > 1 else None
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
out = {}
for label, kw in VARIANTS.items():
    if only and label != only:
        continue
    a = dict(auto_reset=True); a.update(kw)
    env = pkg.VectorEnv("CartPole-v1", n, seed=1, stream=st.cuda_stream, **a)
    acts = torch.empty((ring, n), dtype=torch.int32, device=dev)
    for t in range(ring):
        env.SampleActionsDevice(acts[t], seed=2, tick=t)
    env.ResetDevice(); env.RolloutDevice(acts, 128, n, ring); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st); env.RolloutDevice(acts, T, n, ring); e1.record(st); torch.cuda.synchronize()
    out[label] = {"us_per_step": round(e0.elapsed_time(e1) * 1e3 / T, 3), "kernel": env.KernelName()}
    env.Close()
print(json.dumps(out))
if only:
    sys.exit(0)

# the reference-faithful loop on the device: Step (no auto-reset) followed by the caller's `if (done) Reset()`
env = pkg.VectorEnv("CartPole-v1", n, seed=1, stream=st.cuda_stream, auto_reset=False)
acts = torch.empty((ring, n), dtype=torch.int32, device=dev)
for t in range(ring):
    env.SampleActionsDevice(acts[t], seed=2, tick=t)
env.ResetDevice()
for t in range(64):
    env.StepDevice(acts[t % ring]); env.ResetWhereDevice()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(st)
for t in range(T):
    env.StepDevice(acts[t % ring]); env.ResetWhereDevice()
e1.record(st); torch.cuda.synchronize()
print(json.dumps({"step + reset_where (reference loop shape), us per iteration": e0.elapsed_time(e1) * 1e3 / T}))
env.Close()

# closed loop on the device: a (random) policy kernel produces the actions of step t from nothing but the tick,
# then the env steps — two launches per step, no host hop
env = pkg.VectorEnv("CartPole-v1", n, seed=1, stream=st.cuda_stream, auto_reset=True)
a = torch.empty(n, dtype=torch.int32, device=dev); torch.cuda.synchronize()
env.ResetDevice()
for t in range(64):
    env.SampleActionsDevice(a, seed=3, tick=t); env.StepDevice(a)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(st)
for t in range(T):
    env.SampleActionsDevice(a, seed=3, tick=t); env.StepDevice(a)
e1.record(st); torch.cuda.synchronize()
print(json.dumps({"closed loop: sample_actions kernel + step kernel, us per step": e0.elapsed_time(e1) * 1e3 / T}))
env.Close()
