#!/usr/bin/env python3
"""Round 5 probe (runs ON THE GPU BOX): where the time of the fused rollout with bookkeeping goes, at 2^20 CartPole lanes, 256 steps
per launch.  us per vector step for: the lean kernel with ring / sampled actions; a bookkeeping handle (EPISODE_STATS, time limit 500)
without and with the per-rollout episode records; the same with GYMNET_FLAG_COMPACT_RECORDS_ONLY (no dense last-finished-episode views)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
n, ring, seed = 1 << 20, 256, 0x5EED
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(dev)
torch.cuda.set_stream(stream)


def timed(fn, launches=8, reps=3):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for _ in range(launches):
            fn()
        b.record(stream)
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / (launches * ring))
    return sorted(ts)[len(ts) // 2]


acts = torch.empty((ring, n), dtype=torch.int32, device=dev)
with pkg.VectorEnv("CartPole-v1", n, seed=seed, auto_reset=True, stream=stream.cuda_stream) as e:
    for t in range(ring):
        e.SampleActionsDevice(acts[t], seed=seed + 1, tick=t)
    e.ResetDevice()
    print(f"lean, ring actions                         {timed(lambda: e.RolloutFusedDevice(acts, ring, n, ring)):7.3f} us/step")
    print(f"lean, sampled actions                      {timed(lambda: e.RolloutFusedDevice(None, ring, actions='sample', action_seed=7)):7.3f} us/step")
cap = n * ring // 8
bufs = {"step": torch.empty(cap, dtype=torch.int32, device=dev), "lane": torch.empty(cap, dtype=torch.int32, device=dev),
        "ret": torch.empty(cap, dtype=torch.float32, device=dev), "length": torch.empty(cap, dtype=torch.int32, device=dev),
        "capacity": cap, "count": torch.zeros(2, dtype=torch.uint32, device=dev)}
for label, kw in (("EPISODE_STATS", dict(episode_stats=True, max_episode_steps=500)),
                  ("EPISODE_STATS + DONE_LIST + COMPACT_RECORDS_ONLY", dict(episode_stats=True, max_episode_steps=500, done_list=True, compact_records_only=True)),
                  ("EPISODE_STATS + DONE_LIST + FINAL_OBS", dict(episode_stats=True, max_episode_steps=500, done_list=True, final_obs=True))):
    with pkg.VectorEnv("CartPole-v1", n, seed=seed, auto_reset=True, stream=stream.cuda_stream, **kw) as e:
        e.ResetDevice()
        print(f"{label}:")
        print(f"   ring actions, no episode records        {timed(lambda: e.RolloutFusedDevice(acts, ring, n, ring)):7.3f} us/step")
        print(f"   sampled actions, no episode records     {timed(lambda: e.RolloutFusedDevice(None, ring, actions='sample', action_seed=7)):7.3f} us/step")
        print(f"   sampled actions + episode records       {timed(lambda: e.RolloutFusedDevice(None, ring, actions='sample', action_seed=7, episodes=bufs)):7.3f} us/step"
              f"   (records kept / episodes ended: {bufs['count'].cpu().numpy().tolist()})")
        print(f"   ... the same, GYMNET_RECORDS_NO_OVERFLOW  {timed(lambda: e.RolloutFusedDevice(None, ring, actions='sample', action_seed=7, episodes=dict(bufs, no_overflow=True))):7.3f} us/step")
        print(f"   one launch per step (stepwise EXTRAS)   {timed(lambda: e.RolloutDevice(acts, ring, n, ring), launches=4):7.3f} us/step")
