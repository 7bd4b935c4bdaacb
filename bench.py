#!/usr/bin/env python3
"""bench.py — env-steps/sec of the batched CartPole hot path on N MI355X (BASELINE.json's metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is ONE vector step = ONE launch of the step kernel over this GPU's whole batch
(2^20 CartPole-v1 lanes, float32 structure-of-arrays state, fused auto-reset, iid random {0,1}
actions pre-generated on the device by the engine's Philox sampler: BASELINE.json configs[1]).
Weak scaling: every GPU owns 2^20 lanes of one global batch of N * 2^20 lanes (configs[4] at N = 8);
lanes are independent, so the data path has no collective (`--allgather` adds the per-step RCCL
observation all-gather north_star mentions, for measuring what it costs).

The timed region is K steps, one kernel launch per step (gymnet_vecenv_rollout_device: back-to-back stream
launches at this size, hipGraph replay for batches that are launch-bound), bracketed by barrier + synchronize;
inputs are resident in HBM before it starts.  Prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0     # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 achievable)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=4096)
    p.add_argument("--warmup", type=int, default=512)
    p.add_argument("--env", default="CartPole-v1")
    p.add_argument("--num-envs", type=int, default=1 << 20, help="lanes per GPU")
    p.add_argument("--ring", type=int, default=256, help="distinct pre-generated action slices (ring * num_envs * 4 B)")
    p.add_argument("--allgather", action="store_true", help="all-gather observations over RCCL after every step")
    p.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="target duration of the CPU baseline sample")
    p.add_argument("--force-dist", action="store_true", help="initialise torch.distributed (RCCL) even at world size 1 "
                   "(exercises the collective code path on a 1-GPU box)")
    return p.parse_args()


def cpu_baseline(num_envs, target_seconds):
    """The oracle's per-instance float64 path (oracle/cpu_baseline.c, kind = "port": the reference's C#
    cannot run here) timed on all host cores over a bounded sample of the same workload."""
    from oracle import capi as oracle
    oracle.build()
    cores = os.cpu_count() or 1
    probe = oracle.cpu_baseline(num_envs, 4, cores, alloc_faithful=True)
    rate = probe["steps_per_sec"]
    t_steps = int(max(8, min(4096, target_seconds * rate / num_envs)))
    r = oracle.cpu_baseline(num_envs, t_steps, cores, alloc_faithful=True)
    r0 = oracle.cpu_baseline(num_envs, max(8, t_steps // 4), cores, alloc_faithful=False)   # variant (B): no heap traffic
    r1 = oracle.cpu_baseline(1, 100_000, 1, alloc_faithful=True)     # BASELINE.json configs[0]
    return {"value": r["steps_per_sec"], "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{num_envs} per-instance float64 CartPole envs x {t_steps} steps, reset-on-done, "
                      f"split over {cores} threads, 2 heap allocations per step like the C# path "
                      f"({r['seconds']:.1f} s)",
            "no_alloc_variant_value": r0["steps_per_sec"],
            "single_instance_100k_steps_per_sec": r1["steps_per_sec"]}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    pkg = ge.load_package()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world == 1 and args.gpus > 1:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    if pkg.device_count() < 1:
        raise SystemExit("bench.py needs an AMD GPU: the engine has no CPU fallback")
    # one process per GPU; GYMNET_BENCH_BACKEND=gloo lets several ranks share the GPUs that exist (a 1-GPU box can then
    # exercise the N > 1 plumbing: lane offsets, shard buffers, barrier, max-over-ranks) — RCCL needs one GPU per rank
    backend = os.environ.get("GYMNET_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    red_dev = dev if backend == "nccl" else torch.device("cpu")      # where the timing reduction tensors live
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    n = args.num_envs
    ring = max(2, args.ring + (args.ring % 2))
    K, W = args.steps, args.warmup
    seed = 0x5EED
    # a non-default stream: the engine orders all of its work on it, and torch.cuda.Event timing below
    # records on the same stream (the null stream would make the library create a private one)
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)

    # this rank's shard of the global batch; observations live inside the (optional) gather buffer
    env = pkg.ShardedVectorEnv(args.env, n * world, rank=rank, world_size=world, device=dev_index, seed=seed,
                               auto_reset=True, gather_obs=use_dist, tensor_device=dev,
                               force_gather=args.force_dist)
    local = env.local
    adtype = torch.float32 if local._adtype.__name__ == "float32" else torch.int32
    actions = torch.empty((ring, n), dtype=adtype, device=dev)
    for t in range(ring):      # ActionSpace.Sample() per lane per step, on the device (Philox key = seed + 1)
        local.SampleActionsDevice(actions[t].data_ptr(), seed=seed + 1, tick=t)
    env.ResetDevice()
    env.Sync()

    def run(steps):
        if args.allgather and use_dist:
            for t in range(steps):
                env.StepDevice(actions[t % ring].data_ptr())
                env.AllGatherObs()
        elif args.no_graph:
            for t in range(steps):
                env.StepDevice(actions[t % ring].data_ptr())
        else:
            local.RolloutDevice(actions.data_ptr(), steps, n, ring)

    def barrier():
        if use_dist:
            dist.barrier()

    run(W)
    torch.cuda.synchronize(dev)
    barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    e0.record(stream)
    run(K)
    e1.record(stream)
    torch.cuda.synchronize(dev)
    barrier()
    t1 = time.perf_counter()
    wall = t1 - t0
    ev_ms = e0.elapsed_time(e1)
    if use_dist:
        tw = torch.tensor([wall, ev_ms], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        wall, ev_ms = float(tw[0]), float(tw[1])

    # sanity: the engine really ran K + W steps on every lane
    c = local.Counters()
    assert c["lane_steps"] == (K + W) * n, c

    # Cross-check of the per-launch figure: 200 single launches, each bracketed by its own HIP-event pair on the
    # engine's stream (isolated launches: no back-to-back overlap with a neighbour's ramp / drain).
    single_us = None
    if not args.allgather:
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(200)]
        for t, (a0, a1) in enumerate(evs):
            a0.record(stream)
            local.StepDevice(actions[t % ring].data_ptr())
            a1.record(stream)
        torch.cuda.synchronize(dev)
        ds = sorted(a0.elapsed_time(a1) * 1e3 for a0, a1 in evs)
        single_us = ds[len(ds) // 2]
        K_extra = len(evs)
    else:
        K_extra = 0

    # Secondary figure, NOT the headline: the same K steps fused into one launch per `ring` steps (state stays in
    # registers, gymnet_vecenv_rollout_fused_device) — open-loop rollouts only, so it is reported beside, not as, `value`.
    fused = None
    if not args.allgather:
        fsteps = max(ring, (min(K, 2048) // ring) * ring)
        local.RolloutFusedDevice(actions.data_ptr(), ring, n, ring)
        torch.cuda.synchronize(dev)
        f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        f0.record(stream)
        for _ in range(fsteps // ring):
            local.RolloutFusedDevice(actions.data_ptr(), ring, n, ring)
        f1.record(stream)
        torch.cuda.synchronize(dev)
        fused_us = f0.elapsed_time(f1) * 1e3 / fsteps
        fused = {"env_steps_per_sec_per_gpu": n / (fused_us * 1e-6), "us_per_step": fused_us, "steps_per_launch": ring,
                 "note": "T-step fused kernel, no per-step observation hand-off; not comparable to `value`"}

    # Secondary figure for N > 1, NOT the headline: the same stepping with the RCCL all-gather of observations
    # north_star mentions after EVERY step (in place, rank-major [G][D][N/G] buffer).  The stepping path itself needs no
    # collective; this shows what a consumer that wants every rank to see all observations pays over xGMI.
    gathered = None
    if use_dist and not args.allgather and backend == "nccl":
        try:
            gs = 128
            for t in range(16):
                env.StepDevice(actions[t % ring].data_ptr()); env.AllGatherObs()
            torch.cuda.synchronize(dev); barrier()
            g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            tg0 = time.perf_counter()
            g0.record(stream)
            for t in range(gs):
                env.StepDevice(actions[t % ring].data_ptr()); env.AllGatherObs()
            g1.record(stream)
            torch.cuda.synchronize(dev); barrier()
            gwall = time.perf_counter() - tg0
            tg = torch.tensor([gwall], dtype=torch.float64, device=red_dev)
            dist.all_reduce(tg, op=dist.ReduceOp.MAX)
            ok = bool(torch.isfinite(env.GlobalObs()).all()) and all(float(env.GlobalObs()[r].abs().sum()) > 0 for r in range(world))
            gathered = {"value": n * world * gs / float(tg[0]), "unit": "env-steps/s", "ms_per_step": float(tg[0]) * 1e3 / gs,
                        "steps": gs, "allgather_bytes_per_rank_per_step": env.obs_dim * n * 4, "gathered_obs_finite_and_nonzero": ok}
        except Exception as e:                      # never lose the headline over the optional collective
            gathered = {"error": repr(e)[:200]}

    # Attainable copy bandwidth on THIS box (read + write bytes / time of a device-to-device float copy), reported
    # beside the 8 TB/s spec peak the roofline fraction uses: cache-resident (32 MiB) and HBM-resident (2 GiB).
    copy_bw = None
    if rank == 0 and world == 1:
        copy_bw = {}
        for label, mib in (("32MiB", 32), ("2GiB", 2048)):
            src = torch.empty(mib * (1 << 18), dtype=torch.float32, device=dev).normal_()
            dst = torch.empty_like(src)
            for _ in range(3):
                dst.copy_(src)
            torch.cuda.synchronize(dev)
            c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 50 if mib <= 64 else 10
            c0.record(stream)
            for _ in range(reps):
                dst.copy_(src)
            c1.record(stream)
            torch.cuda.synchronize(dev)
            copy_bw[label] = 2 * src.numel() * 4 / (c0.elapsed_time(c1) * 1e-3 / reps) / 1e9
            del src, dst

    if rank == 0:
        bytes_per_step = local.AlgorithmicBytesPerStep                      # CartPole: 41 B
        launch_us = ev_ms * 1e3 / K                                          # HIP events over the timed region / launches
        achieved = bytes_per_step * n / (launch_us * 1e-6) / 1e9             # GB/s per GPU, algorithmic bytes
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")               # rocprofv3 --pmc result, per launch
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(args.env, {}).get(str(n))
            except Exception:
                traffic = None
        out = {
            "metric": "env-steps/sec", "value": n * world * K / wall, "unit": "env-steps/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": wall * 1e3 / K,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.env} batched, batch={n} lanes per GPU (global {n * world}), float32 SoA state, "
                                   "fused auto-reset, iid random actions pre-generated in HBM",
                       "num_envs_per_gpu": n, "global_num_envs": n * world, "action_ring": ring,
                       "launch": ("one kernel launch per step, eager (python loop)" if (args.no_graph or args.allgather) else
                                  "one kernel launch per step; gymnet_vecenv_rollout_device: " +
                                  ("hipGraph replay" if n * local.AlgorithmicBytesPerStep < (24 << 20) else "back-to-back stream launches")),
                       "allgather_obs": bool(args.allgather and use_dist), "parallelism": f"lane-sharded x{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel": "step_kernel<CartPole,4,autoreset>" if args.env == "CartPole-v1" else "step_kernel",
                         "algorithmic_bytes_per_launch": bytes_per_step * n, "launch_us": launch_us,
                         "isolated_launch_us_median": single_us, "measured_copy_GBps": copy_bw},
        }
        if fused:
            out["fused_rollout"] = fused
        if gathered:
            out["with_obs_allgather"] = gathered
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(n, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    env.Close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
