#!/usr/bin/env python3
"""bench.py — env-steps/sec of the batched CartPole hot path on N MI355X (BASELINE.json's metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Plain `python bench.py --gpus N` (no WORLD_SIZE in the environment) starts the N rank processes itself: the
parent spawns N fresh children before it makes any GPU call, forwards rank 0's JSON line and exits non-zero
if any child fails.  When the box has fewer than N GPUs the children share the GPUs that exist over gloo
(`config.backend` says so; that exercises the N > 1 plumbing, it is NOT a multi-GPU measurement).

A "step" is ONE vector step = ONE launch of the step kernel over this GPU's whole batch
(2^20 CartPole-v1 lanes, float32 structure-of-arrays state, fused auto-reset, iid random {0,1}
actions pre-generated on the device by the engine's Philox sampler: BASELINE.json configs[1]).
Weak scaling: every GPU owns 2^20 lanes of one global batch of N * 2^20 lanes (configs[4] at N = 8);
lanes are independent, so the data path has no collective.  `value` is ALWAYS the step-only rate; the
per-step RCCL observation all-gather north_star mentions is measured beside it at N > 1
(`with_obs_allgather`: on the critical path, and overlapped with the next step through double-buffered
observation arrays).

Timing.  One timed region = EXACTLY K steps, one kernel launch per step, bracketed by barrier +
synchronize on both sides, MAX over ranks; inputs are resident in HBM before it starts.  A K-step region
of 7 us kernels can be as short as 0.15 ms, where host launch ramp and synchronize latency are a fifth of
the time, so the region is REPEATED (same K, fresh bracket each time) until >= 50 ms have been timed and the
MEDIAN repeat is reported: ms_per_step = median region wall / K, `repeats` says how many regions ran.
Prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0     # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 achievable)
MIN_TIMED_SECONDS = 0.05   # repeat the K-step region until this much has been timed
MAX_REPEATS = 2000
# Exit status of every rank when a SECONDARY section (the optional observation all-gather, the teardown) hung and the watchdog
# fired: NON-ZERO.  A GPU process whose collective hung must not leave with status 0.  The headline has been measured
# by then and rank 0 prints it FIRST — with `"watchdog_fired": <section>` and the section's error record — so the measurement is
# in the log either way; spawn_ranks() propagates the status.  (GYMNET_BENCH_WATCHDOG_RC overrides the value.)
WATCHDOG_EXIT = int(os.environ.get("GYMNET_BENCH_WATCHDOG_RC", "3"))
# xGMI: 7 links per GPU, ~153 GB/s each (/opt/skills/guides/MI355X_MICROARCH.md) — the SURVEY §8(e) all-gather model
XGMI_LINK_GBPS = 153.0


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=4096)
    p.add_argument("--warmup", type=int, default=512)
    p.add_argument("--env", default="CartPole-v1")
    p.add_argument("--num-envs", type=int, default=1 << 20, help="lanes per GPU")
    p.add_argument("--ring", type=int, default=256, help="distinct pre-generated action slices (ring * num_envs * 4 B)")
    p.add_argument("--allgather", action="store_true", help="put the per-step RCCL observation all-gather INSIDE the timed "
                   "region (then `value` is no longer the headline and config.workload says so)")
    p.add_argument("--no-graph", action="store_true", help="eager launches from a python loop instead of gymnet_vecenv_rollout_device")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-extras", action="store_true", help="skip the secondary figures (fused rollout, 2^27 lanes, copy probe)")
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="target duration of the CPU baseline sample")
    p.add_argument("--force-dist", action="store_true", help="initialise torch.distributed (RCCL) even at world size 1 "
                   "(exercises the collective code path on a 1-GPU box)")
    p.add_argument("--min-seconds", type=float, default=MIN_TIMED_SECONDS)
    p.add_argument("--no-overlap", action="store_true", help="N > 1: single observation buffer (no gather/step overlap)")
    p.add_argument("--no-group-leg", action="store_true", help="N > 1: skip the single-process gymnet_group_* leg")
    p.add_argument("--no-host-boundary", action="store_true", help="skip the NDArray-shaped host path figure (gymnet_vecenv_step)")
    p.add_argument("--group-child", type=int, default=0, help=argparse.SUPPRESS)   # internal: run the gymnet_group_* leg over this many members
    p.add_argument("--rollout-child", default="", help=argparse.SUPPRESS)   # internal: run the fused-rollout variants (comma list or "all") for a profiler
    p.add_argument("--no-rollout-pmc", action="store_true", help="do not measure the fused rollouts' VALU instructions per env-step with a "
                   "rocprofv3 --pmc child pass (the constants from profiles/rollout_valu.json are reported instead, labelled)")
    p.add_argument("--no-traffic", action="store_true", help="do not measure roofline.traffic with rocprofv3 --pmc child passes "
                   "(the constant from profiles/traffic.json is reported instead, labelled)")
    p.add_argument("--policy", default="", help="launch policy overrides for the headline handle, e.g. vec=4,nt=12,block=128 "
                   "(gymnet_vecenv_set_launch_policy; every configuration computes the same bits)")
    p.add_argument("--dtype", default="f32", choices=["f32", "f64"], help="f64: CartPole in the reference's own float64 arithmetic "
                   "(GYMNET_FLAG_F64, 73 B per env-step); the default line is always f32")
    return p.parse_args()


def cpu_baseline(num_envs, target_seconds):
    """The oracle's per-instance float64 path (oracle/cpu_baseline.c, kind = "port": the reference's C#
    cannot run here) timed on all host cores over a bounded sample of the same workload."""
    from oracle import capi as oracle
    oracle.build()
    hw, cores, why = usable_cpus()
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
            "host": f"{hw} hardware threads; {cores} usable by this process ({why})",
            "no_alloc_variant_value": r0["steps_per_sec"],
            "single_instance_100k_steps_per_sec": r1["steps_per_sec"]}


def usable_cpus():
    """(hardware threads, CPUs this process may actually use, what limits them): the affinity mask and the cgroup CPU
    quota both count — a 256-thread host under a 16-CPU quota runs 16 threads' worth of work however many are started."""
    hw = os.cpu_count() or 1
    n, why = hw, "no limit below the hardware"
    try:
        a = len(os.sched_getaffinity(0))
        if a < n:
            n, why = a, "affinity mask"
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1"):
                q = max(1, int(float(quota) / period))
                if q < n:
                    n, why = q, f"cgroup CPU quota {quota}/{int(period)} us"
            break
        except (OSError, ValueError, IndexError):
            continue
    return hw, n, why


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes.  Nothing in this
    function (or before it) touches the GPU; the children are started with subprocess (never exec from a process
    that has initialised the GPU)."""
    import torch                                        # device_count() does not initialise the GPU on this image
    ngpu = torch.cuda.device_count()
    env = dict(os.environ)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env["MASTER_PORT"] = str(_free_port())
    env["WORLD_SIZE"] = str(args.gpus)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if ngpu < 1:
        raise SystemExit("bench.py needs an AMD GPU: the engine has no CPU fallback")
    if ngpu < args.gpus:
        env.setdefault("GYMNET_BENCH_BACKEND", "gloo")   # ranks share the GPUs that exist; labelled in config.backend
    procs = []
    for r in range(args.gpus):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e))
    import signal

    def stop_children(signum, frame):                    # the launcher is being stopped: take the ranks with it
        for q in procs:
            if q.poll() is None:
                q.terminate()
        raise SystemExit(128 + signum)
    for sig in (signal.SIGTERM, signal.SIGINT):
        signal.signal(sig, stop_children)
    rc = 0
    pending = set(range(args.gpus))
    while pending:
        for r in list(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                for q in pending:                        # one rank died: the others would wait in a barrier forever
                    procs[q].terminate()
        time.sleep(0.05)
    raise SystemExit(rc)


class NodeBarrier:
    """Barrier for the ranks of ONE node (the contract runs N GPUs of one node): every rank owns one cache line of a
    /dev/shm file, bumps its own sequence number and spins until every line has caught up.  A few microseconds, against
    the tens of microseconds of a collective-based barrier — which matters because the closing barrier sits INSIDE the timed
    bracket and a 20-step region is only ~160 us long.  Single-writer slots on a TSO machine: no atomics needed."""

    def __init__(self, rank, world, key):
        import mmap
        import numpy as np
        self.rank, self.world, self.seq = rank, world, 0
        self.path = f"/dev/shm/gymnet_bench_{os.getuid()}_{key}"
        size = 64 * world
        if rank == 0:
            with open(self.path, "wb") as f:
                f.write(b"\0" * size)
        self._np = np

    def attach(self):
        import mmap
        self._f = open(self.path, "r+b")
        self._mm = mmap.mmap(self._f.fileno(), 64 * self.world)
        self.slots = self._np.frombuffer(self._mm, dtype=self._np.int64)[::8]       # one int64 per 64-byte line

    def __call__(self, timeout=120.0):
        self.seq += 1
        self.slots[self.rank] = self.seq
        t_end = None
        spins = 0
        while int(self.slots.min()) < self.seq:
            spins += 1
            if spins & 0xFFF == 0:
                now = time.perf_counter()
                t_end = t_end or now + timeout
                if now > t_end:
                    raise RuntimeError("node barrier timed out (a rank died?)")

    def close(self):
        try:
            del self.slots
            self._mm.close()
            self._f.close()
            if self.rank == 0:
                os.unlink(self.path)
        except Exception:
            pass


def make_watchdog(rank, out, emitted, section, seconds, exit_fn=os._exit):
    """Watchdog for a secondary section that could hang rather than fail (a collective waiting for a peer): after `seconds` rank 0
    prints the line it already has — headline first, the section marked as timed out — and every rank leaves with
    WATCHDOG_EXIT (non-zero).  Never re-execs."""
    def fire():
        if emitted.acquire(blocking=False):
            import faulthandler
            sys.stderr.write(f"[bench rank {rank}] section {section!r} timed out after {seconds} s; stacks:\n")
            faulthandler.dump_traceback(file=sys.stderr, all_threads=True)    # where each rank was stuck
            if rank == 0:
                out[section] = {"error": f"timed out after {seconds} s; headline unaffected"}
                out["watchdog_fired"] = section
                print(json.dumps(out), flush=True)
            exit_fn(WATCHDOG_EXIT)
    t = threading.Timer(seconds, fire)
    t.daemon = True
    t.start()
    return t


# bytes a step WRITES per env-step (new state rows the kernel owns + observation rows + reward + done): the write path is what binds
WRITTEN_BYTES = {("CartPole-v1", False): 21, ("CartPole-v1", True): 37, ("Pendulum-v1", False): 21, ("MountainCar-v0", False): 13,
                 ("Acrobot-v1", False): 37}


def allgather_model(world, bytes_per_rank):
    """what a per-step observation all-gather costs on point-to-point xGMI — direct (every rank stores its slice to
    each peer over that peer's own link, all links concurrently) vs a ring (per-link bound, G - 1 hops)."""
    direct_us = bytes_per_rank / (XGMI_LINK_GBPS * 1e9) * 1e6
    return {"link_GBps": XGMI_LINK_GBPS, "bytes_per_rank": bytes_per_rank, "direct_us": direct_us, "ring_us": direct_us * max(0, world - 1),
            "note": "analytic: 16 MiB per rank at 2^20 float32 CartPole lanes (32 MiB in float64) -> ~110 us direct, ~770 us ring at 8 GPUs (x2 in float64)"}


def measure_traffic(args, wide16=True, timeout=90):
    """roofline.traffic measured IN THIS RUN: HBM-side bytes per launch of the step kernel from the PMC counters, collected as
    /opt/skills/guides/MI355X_MICROARCH.md prescribes — FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (no trace
    flags beside --pmc), bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (both count KiB; on gfx950 FETCH_SIZE reports half the bytes
    of a wide coalesced streaming read).  Each pass is a fresh CHILD process (`rocprofv3 ... -- python3 bench.py ...`: the
    program after `--` is python3 itself, started with subprocess — never an exec from this GPU process) running 10 + 3 x 100
    eager launches of the same kernel on the same batch.  Returns (bytes per launch, description); raises on any failure — the
    caller then falls back to the committed constant and says so."""
    import shutil
    import sqlite3
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        raise RuntimeError("rocprofv3 not found")
    vals = {}
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        out_dir = tempfile.mkdtemp(prefix=f"gymnet_pmc_{counter}_", dir="/tmp")
        try:
            cmd = [exe, "--pmc", counter, "-d", out_dir, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__),
                   "--no-cpu-baseline", "--no-extras", "--no-traffic", "--no-graph", "--steps", "100", "--warmup", "10", "--min-seconds", "0",
                   "--env", args.env, "--num-envs", str(args.num_envs), "--dtype", getattr(args, "dtype", "f32")]
            if args.policy:
                cmd += ["--policy", args.policy]
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout)
            db = None
            for dirpath, _, files in os.walk(out_dir):
                for f in files:
                    if f.endswith("_results.db"):
                        db = os.path.join(dirpath, f)
            if r.returncode != 0 or db is None:
                raise RuntimeError(f"rocprofv3 --pmc {counter}: rc {r.returncode}, {(r.stderr or r.stdout)[-200:]}")
            c = sqlite3.connect(db)
            row = c.execute("select avg(value), count(*) from counters_collection where kernel_name like '%step_kernel%' and counter_name = ?",
                            (counter,)).fetchone()
            c.close()
            if not row or row[0] is None or row[1] < 50:
                raise RuntimeError(f"rocprofv3 --pmc {counter}: no step-kernel dispatches in the database")
            vals[counter] = (row[0], row[1])
        finally:
            shutil.rmtree(out_dir, ignore_errors=True)
    if wide16:
        traffic = (2.0 * vals["FETCH_SIZE"][0] + vals["WRITE_SIZE"][0]) * 1024.0
        how = "(2 x FETCH_SIZE + WRITE_SIZE) x 1024 B per launch (the guide's gfx950 correction for 16-byte-per-lane coalesced reads)"
    else:
        # the guide calibrates the 2x read factor for 16 B / lane accesses only; this kernel form reads 4 (8) B per lane: raw counters
        traffic = (vals["FETCH_SIZE"][0] + vals["WRITE_SIZE"][0]) * 1024.0
        how = ("UNCALIBRATED: (FETCH_SIZE + WRITE_SIZE) x 1024 B per launch, raw counters — the kernel form in use does not make 16-byte-per-lane "
               "reads, for which alone the guide gives the 2x FETCH_SIZE correction; the read side may be under-counted by up to 2x")
    return traffic, (f"measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in two separate child passes "
                     f"({vals['FETCH_SIZE'][1]} / {vals['WRITE_SIZE'][1]} step-kernel dispatches), {how}")


# ---- fused rollout: the VALU-issue roofline --------------------------------------------------------------------
# The fused rollouts keep the state in registers and move 0-4 bytes per env-step: they are bound by VALU instruction ISSUE, not by
# memory (profiles/pmc_rollout_r05.txt).  A SIMD issues one wave-instruction of 64 lanes per 4 clocks (16 lanes per clock), so
#     issue_floor_us = lanes x VALU instructions per env-step / (16 lanes x SIMDs) / clock
# where "VALU per env-step" = SQ_INSTS_VALU / SQ_WAVES / lanes per thread / steps per launch (an instruction of a thread that serves
# four lanes counts a quarter for each).  Two instruction classes hold the pipe longer than 4 clocks — v_mad_u64_u32 ~5, the
# transcendentals ~9, float64 arithmetic ~5.3 (tools/issue_rate_probe.hip) — and `frac_measured_rates` prices them so; the plain `frac` does not.
ROLLOUT_VARIANTS = ("f32_ring", "f32_sampled", "f32_epsilon_greedy", "f64_ring", "f64_sampled", "f64_epsilon_greedy")
ROLLOUT_CHILD_STEPS, ROLLOUT_CHILD_LAUNCHES = 64, 3
ENGINE_CLOCK_GHZ = 2.4      # MI355X peak engine clock (rocminfo "Max Clock Freq": 2400 MHz); the floor is priced at the peak


def rollout_variant_call(env, variant, acts, n, ring, steps, seed):
    kind = variant.split("_", 1)[1]
    if kind == "ring":
        return lambda: env.RolloutFusedDevice(acts.data_ptr(), steps, n, ring)
    if kind == "sampled":
        return lambda: env.RolloutFusedDevice(None, steps, actions="sample", action_seed=seed + 1, action_tick0=0)
    return lambda: env.RolloutFusedDevice(acts.data_ptr(), steps, n, ring, actions="epsilon_greedy", action_seed=seed + 1, action_tick0=0, epsilon=0.1)


def rollout_child(args):
    """`bench.py --rollout-child all|v1,v2`: each variant's fused rollout, one warm-up launch + ROLLOUT_CHILD_LAUNCHES launches of
    ROLLOUT_CHILD_STEPS steps, in the order given — the program a profiler wraps (rocprofv3 ... -- python3 bench.py --rollout-child ...).
    Prints one JSON line: the sequence it ran (the parent tells the dispatches of one kernel name apart by their order)."""
    import torch
    import __graft_entry__ as ge
    pkg = ge.load_package()
    n, ring, seed = args.num_envs, 8, 0x5EED
    want = ROLLOUT_VARIANTS if args.rollout_child == "all" else tuple(v for v in args.rollout_child.split(",") if v)
    dev = torch.device("cuda", 0)
    acts = torch.empty((ring, n), dtype=torch.int32, device=dev)
    seq = []
    for dt in ("f32", "f64"):
        mine = [v for v in want if v.startswith(dt)]
        if not mine:
            continue
        with pkg.VectorEnv(args.env, n, device=0, seed=seed, auto_reset=True, dtype="float64" if dt == "f64" else "float32") as e:
            for t in range(ring):
                e.SampleActionsDevice(acts[t].data_ptr(), seed=seed + 1, tick=t)
            e.ResetDevice()
            for v in mine:
                fn = rollout_variant_call(e, v, acts, n, ring, ROLLOUT_CHILD_STEPS, seed)
                for _ in range(1 + ROLLOUT_CHILD_LAUNCHES):
                    fn()
                e.Sync()
                seq.append({"variant": v, "launches": 1 + ROLLOUT_CHILD_LAUNCHES, "steps": ROLLOUT_CHILD_STEPS})
    print(json.dumps({"rollout_child": seq, "num_envs": n}), flush=True)


def read_rollout_counters(db, seq):
    """Per variant of `seq` (the child's own account of what it ran): the counters of its rollout_kernel dispatches, summed over a
    dispatch's rows and averaged over the variant's timed launches (the first launch of every variant is its warm-up and is left out)."""
    import re
    import sqlite3
    c = sqlite3.connect(db)
    try:
        cols = [r[1] for r in c.execute("pragma table_info(counters_collection)").fetchall()]
        dur = "max(duration)" if "duration" in cols else "0"
        rows = c.execute(f"select dispatch_id, kernel_name, counter_name, sum(value), {dur} from counters_collection "
                         "where kernel_name like '%rollout_kernel%' group by dispatch_id, kernel_name, counter_name order by dispatch_id").fetchall()
    finally:
        c.close()
    disp = {}
    for did, k, cn, v, d in rows:
        disp.setdefault(did, {"kernel": k})[cn] = v
        disp[did]["_duration_ns"] = d
    order = [disp[d] for d in sorted(disp)]
    if len(order) != sum(x["launches"] for x in seq):
        raise RuntimeError(f"{len(order)} rollout_kernel dispatches in the database, the child reported {sum(x['launches'] for x in seq)}")
    out, at = {}, 0
    for x in seq:
        mine = order[at + 1:at + x["launches"]]
        at += x["launches"]
        k = mine[0]["kernel"]
        lanes_per_thread = int(re.search(r"rollout_kernel<[^,]+,\s*(\d+)", k).group(1))
        cs = {cn: sum(m[cn] for m in mine) / len(mine) for cn in mine[0] if cn != "kernel"}     # (incl. "_duration_ns": the profiled launch's length)
        out[x["variant"]] = {"kernel": k, "lanes_per_thread": lanes_per_thread, "steps_per_launch": x["steps"], "counters": cs,
                             "valu_per_env_step": cs["SQ_INSTS_VALU"] / cs["SQ_WAVES"] / lanes_per_thread / x["steps"]}
    return out


def measure_rollout_valu(args, timeout=120):
    """VALU instructions per env-step of every fused-rollout variant, measured IN THIS RUN: one rocprofv3 --pmc child pass (SQ counters
    only, no trace flags; the program after `--` is python3 itself, started with subprocess) over `bench.py --rollout-child all`."""
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        raise RuntimeError("rocprofv3 not found")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    out_dir = tempfile.mkdtemp(prefix="gymnet_pmc_rollout_", dir="/tmp")
    try:
        cmd = [exe, "--pmc", "SQ_INSTS_VALU", "SQ_WAVES", "SQ_INSTS_VALU_INT64", "SQ_INSTS_VALU_TRANS_F32", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_ADD_F64",
               "SQ_INSTS_VALU_MUL_F64", "GRBM_GUI_ACTIVE", "-d", out_dir, "-o", "pmc", "--",
               sys.executable, os.path.abspath(__file__), "--rollout-child", "all", "--env", args.env, "--num-envs", str(args.num_envs)]
        r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout)
        seq = None
        for line in r.stdout.splitlines():
            if line.startswith('{"rollout_child"'):
                seq = json.loads(line)["rollout_child"]
        db = None
        for dirpath, _, files in os.walk(out_dir):
            for f in files:
                if f.endswith("_results.db"):
                    db = os.path.join(dirpath, f)
        if r.returncode != 0 or db is None or not seq:
            raise RuntimeError(f"rocprofv3 --pmc SQ_*: rc {r.returncode}, {(r.stderr or r.stdout)[-300:]}")
        return read_rollout_counters(db, seq)
    finally:
        shutil.rmtree(out_dir, ignore_errors=True)


def valu_busy_in_pass(counters, simds):
    """VALU-pipe occupancy inside the --pmc pass itself, free of any assumed clock: every counted instruction holds its SIMD for at
    least 4 clocks, and GRBM_GUI_ACTIVE (summed over the chip's XCDs, 128 SIMDs each) is the launch's length in the chip's OWN clocks:
    busy = SQ_INSTS_VALU x 4 / simds / (GRBM_GUI_ACTIVE / XCDs).  (tools/gpu_busy_check_r06.sh: GRBM_GUI_ACTIVE / 8 / duration = 2.23 GHz
    for a 181 us rollout launch — the ramping clock tools/issue_rate_probe.hip reads from s_memtime — whereas SQ_BUSY_CYCLES / 32 shader
    engines under-counts the launch's length by ~10 % for these kernels and made the first version of this figure read 1.08-1.11.)
    Returns (busy, cycles) or (None, None)."""
    if not counters.get("SQ_INSTS_VALU"):
        return None, None
    if counters.get("GRBM_GUI_ACTIVE"):
        cycles = counters["GRBM_GUI_ACTIVE"] / max(1, simds // 128)
    elif counters.get("SQ_BUSY_CYCLES"):
        cycles = counters["SQ_BUSY_CYCLES"] / max(1, simds // 32)
    else:
        return None, None
    return counters["SQ_INSTS_VALU"] * 4.0 / simds / cycles, cycles


def valu_roofline(n, valu_per_env_step, measured_us, simds, source, quarter_rate_per_env_step=None):
    """The fused rollout's roofline object: bound by VALU issue (see the block comment above)."""
    lanes_per_clock = 16 * simds
    floor_us = n * valu_per_env_step / lanes_per_clock / (ENGINE_CLOCK_GHZ * 1e3)
    r = {"bound": "valu_issue", "valu_per_env_step": valu_per_env_step, "valu_source": source, "lanes": n, "simds": simds,
         "issue_lanes_per_clock": lanes_per_clock, "clock_GHz": ENGINE_CLOCK_GHZ, "issue_floor_us": floor_us, "measured_us": measured_us,
         "frac": floor_us / measured_us,
         "formula": "issue_floor_us = lanes x valu_per_env_step / (16 x simds) / clock_GHz / 1e3; frac = issue_floor_us / measured_us"}
    if quarter_rate_per_env_step is not None:
        # the same floor with the slower instruction classes at their MEASURED cost (tools/issue_rate_probe.hip,
        # profiles/issue_rate_r06.txt: v_mad_u64_u32 ~5 clocks per wave, transcendentals ~9, float64 arithmetic ~5.3, everything else 4)
        i64, trans, f64ops = (tuple(quarter_rate_per_env_step) + (0.0,))[:3]
        clocks = 4.0 * valu_per_env_step + 1.0 * i64 + 5.0 * trans + 1.3 * f64ops
        r.update(int64_per_env_step=i64, trans_per_env_step=trans, f64_arith_per_env_step=f64ops,
                 issue_floor_measured_rates_us=n * clocks / 64.0 / simds / (ENGINE_CLOCK_GHZ * 1e3),
                 frac_measured_rates=n * clocks / 64.0 / simds / (ENGINE_CLOCK_GHZ * 1e3) / measured_us,
                 measured_rates_note="clocks per env-step = 4 x VALU + 1 x SQ_INSTS_VALU_INT64 + 5 x SQ_INSTS_VALU_TRANS_F32 + 1.3 x SQ_INSTS_VALU_{FMA,ADD,MUL}_F64 "
                                     "(per env-step: v_mad_u64_u32 ~5 clocks, transcendentals ~9, float64 arithmetic ~5.3 — profiles/issue_rate_r06.txt); "
                                     "floor = lanes x clocks / 64 / simds / clock")
    return r


def write_roofline(n, written_bytes_per_env_step, measured_us):
    """A recording rollout writes its trajectory and reads nothing: bound by the write path, whose measured pure-write ceiling on this
    chip is 4.4-4.8 TB/s (profiles/write_path_probe_r02.txt, store_flavour_r05.txt); the fraction of the 8 TB/s spec peak beside it."""
    gbps = written_bytes_per_env_step * n / (measured_us * 1e-6) / 1e9
    return {"bound": "hbm_write", "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS,
            "written_bytes_per_env_step": written_bytes_per_env_step, "bytes_per_launch_step": written_bytes_per_env_step * n, "measured_us": measured_us,
            "measured_pure_write_GBps": [4400, 4800], "frac_of_write_ceiling": gbps / 4800.0,
            "write_ceiling_note": "4.4-4.8 TB/s is what the ONE-STEP kernels' write pattern reaches (rounds 2-5 probes); a rollout's append-only trajectory "
                                  "streams can exceed it — `frac` (of the 8 TB/s spec peak) is the figure to quote"}


def parse_policy(text):
    pol = {}
    for item in filter(None, (x.strip() for x in text.split(","))):
        k, v = item.split("=")
        pol[k.strip()] = int(v)
    return pol


def median(xs):
    s = sorted(xs)
    m = len(s) // 2
    return s[m] if len(s) % 2 else 0.5 * (s[m - 1] + s[m])


def measure_host_boundary(pkg, env_name, n, device, seed, algorithmic_bytes, steps=20):
    """ms per host-boundary step (actions in, observations / reward / done out over PCIe), median of `steps` calls."""
    import numpy as np
    out = {"num_envs": n, "unit": "ms/step", "note": "PCIe-inclusive; never `value` (inputs are NOT resident in HBM)"}
    with pkg.VectorEnv(env_name, n, device=device, seed=seed, auto_reset=True) as e:
        e.Reset()
        obs_dim = e.ObsDim
        boundary_bytes = n * (4 + 4 * obs_dim + 4 + 1)           # actions in; obs + reward + done out
        for label in ("pageable_caller_buffers", "pinned_library_buffers"):
            try:
                if label.startswith("pinned"):
                    a, o, r, d = e.HostBuffers()
                else:
                    a = np.empty(n, e._adtype); o = np.empty((n, obs_dim), np.float32)
                    r = np.empty(n, np.float32); d = np.empty(n, np.uint8)
                    o.fill(0); r.fill(0); d.fill(0)             # touch the pages: no first-use faults inside the timing
                a[:] = e.SampleActions(seed=seed + 1, tick=0)
                for _ in range(3):
                    e.StepInto(a, o, r, d)
                ts = []
                for _ in range(steps):
                    t0 = time.perf_counter()
                    e.StepInto(a, o, r, d)
                    ts.append(time.perf_counter() - t0)
                ms = median(ts) * 1e3
                out[label] = {"ms_per_step": ms, "env_steps_per_sec": n / (ms * 1e-3), "pcie_GBps": boundary_bytes / (ms * 1e-3) / 1e9}
            except Exception as ex:                               # noqa: BLE001
                out[label] = {"error": repr(ex)[:200]}
    out["boundary_bytes_per_step"] = boundary_bytes
    return out


def group_leg(args, members):
    """The multi-GPU entry a P/Invoking host has (the reference has no torch.distributed): ONE process driving `members`
    GPUs through gymnet_group_* (csrc/group.hip) — step-only, then a step + observation all-gather per step with the
    hand-written direct push (serial, and overlapped with the next step through double-buffered observations) and with
    RCCL (ncclCommInitAll inside the library).  Runs in a FRESH child process (started with subprocess, never exec'd from a
    process that touched the GPU); prints one JSON object.  With fewer devices than members the members are logical
    (several per device): plumbing only, labelled."""
    import torch
    import __graft_entry__ as ge
    pkg = ge.load_package()
    ndev = pkg.device_count()
    G, n, ring = members, args.num_envs, 8
    devices = [m % ndev for m in range(G)]
    res = {"members": G, "devices_visible": ndev, "device_of_member": devices, "lanes_per_member": n,
           "real_multi_gpu": ndev >= G,
           "note": ("one process, one device per member" if ndev >= G else
                    "FEWER DEVICES THAN MEMBERS: members are logical and share devices — plumbing check, not a multi-GPU measurement")}
    adtype = torch.float32 if args.env == "Pendulum-v1" else torch.int32
    acts = [torch.empty((ring, n), dtype=adtype, device=f"cuda:{d}") for d in devices]
    K = max(args.steps, 64)

    def timed(fn, sync, reps=5):
        ts = []
        for _ in range(reps):
            sync()
            t0 = time.perf_counter()
            fn()
            sync()
            ts.append(time.perf_counter() - t0)
        return median(ts)

    for label, gather, overlap in (("step_only", "none", False), ("direct", "direct", False), ("direct_overlapped", "direct", True),
                                   ("rccl", "rccl", False)):
        if gather == "rccl" and ndev < G:
            res[label] = {"skipped": "RCCL needs one device per member"}
            continue
        try:
            with pkg.GroupVectorEnv(args.env, n * G, G, devices=devices, seed=0x5EED, auto_reset=True, gather=gather, overlap=overlap,
                                    dtype="float64" if args.dtype == "f64" else "float32") as grp:
                for m, mem in enumerate(grp.Members):
                    for t in range(ring):
                        mem.SampleActionsDevice(acts[m][t].data_ptr(), seed=0x5EED + 1, tick=t)
                grp.ResetDevice()
                grp.Sync()
                if gather == "none":
                    ptrs = [a.data_ptr() for a in acts]
                    grp.RolloutDevice(ptrs, K, n, ring)
                    wall = timed(lambda: grp.RolloutDevice(ptrs, K, n, ring), grp.Sync)
                    steps = K
                else:
                    steps = 64
                    slices = [[acts[m][t].data_ptr() for m in range(G)] for t in range(ring)]

                    def loop():
                        for t in range(steps):
                            grp.StepDevice(slices[t % ring])
                            grp.AllGatherObs()
                        grp.WaitGather()
                    loop()
                    wall = timed(loop, grp.Sync)
                    rep = grp.ReadReplica(G - 1)             # [G, D, n]: the LAST member's view of everyone's observations
                    ok = bool((abs(rep).reshape(G, -1).sum(axis=1) > 0).all()) and bool((rep == rep).all())
                res[label] = {"value": n * G * steps / wall, "unit": "env-steps/s", "us_per_step": wall / steps * 1e6, "steps": steps}
                if gather != "none":
                    res[label]["every_member_slice_arrived"] = ok
                    res[label]["allgather_bytes_per_member_per_step"] = grp.ObsDim * n * 4
        except Exception as e:                                   # noqa: BLE001 - the leg is optional; report, never hide
            res[label] = {"error": repr(e)[:300]}
    print(json.dumps(res), flush=True)


def run_group_child(args, members, timeout=300):
    """Starts the gymnet_group_* leg as a fresh child process and returns its JSON (or an error record)."""
    cmd = [sys.executable, os.path.abspath(__file__), "--group-child", str(members), "--env", args.env,
           "--num-envs", str(args.num_envs), "--steps", str(args.steps), "--dtype", getattr(args, "dtype", "f32")]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_PORT",
                                                            "GROUP_RANK", "ROLE_RANK", "TORCHELASTIC_RUN_ID")}
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
        for line in reversed(r.stdout.splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {"error": f"rc {r.returncode}: {(r.stderr or r.stdout)[-300:]}"}
    except subprocess.TimeoutExpired:
        return {"error": f"timed out after {timeout} s"}
    except Exception as e:                                       # noqa: BLE001
        return {"error": repr(e)[:300]}


def main():
    args = parse()
    # the host driver only supports dmabuf IPC: without this, RCCL and the HIP-IPC peer buffers fail with
    # "hipIpcGetMemHandle: invalid argument" — set before anything loads the HIP runtime, in every launch form
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.group_child:
        return group_leg(args, args.group_child)
    if args.rollout_child:
        return rollout_child(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)
    # lower completion latency of the closing synchronize: ROCr polls its signals instead of sleeping on an interrupt
    os.environ.setdefault("HSA_ENABLE_INTERRUPT", os.environ.get("GYMNET_BENCH_HSA_INTERRUPT", "0"))
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    pkg = ge.load_package()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if pkg.device_count() < 1:
        raise SystemExit("bench.py needs an AMD GPU: the engine has no CPU fallback")
    # one process per GPU; GYMNET_BENCH_BACKEND=gloo lets several ranks share the GPUs that exist (a 1-GPU box can then
    # exercise the N > 1 plumbing: lane offsets, shard buffers, barrier, max-over-ranks) — RCCL needs one GPU per rank
    # — also chosen automatically when a launcher (torch.distributed.run) starts more ranks than the node has GPUs
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    n_dev = torch.cuda.device_count()            # does not initialise the GPU on this image
    backend = os.environ.get("GYMNET_BENCH_BACKEND", "nccl" if n_dev >= local_world else "gloo")
    dev_index = local_rank if backend == "nccl" else local_rank % max(1, n_dev)
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
    can_gather = use_dist and backend == "nccl"
    # the headline always steps IN PLACE (double-buffered observation arrays cost the step kernel ~0.8 us at this size,
    # profiles/double_buffer_probe_r02.txt); the overlapped-gather figure below builds its own double-buffered shard
    f64 = args.dtype == "f64"
    if f64 and args.env != "CartPole-v1":
        raise SystemExit("--dtype f64 is CartPole's reference-arithmetic mode (the one env whose float64 arithmetic the reference defines)")
    esz = 8 if f64 else 4          # bytes per observation element
    dt_name = "float64" if f64 else "float32"
    # (since round 5 the float64 mode shards, gathers and groups like the float32 engine: the same ShardedVectorEnv, gather buffers of doubles)
    env = pkg.ShardedVectorEnv(args.env, n * world, rank=rank, world_size=world, device=dev_index, seed=seed,
                               auto_reset=True, gather_obs=use_dist, tensor_device=dev,
                               force_gather=args.force_dist, overlap=False, dtype=dt_name)
    local = env.local
    if args.policy:
        local.SetLaunchPolicy(**parse_policy(args.policy))
    adtype = torch.float32 if local._adtype.__name__ == "float32" else torch.int32
    actions = torch.empty((ring, n), dtype=adtype, device=dev)
    for t in range(ring):      # ActionSpace.Sample() per lane per step, on the device (Philox stream "action", key = seed + 1)
        local.SampleActionsDevice(actions[t].data_ptr(), seed=seed + 1, tick=t)
    env.ResetDevice()
    env.Sync()

    gather_in_region = bool(args.allgather and use_dist)

    def run(steps, t0=0):
        if gather_in_region:
            for t in range(steps):
                env.StepDevice(actions[(t0 + t) % ring].data_ptr())
                env.AllGatherObs()
        elif args.no_graph:
            for t in range(steps):
                env.StepDevice(actions[(t0 + t) % ring].data_ptr())
        else:               # the C loop handles the double-buffered observation arrays itself
            local.RolloutDevice(actions.data_ptr(), steps, n, ring)

    # barrier: a shared-memory spin barrier between the node's ranks when it can be set up, else the process group's
    node_barrier = None
    if use_dist and os.environ.get("GYMNET_BENCH_BARRIER", "shm") == "shm":
        ok, nb = 1, None
        try:
            nb = NodeBarrier(rank, world, os.environ.get("MASTER_PORT", "0"))
        except Exception:
            ok = 0
        dist.barrier()                                       # rank 0 has created the file
        try:
            if ok:
                nb.attach()
        except Exception:
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=red_dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)          # every rank uses the same kind of barrier, or none does
        if int(flag[0]) == 1:
            nb()                                             # and it works
            node_barrier = nb

    def barrier():
        if node_barrier is not None:
            node_barrier()
        elif use_dist:
            dist.barrier()

    def timed_region(fn, events=True):
        """One bracketed region: barrier + synchronize, fn(), synchronize + barrier.  Returns (wall s, event ms or -1).
        events=False leaves the two HIP-event records (a few host us each) out of the wall-clock bracket."""
        if events:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(dev)
        barrier()
        t0 = time.perf_counter()
        if events:
            e0.record(stream)
        fn()
        if events:
            e1.record(stream)
        torch.cuda.synchronize(dev)
        barrier()
        t1 = time.perf_counter()
        return t1 - t0, (e0.elapsed_time(e1) if events else -1.0)      # -1: a bare region (survives the MAX all-reduce, unlike NaN)

    def repeat_until(fn, min_seconds, max_repeats=MAX_REPEATS):
        """Repeats the bracketed region until min_seconds have been timed (every rank runs the same count: the count
        is fixed from rank 0's first region).  Returns per-repeat (wall, event_ms) maxima over ranks."""
        first = timed_region(fn)
        reps = max(3, min(max_repeats, int(min_seconds / max(first[0], 1e-7)) + 1))
        if use_dist:
            tr = torch.tensor([reps], dtype=torch.int64, device=red_dev)
            dist.broadcast(tr, 0)
            reps = int(tr[0])
        # odd repeats carry the HIP events (kernel-duration figure), even repeats are the bare wall-clock bracket
        rows = [first] + [timed_region(fn, events=(i % 2 == 0)) for i in range(reps - 1)]
        own_rows[:] = rows                                       # this rank's own clocks, before the MAX over ranks
        if use_dist:
            tw = torch.tensor(rows, dtype=torch.float64, device=red_dev)
            dist.all_reduce(tw, op=dist.ReduceOp.MAX)
            rows = [(float(a), float(b)) for a, b in tw.tolist()]
        return rows

    run(W)
    torch.cuda.synchronize(dev)
    steps_before = local.Counters()["lane_steps"]
    own_rows = []
    rows = repeat_until(lambda: run(K), args.min_seconds)
    headline_rows = list(own_rows)                               # the headline's regions (later repeat_until calls overwrite own_rows)
    bare = [r[0] for r in rows if r[1] < 0] or [r[0] for r in rows]          # regions without event records
    walls = bare
    wall = median(bare)
    ev_ms = median([r[1] for r in rows if r[1] >= 0])
    repeats = len(rows)

    # sanity: the engine really ran K steps on every lane in every repeat
    c = local.Counters()
    assert c["lane_steps"] - steps_before == repeats * K * n, (c, steps_before, repeats)

    # CartPole: 41 B; float64 mode: 73 B (32 + 4 read, 32 + 4 + 1 written).  Envs that store a state row once,
    # in the observation (Pendulum 33 of 37 B, Acrobot 57 of 65 B), are priced on the bytes they MOVE: the
    # algorithmic figure is printed beside it, never used for the fraction.
    algo_bytes_per_step = 73 if f64 else local.AlgorithmicBytesPerStep
    bytes_per_step = 73 if f64 else local.TrafficBytesPerStep
    launch_policy = local.LaunchPolicy()

    def headline():
        """The contract's JSON line from the main timing alone; the secondary figures are added to it as they arrive."""
        launch_us = ev_ms * 1e3 / K                                          # HIP events over the (median) timed region / launches
        by_events = bytes_per_step * n / (launch_us * 1e-6) / 1e9            # GB/s per GPU from the kernel-side clock
        achieved = bytes_per_step * n / (wall / K) / 1e9                     # GB/s per GPU from the SAME wall clock `value` uses
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")               # rocprofv3 --pmc result, per launch
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic = tj.get(args.env + ("-f64" if f64 else ""), {}).get(str(n))
                if traffic is not None:
                    traffic_source = ("NOT measured in this run: constant read from profiles/traffic.json — "
                                      + str(tj.get("_source", "rocprofv3 --pmc passes")))
            except Exception:
                traffic = None
        if gather_in_region:
            what = "value = step + RCCL observation all-gather after every step (NOT the step-only headline)"
        elif world > 1:
            what = "value = step-only rate (no collective on the data path); with_obs_allgather = the same stepping plus the per-step RCCL gather"
        else:
            what = "value = step-only rate"
        kernel = local.KernelName()          # the instantiation the launcher resolves to, printed by the library itself
        return {
            "metric": "env-steps/sec", "value": n * world * K / wall, "unit": "env-steps/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": wall * 1e3 / K,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64" if f64 else "f32", "data": "synthetic",
            "repeats": repeats, "region_ms_median": wall * 1e3, "region_ms_min": min(walls) * 1e3, "region_ms_max": max(walls) * 1e3,
            "config": {"workload": f"{args.env} batched, batch={n} lanes per GPU (global {n * world}), {'float64' if f64 else 'float32'} SoA state, "
                                   f"fused auto-reset, iid random actions pre-generated in HBM; {what}",
                       "num_envs_per_gpu": n, "global_num_envs": n * world, "action_ring": ring,
                       "launch": ("one kernel launch per step, eager (python loop)" if (args.no_graph or gather_in_region) else
                                  "one kernel launch per step; gymnet_vecenv_rollout_device: " +
                                  ("hipGraph replay" if (parse_policy(args.policy).get("graph", -1) == 1 or
                                                         (parse_policy(args.policy).get("graph", -1) != 0 and n * bytes_per_step < (24 << 20)))
                                   else "back-to-back stream launches")),
                       "launch_policy": launch_policy,
                       "timing": f"median of {repeats} bracketed {K}-step regions (>= {args.min_seconds * 1e3:.0f} ms timed in total)",
                       "backend": ("rccl" if backend == "nccl" else backend + " (ranks SHARE the GPUs that exist: plumbing check, not a multi-GPU measurement)") if use_dist else "single process",
                       "barrier": ("shared-memory spin barrier (ranks of one node)" if node_barrier is not None else
                                   ("process-group barrier" if use_dist else "none (one rank)")),
                       "allgather_obs_in_timed_region": gather_in_region, "double_buffered_obs": False,
                       "parallelism": f"lane-sharded x{world}"},
            # `achieved` / `frac` are the WALL-CLOCK figures (driver-comparable: bytes per launch / ms_per_step), the conservative
            # ones; the HIP-event figures — the kernel's own duration over the timed region, what rocprofv3 --stats reproduces —
            # sit beside them (the event-based fraction flatters a 20-step region that pays a ~14 us bracket)
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": kernel,
                         "bytes_per_env_step": bytes_per_step, "algorithmic_bytes_per_env_step": algo_bytes_per_step,
                         "bytes_per_launch": bytes_per_step * n, "algorithmic_bytes_per_launch": algo_bytes_per_step * n,
                         "launch_us": launch_us, "achieved_by_events": by_events, "frac_by_events": by_events / HBM_PEAK_GBPS,
                         "frac_by_wall": achieved / HBM_PEAK_GBPS,
                         "note": "at 2^20 lanes the working set is Infinity-Cache resident; see hbm_resident_2p27 for real HBM",
                         # the step kernels are short of WRITE bandwidth, not of total bytes: the written half of this pattern alone runs
                         # at 4.4-4.8 TB/s whatever the store flavour, the read half overlaps it almost entirely (a copy of both takes
                         # 5-6 % longer than the writes alone) — profiles/write_path_probe_r02.txt, store_flavour_r05.txt, skeleton_floor_r05.txt
                         "write_path": {"written_bytes_per_env_step": WRITTEN_BYTES.get((args.env, f64)),
                                        "measured_pure_write_GBps": [4400, 4800],
                                        "note": "launch time ~ time to the first store (~1-1.7 us) + written bytes / pure-write rate"}},
        }

    out = headline() if rank == 0 else {}
    emitted = threading.Lock()

    def emit_and_exit_on_timeout(section, seconds):
        return make_watchdog(rank, out, emitted, section, seconds)

    # N > 1: what each rank ran on and measured by itself, and proof that the collective backend really spans `world` ranks.
    # Nothing here may cost the headline: a rank that cannot describe itself still takes part in both collectives (so nobody
    # waits for it), and the section runs under the watchdog.
    if use_dist:
        watchdog = emit_and_exit_on_timeout("ranks", 90)
        mine = {"rank": rank, "local_rank": local_rank, "device": dev_index, "pid": os.getpid()}
        try:
            props = torch.cuda.get_device_properties(dev)
            mine.update(name=props.name, uuid=str(getattr(props, "uuid", "")), pci_bus_id=getattr(props, "pci_bus_id", None),
                        kernel=local.KernelName(),
                        events_us_per_step=median([r[1] for r in headline_rows if r[1] >= 0]) * 1e3 / K,
                        wall_us_per_step=median([r[0] for r in headline_rows if r[1] < 0] or [r[0] for r in headline_rows]) * 1e6 / K)
        except Exception as e:                                   # noqa: BLE001
            mine["error"] = repr(e)[:200]
        everyone = [None] * world
        dist.all_gather_object(everyone, mine)
        probe = torch.tensor([rank + 1], dtype=torch.int64, device=red_dev)
        dist.all_reduce(probe, op=dist.ReduceOp.SUM)            # every rank contributes rank + 1: sum = N (N + 1) / 2
        if rank == 0:
            out["ranks"] = everyone
            out["collective"] = {"backend": dist.get_backend(), "is_rccl": dist.get_backend() == "nccl",
                                 "world_size": dist.get_world_size(),
                                 "allreduce_sum_of_rank_plus_1": int(probe[0]), "expected": world * (world + 1) // 2,
                                 "distinct_devices": len({(e.get("uuid"), e.get("pci_bus_id"), e["device"]) for e in everyone})}
        watchdog.cancel()


    extras = rank == 0 and world == 1 and not args.no_extras and not gather_in_region and not f64
    # Cross-check of the per-launch figure: 200 single launches, each bracketed by its own HIP-event pair on the
    # engine's stream (isolated launches: no back-to-back overlap with a neighbour's ramp / drain).
    single_us = None
    if extras:
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(200)]
        for t, (a0, a1) in enumerate(evs):
            a0.record(stream)
            local.StepDevice(actions[t % ring].data_ptr())
            a1.record(stream)
        torch.cuda.synchronize(dev)
        ds = sorted(a0.elapsed_time(a1) * 1e3 for a0, a1 in evs)
        single_us = ds[len(ds) // 2]

    # Secondary figure, NOT the headline: the same steps fused into one launch per `ring` steps (state stays in
    # registers, gymnet_vecenv_rollout_fused_device) — open-loop rollouts only, so it is reported beside, not as, `value`.
    fused = None
    if extras:
        try:
            fsteps = max(ring, (min(max(K, 1024), 2048) // ring) * ring)
            local.RolloutFusedDevice(actions.data_ptr(), ring, n, ring)
            torch.cuda.synchronize(dev)
            f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            f0.record(stream)
            for _ in range(fsteps // ring):
                local.RolloutFusedDevice(actions.data_ptr(), ring, n, ring)
            f1.record(stream)
            torch.cuda.synchronize(dev)
            fused_us = f0.elapsed_time(f1) * 1e3 / fsteps
            obs_dim = local.ObsDim
            fused = {"env_steps_per_sec_per_gpu": n / (fused_us * 1e-6), "us_per_step": fused_us, "steps_per_launch": ring,
                     "bytes_per_env_step": 4, "bytes_note": "4 B action read per env-step (nothing recorded); state in registers",
                     "note": "T-step fused kernel, no per-step observation hand-off; not comparable to `value`"}

            def fused_time(fn, reps=3):
                fn()
                torch.cuda.synchronize(dev)
                ts = []
                for _ in range(reps):
                    g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    g0.record(stream)
                    for _ in range(fsteps // ring):
                        fn()
                    g1.record(stream)
                    torch.cuda.synchronize(dev)
                    ts.append(g0.elapsed_time(g1) * 1e3 / fsteps)
                return median(ts)
            # Round 5 (gymnet_vecenv_rollout_fused_ex_device): the actions DRAWN IN THE KERNEL — ActionSpace.Sample() per lane and
            # step, the words gymnet_vecenv_sample_actions_device would write — so a random rollout reads no action ring at all:
            # 0 B per env-step from memory, ~70 more VALU per lane-step for the Philox call.
            su = fused_time(lambda: local.RolloutFusedDevice(None, ring, actions="sample", action_seed=seed + 1, action_tick0=0))
            fused["sampled_actions"] = {"us_per_step": su, "env_steps_per_sec_per_gpu": n / (su * 1e-6), "bytes_per_env_step": 0,
                                        "note": "actions drawn in the kernel (Philox action stream v2: one call per four lanes): no action ring is read, nothing recorded"}
            # ... and composed epsilon-greedy over the ring as the policy's actions (TrainingPlaySession.cs:46-52): a second word per lane
            if adtype == torch.int32:
                gu = fused_time(lambda: local.RolloutFusedDevice(actions.data_ptr(), ring, n, ring, actions="epsilon_greedy", action_seed=seed + 1,
                                                                 action_tick0=0, epsilon=0.1))
                fused["epsilon_greedy_actions"] = {"us_per_step": gu, "env_steps_per_sec_per_gpu": n / (gu * 1e-6), "bytes_per_env_step": 4, "epsilon": 0.1,
                                                   "over_sampled": gu / su,
                                                   "note": "the ring holds the policy's actions; explore iff u01(word B) <= epsilon, then ActionSpace.Sample() (word A)"}
            # ... and the same recording what a replay memory stores (ReplayMemory.cs:53-67): observation, action, reward, done
            rec_o = torch.empty((ring, obs_dim, n), dtype=torch.float32, device=dev)
            rec_r = torch.empty((ring, n), dtype=torch.float32, device=dev)
            rec_d = torch.empty((ring, n), dtype=torch.uint8, device=dev)
            rec_a = torch.empty((ring, n), dtype=adtype, device=dev)
            ru = fused_time(lambda: local.RolloutFusedDevice(None, ring, actions="sample", action_seed=seed + 1, rec_obs=rec_o.data_ptr(),
                                                             rec_reward=rec_r.data_ptr(), rec_done=rec_d.data_ptr(), rec_actions=rec_a.data_ptr()))
            rb = 4 * obs_dim + 4 + 4 + 1
            fused["sampled_actions_recorded"] = {"us_per_step": ru, "env_steps_per_sec_per_gpu": n / (ru * 1e-6), "bytes_per_env_step": rb,
                                                 "achieved_GBps": rb * n / (ru * 1e-6) / 1e9, "frac_of_peak": rb * n / (ru * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                                                 "roofline": write_roofline(n, rb, ru),
                                                 "note": f"{rb} B written per env-step (obs {4 * obs_dim} + action 4 + reward 4 + done 1), nothing read"}
            del rec_o, rec_r, rec_d, rec_a
            # ... and on a BOOKKEEPING handle: episode return / length in registers, 500-step time limit, one compact
            # (t, lane, return, length) record per finished episode (BasePlaySession.cs:58-69)
            if args.env == "CartPole-v1":
                with pkg.VectorEnv(args.env, n, device=dev_index, seed=seed, auto_reset=True, episode_stats=True, max_episode_steps=500,
                                   stream=stream.cuda_stream) as be:
                    cap = n * ring // 8
                    keep = {"step": torch.empty(cap, dtype=torch.int32, device=dev), "lane": torch.empty(cap, dtype=torch.int32, device=dev),
                            "ret": torch.empty(cap, dtype=torch.float32, device=dev), "length": torch.empty(cap, dtype=torch.int32, device=dev),
                            "capacity": cap, "count": torch.zeros(2, dtype=torch.uint32, device=dev)}
                    ep = {k: (v.data_ptr() if hasattr(v, "data_ptr") else v) for k, v in keep.items()}
                    be.ResetDevice()
                    bu = fused_time(lambda: be.RolloutFusedDevice(None, ring, actions="sample", action_seed=seed + 1, episodes=ep))
                    cnt = keep["count"].cpu().numpy()
                    bf = fused_time(lambda: be.RolloutFusedDevice(None, ring, actions="sample", action_seed=seed + 1, episodes=dict(ep, no_overflow=True)))
                    fused["sampled_actions_with_episode_records"] = {
                        "us_per_step": bu, "env_steps_per_sec_per_gpu": n / (bu * 1e-6), "episodes_per_launch": int(cnt[1]), "records_kept": int(cnt[0]),
                        "us_per_step_no_overflow_variant": bf,
                        "note": "bookkeeping handle (EPISODE_STATS, max_episode_steps 500): episode statistics in registers + one compact "
                                "(t, lane, return, length) record per finished episode, gathered after the launch; default = nothing lost below the capacity "
                                "(per-shard segments spill to a shared overflow segment), *_no_overflow_variant = GYMNET_RECORDS_NO_OVERFLOW"}
        except Exception as e:                                   # noqa: BLE001 - a secondary figure never costs the headline
            fused = {"error": repr(e)[:300]}

    # (The per-configuration figures below run BEFORE the sections that allocate and free gigabytes — the 2 GiB copy probe, the
    # 2^27-lane batch: measured after them, the very same kernels ran 4-8 % slower (Acrobot 12.4-13.0 vs 11.5-11.9 us, the float64
    # kernel 14.3-14.8 vs 13.1 us standalone), a placement effect of buffers allocated out of a fragmented pool.)
    def timed_rollouts(e, acts, ring_len, launches=1024, reps=5):
        """(HIP-event us per launch, wall seconds per region): 2048 untimed launches, then the median of `reps` back-to-back regions
        of `launches` one-launch steps.  The arithmetic-heavy kernels (Acrobot, the float64 CartPole) keep speeding up over the
        first ~50 ms of sustained launches — 14.1, 12.3, 12.2, 12.1, 11.9 us in consecutive 1024-launch regions
        (tools/acrobot_alloc_probe.py) — so a figure taken right after a 128-launch warm-up reads 5-8 % slow against the
        headline protocol's (`--env E`: 512 warm-up launches, median of >= 50 ms of regions)."""
        e.RolloutDevice(acts.data_ptr(), 2048, n, ring_len)
        evs, walls = [], []
        for _ in range(reps):
            torch.cuda.synchronize(dev)
            q0, q1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            q0.record(stream)
            e.RolloutDevice(acts.data_ptr(), launches, n, ring_len)
            q1.record(stream)
            torch.cuda.synchronize(dev)
            walls.append(time.perf_counter() - t0)
            evs.append(q0.elapsed_time(q1) * 1e3 / launches)
        return median(evs), median(walls)

    # Secondary figures, NOT the headline: BASELINE.json's other single-GPU configs (3: Pendulum-v1, 4: Acrobot-v1 at 2^20 lanes) and
    # MountainCar, each through its own bench-shaped rollout — HIP events over 1024 back-to-back launches — so that the driver's
    # record carries their roofline fractions too (they are parity-test cases; their full lines come from `--env E`).
    other = None
    if extras and args.env == "CartPole-v1" and n == (1 << 20):
        other = {}
        for name in ("Pendulum-v1", "Acrobot-v1", "MountainCar-v0"):
            try:
                with pkg.VectorEnv(name, n, device=dev_index, seed=seed, auto_reset=True, stream=stream.cuda_stream) as e3:
                    r3 = 32
                    a3 = torch.empty((r3, n), dtype=torch.float32 if name == "Pendulum-v1" else torch.int32, device=dev)
                    for t in range(r3):
                        e3.SampleActionsDevice(a3[t].data_ptr(), seed=seed + 1, tick=t)
                    e3.ResetDevice()
                    e3.RolloutDevice(a3.data_ptr(), 128, n, r3)
                    us, _ = timed_rollouts(e3, a3, r3)
                    gb = e3.TrafficBytesPerStep * n / (us * 1e-6) / 1e9              # the bytes the kernel MOVES
                    gba = e3.AlgorithmicBytesPerStep * n / (us * 1e-6) / 1e9
                    other[name] = {"kernel": e3.KernelName(), "launch_us": us, "env_steps_per_sec": n / (us * 1e-6),
                                   "algorithmic_bytes_per_step": e3.AlgorithmicBytesPerStep, "moved_bytes_per_step": e3.TrafficBytesPerStep,
                                   "achieved_GBps": gb, "frac_of_peak": gb / HBM_PEAK_GBPS,
                                   "algorithmic_GBps": gba, "frac_of_peak_algorithmic_bytes": gba / HBM_PEAK_GBPS,
                                   "clock": "HIP events over 1024 back-to-back launches, median of 5 such regions after 2048 warm-up launches"}
                    del a3
            except Exception as e:                               # noqa: BLE001 - a secondary figure never costs the headline
                other[name] = {"error": repr(e)[:200]}

    # Secondary figure, NEVER `value`: CartPole at the same batch in the reference's own float64 arithmetic (GYMNET_FLAG_F64,
    # cartpole64.hpp) — 73 B per env-step: 32 + 4 read, 32 + 4 + 1 written — with its own 73 B roofline fraction.
    f64_fig = None
    if extras and args.env == "CartPole-v1":
        try:
            with pkg.VectorEnv("CartPole-v1", n, device=dev_index, seed=seed, auto_reset=True, stream=stream.cuda_stream, dtype="float64") as e4:
                r4 = 32
                a4 = torch.empty((r4, n), dtype=torch.int32, device=dev)
                for t in range(r4):
                    e4.SampleActionsDevice(a4[t].data_ptr(), seed=seed + 1, tick=t)
                e4.ResetDevice()
                e4.RolloutDevice(a4.data_ptr(), 128, n, r4)
                us, w4 = timed_rollouts(e4, a4, r4)
                fsteps4 = 1024
                e4.RolloutFusedDevice(a4.data_ptr(), 256, n, r4)
                torch.cuda.synchronize(dev)
                g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                g0.record(stream)
                for _ in range(fsteps4 // 256):
                    e4.RolloutFusedDevice(a4.data_ptr(), 256, n, r4)
                g1.record(stream)
                torch.cuda.synchronize(dev)
                fused64_us = g0.elapsed_time(g1) * 1e3 / fsteps4

                def fused64_time(fn):
                    fn()
                    torch.cuda.synchronize(dev)
                    h0, h1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    h0.record(stream)
                    for _ in range(fsteps4 // 256):
                        fn()
                    h1.record(stream)
                    torch.cuda.synchronize(dev)
                    return h0.elapsed_time(h1) * 1e3 / fsteps4
                sampled64_us = fused64_time(lambda: e4.RolloutFusedDevice(None, 256, actions="sample", action_seed=seed + 1, action_tick0=0))
                eps64_us = fused64_time(lambda: e4.RolloutFusedDevice(a4.data_ptr(), 256, n, r4, actions="epsilon_greedy", action_seed=seed + 1,
                                                                      action_tick0=0, epsilon=0.1))
                f64_fig = {"kernel": e4.KernelName(), "num_envs": n, "bytes_per_env_step": 73, "launch_us": us,
                           "fused_rollout": {"us_per_step": fused64_us, "env_steps_per_sec": n / (fused64_us * 1e-6), "steps_per_launch": 256,
                                             "sampled_actions": {"us_per_step": sampled64_us, "env_steps_per_sec": n / (sampled64_us * 1e-6)},
                                             "epsilon_greedy_actions": {"us_per_step": eps64_us, "env_steps_per_sec": n / (eps64_us * 1e-6), "epsilon": 0.1,
                                                                        "over_sampled": eps64_us / sampled64_us},
                                             "note": "T-step fused float64 kernel (state in registers, arithmetic-bound); open-loop rollouts only"},
                           "env_steps_per_sec": n * 1024 / w4, "achieved_GBps": 73 * n / (w4 / 1024) / 1e9,
                           "frac_of_peak": 73 * n / (w4 / 1024) / 1e9 / HBM_PEAK_GBPS,
                           "frac_of_peak_by_events": 73 * n / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                           "note": "reference-exact float64 mode; reported beside, never as, `value`"}
                del a4
        except Exception as e:                                   # noqa: BLE001 - a secondary figure never costs the headline
            f64_fig = {"error": repr(e)[:300]}

    # Secondary figure, NEVER `value`: the NDArray-shaped host boundary a C# VectorEnv.Step(NDArray) reaches — gymnet_vecenv_step
    # with caller-owned host buffers, PCIe both ways inside the call (VecEnvWrapper.cs:22-24, Step.cs:8-10): (a) ordinary
    # pageable caller memory, (b) the library's pinned, device-mapped buffers (gymnet_vecenv_host_buffers: zero staging).
    host_boundary = None
    if extras and not args.no_host_boundary:
        try:
            host_boundary = measure_host_boundary(pkg, args.env, n, dev_index, seed, algo_bytes_per_step)
        except Exception as e:                                   # noqa: BLE001 - a secondary figure never costs the headline
            host_boundary = {"error": repr(e)[:300]}

    # Secondary figure, NEVER `value`: BASELINE config 1's shape on the GPU — ONE instance stepped through the single-instance façade
    # (README.md:32-52: Step, `if (done) Reset()`), which is latency-bound by construction (a kernel launch + a host round trip
    # per step): the honest counterpart of cpu_baseline.single_instance_100k_steps_per_sec.  float64 = the façade's default.
    facade = None
    if extras and args.env == "CartPole-v1":
        try:
            facade = {"note": "N = 1 through the host boundary (Python facade, ctypes).  float64_default = the facade as a drop-in gets it: one kernel launch + "
                              "one synchronize per call.  *_resident = GYMNET_FLAG_RESIDENT (opt-in since round 6): a resident single-wave kernel polling a "
                              "mailbox in pinned host memory, no launch / synchronize per step — for loops that only step (a device-wide synchronize "
                              "elsewhere in the process waits for its ~5 ms idle timeout).  The engine is built for batches"}
            for label, dt, res in (("float64_default", "float64", False), ("float64_resident", "float64", True), ("float32_resident", "float32", True)):
                cp = pkg.CartPoleEnv(device=dev_index, seed=seed, dtype=dt, resident=res)
                try:
                    cp.Reset()
                    for i in range(200):
                        if cp.Step(i % 2).Done:
                            cp.Reset()
                    t0 = time.perf_counter()
                    m, eps = 3000, 0
                    for i in range(m):
                        if cp.Step(i % 2).Done:
                            cp.Reset(); eps += 1
                    dt_s = time.perf_counter() - t0
                    facade[label] = {"steps_per_sec": m / dt_s, "us_per_step": dt_s / m * 1e6, "episodes": eps}
                finally:
                    cp.CloseEnvironment()
        except Exception as e:                                   # noqa: BLE001 - a secondary figure never costs the headline
            facade = {"error": repr(e)[:300]}

    # Secondary figures for N > 1, NOT the headline: the same stepping with the RCCL all-gather of observations
    # north_star mentions after EVERY step (in place, rank-major [G][D][N/G] buffer).  The stepping path itself needs no
    # collective; this shows what a consumer that wants every rank to see all observations pays over xGMI —
    # (a) with the gather on the critical path, (b) overlapped with the next step (double-buffered observation arrays:
    # step t+1 writes buffer B while the gather of buffer A is in flight on a side stream).
    gathered = None
    if use_dist and not gather_in_region:
        gathered = {}
        gs = 128
        watchdog = emit_and_exit_on_timeout("with_obs_allgather", int(os.environ.get("GYMNET_BENCH_WATCHDOG", "120")))
        # serial / overlapped: RCCL all_gather_into_tensor; direct_ipc / direct_ipc_overlapped: the hand-written push of each
        # rank's slice into its peers' replica buffers (HIP IPC peer buffers, gymnet_push_obs_device; one xGMI link per peer)
        for label, overlapped, how in (("serial", False, "rccl"), ("overlapped", True, "rccl"),
                                       ("direct_ipc", False, "direct"), ("direct_ipc_overlapped", True, "direct")):
            if overlapped and args.no_overlap:
                continue
            if how == "rccl" and backend != "nccl":      # ranks sharing a GPU over gloo: only the IPC variants can run
                continue
            genv = None
            own = overlapped or how == "direct"
            try:
                genv = env if not own else pkg.ShardedVectorEnv(
                    args.env, n * world, rank=rank, world_size=world, device=dev_index, seed=seed, auto_reset=True,
                    gather_obs=True, tensor_device=dev, force_gather=args.force_dist, overlap=overlapped, gather=how,
                    barrier=node_barrier, dtype=dt_name)
                if own:
                    genv.ResetDevice()

                def gloop(steps=gs):
                    for t in range(steps):
                        genv.StepDevice(actions[t % ring].data_ptr())
                        genv.AllGatherObs(overlap=overlapped)
                    genv.WaitGather()
                gloop(16)
                grows = repeat_until(gloop, args.min_seconds, max_repeats=16)
                gwall = median([r[0] for r in grows])
                obs = genv.LastGatheredObs()
                ok = bool(torch.isfinite(obs).all()) and all(float(obs[r].abs().sum()) > 0 for r in range(world))
                gathered[label] = {"value": n * world * gs / gwall, "unit": "env-steps/s", "ms_per_step": gwall * 1e3 / gs,
                                   "steps": gs, "repeats": len(grows), "gathered_obs_finite_and_nonzero": ok}
            except Exception as e:                      # never lose the headline over the optional collective
                gathered[label] = {"error": repr(e)[:200]}
                sys.stderr.write(f"[bench rank {rank}] with_obs_allgather/{label}: {e!r}\n")
            finally:
                if own and genv is not None:
                    genv.Sync()
                    genv.Close()
        gathered["allgather_bytes_per_rank_per_step"] = env.obs_dim * n * esz
        gathered["xgmi_model"] = allgather_model(world, env.obs_dim * n * esz)
        watchdog.cancel()

    # Attainable copy bandwidth on THIS box (read + write bytes / time of a device-to-device float copy), reported
    # beside the 8 TB/s spec peak the roofline fraction uses: cache-resident (32 MiB) and HBM-resident (2 GiB).
    copy_bw = None
    if extras:
        copy_bw = {}
        for label, mib in (("32MiB", 32), ("2GiB", 2048)) if torch.cuda.mem_get_info(dev)[0] > (6 << 30) else (("32MiB", 32),):
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

    env.Close()
    del actions
    torch.cuda.empty_cache()

    # Secondary figure, NOT the headline: the same kernel at 2^27 lanes — 5.5 GB per step, 2 GiB of state: nothing
    # can stay in the 256 MiB Infinity Cache, so this is what the path does against real HBM (at 2^20 lanes the
    # 43 MB working set is cache-resident and "HBM GB/s" is really fabric / Infinity-Cache bandwidth).
    big = None
    if extras and args.env == "CartPole-v1" and n == (1 << 20):
        try:
            nb, rb, kb = 1 << 27, 2, 20
            with pkg.VectorEnv(args.env, nb, device=dev_index, seed=seed, auto_reset=True, stream=stream.cuda_stream) as e2:
                a2 = torch.empty((rb, nb), dtype=adtype, device=dev)
                for t in range(rb):
                    e2.SampleActionsDevice(a2[t].data_ptr(), seed=seed + 1, tick=t)
                e2.ResetDevice()
                e2.RolloutDevice(a2.data_ptr(), 4, nb, rb)
                e2.Sync()
                brows = [timed_region(lambda: e2.RolloutDevice(a2.data_ptr(), kb, nb, rb)) for _ in range(3)]
                bev = median([r[1] for r in brows])
                bwall = median([r[0] for r in brows])
                bgbps = bytes_per_step * nb / (bev * 1e-3 / kb) / 1e9
                big = {"num_envs": nb, "steps": kb, "repeats": len(brows), "env_steps_per_sec": nb * kb / bwall,
                       "launch_us": bev * 1e3 / kb, "achieved_GBps": bgbps, "frac_of_peak": bgbps / HBM_PEAK_GBPS,
                       "launch_policy": e2.LaunchPolicy(),
                       "note": "working set 5.5 GB per step >> 256 MiB Infinity Cache: real HBM traffic"}
                del a2
        except Exception as e:
            big = {"error": repr(e)[:200]}

    if rank == 0:
        if f64_fig:
            out["cartpole_f64_2p20"] = f64_fig
        if other:
            out["other_configs_2p20"] = other
        out["roofline"]["isolated_launch_us_median"] = single_us
        out["roofline"]["measured_copy_GBps"] = copy_bw
        if copy_bw and copy_bw.get("32MiB"):
            # the fraction of the MEASURED attainable rate beside the fraction of the 8 TB/s spec peak — a plain
            # device-to-device copy of a working set of the same (Infinity-Cache-resident) size on this very box
            out["roofline"]["frac_of_measured_copy_32MiB"] = out["roofline"]["achieved"] / copy_bw["32MiB"]
        if big:
            out["hbm_resident_2p27"] = big
        if fused:
            out["fused_rollout"] = fused
        if host_boundary:
            out["host_boundary"] = host_boundary
        if facade:
            out["single_instance_gpu_facade"] = facade
        if gathered:
            out["with_obs_allgather"] = gathered
    if use_dist:
        # the process group goes away BEFORE the legs below: ranks other than 0 leave (rc 0), so the single-process group
        # leg and the CPU baseline have the GPUs and the host cores to themselves.  Guarded: a teardown that hangs still
        # yields the line (and a non-zero exit).
        watchdog = emit_and_exit_on_timeout("teardown", 60)
        dist.barrier()
        if node_barrier is not None:
            node_barrier.close()
        dist.destroy_process_group()
        watchdog.cancel()
    if rank == 0 and emitted.acquire(blocking=False):
        # the measured headline goes to STDERR now (stdout carries exactly ONE line, at the end): a driver timeout during the
        # legs below — a fresh child process and a CPU run — can no longer lose a measurement that already exists
        sys.stderr.write("[bench] headline so far: " + json.dumps({k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "ms_per_step") if k in out}) + "\n")
        sys.stderr.flush()
        if world > 1:
            time.sleep(1.0)                                      # the other ranks are exiting
            # N = 1 on THIS box, same kernel, same K / repeats protocol without the ranks: value / (N x this) is the weak-scaling
            # efficiency measured inside one run (the driver computes its own from its separate N = 1 run)
            try:
                with pkg.VectorEnv(args.env, n, device=dev_index, seed=seed, auto_reset=True, stream=stream.cuda_stream) as e1:
                    a1 = torch.empty((ring, n), dtype=adtype, device=dev)
                    for t in range(ring):
                        e1.SampleActionsDevice(a1[t].data_ptr(), seed=seed + 1, tick=t)
                    e1.ResetDevice()
                    e1.RolloutDevice(a1.data_ptr(), W, n, ring)
                    torch.cuda.synchronize(dev)
                    ws = []
                    t_end = time.perf_counter() + max(args.min_seconds, 0.05)
                    while len(ws) < 3 or (time.perf_counter() < t_end and len(ws) < MAX_REPEATS):
                        torch.cuda.synchronize(dev)
                        t0 = time.perf_counter()
                        e1.RolloutDevice(a1.data_ptr(), K, n, ring)
                        torch.cuda.synchronize(dev)
                        ws.append(time.perf_counter() - t0)
                    v1 = n * K / median(ws)
                    out["same_box_n1"] = {"value": v1, "unit": "env-steps/s", "ms_per_step": median(ws) * 1e3 / K, "repeats": len(ws),
                                          "scaling_efficiency": out["value"] / (world * v1),
                                          "note": "rank 0's GPU alone after the other ranks exited; efficiency = value / (N x this)"}
                    del a1
            except Exception as e:                               # noqa: BLE001
                out["same_box_n1"] = {"error": repr(e)[:300]}
            if not args.no_group_leg and not gather_in_region:
                out["group_single_process"] = run_group_child(args, world, timeout=180)
        if world == 1 and not args.no_traffic and not args.no_extras and not gather_in_region:
            try:
                # the guide's 2x FETCH_SIZE correction holds for 16-byte-per-lane reads: 4 float / 2 double lanes per thread
                wide16 = launch_policy["envs_per_thread"] * (8 if f64 else 4) == 16
                tr, how = measure_traffic(args, wide16)
                out["roofline"]["traffic_calibrated"] = bool(wide16)
                out["roofline"]["traffic_constant_from_profiles"] = out["roofline"]["traffic"]
                out["roofline"]["traffic"], out["roofline"]["traffic_source"] = tr, how
                out["roofline"]["traffic_over_moved_bytes"] = tr / out["roofline"]["bytes_per_launch"]
            except Exception as e:                               # noqa: BLE001 - the constant (labelled) stays in the line
                out["roofline"]["traffic_measurement_error"] = repr(e)[:300]
        if world == 1 and fused and "us_per_step" in fused and not gather_in_region and args.env == "CartPole-v1":
            # the fused rollouts' roofline objects: VALU issue, measured with one --pmc child pass or — labelled — constants
            try:
                simds = torch.cuda.get_device_properties(dev).multi_processor_count * 4
                valu, src = None, None
                if not args.no_rollout_pmc and not args.no_traffic:
                    try:
                        valu = measure_rollout_valu(args)
                        src = ("measured in this run: rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_TRANS_F32 SQ_BUSY_CYCLES SQ_INSTS_VALU_{FMA,ADD,MUL}_F64, one "
                               f"child pass, {ROLLOUT_CHILD_LAUNCHES} launches of {ROLLOUT_CHILD_STEPS} steps per variant; SQ_INSTS_VALU / SQ_WAVES / lanes per thread / steps")
                    except Exception as e:                       # noqa: BLE001
                        out["fused_rollout"]["valu_measurement_error"] = repr(e)[:300]
                if valu is None:
                    vj = json.load(open(os.path.join(ROOT, "profiles", "rollout_valu.json")))
                    valu = {k: {"valu_per_env_step": v["valu_per_env_step"], "kernel": v.get("kernel"), "counters": {}} for k, v in vj["variants"].items()}
                    src = "NOT measured in this run: constants from profiles/rollout_valu.json — " + vj.get("_source", "")
                targets = [("f32_ring", out["fused_rollout"]), ("f32_sampled", out["fused_rollout"].get("sampled_actions")),
                           ("f32_epsilon_greedy", out["fused_rollout"].get("epsilon_greedy_actions"))]
                f64r = (out.get("cartpole_f64_2p20") or {}).get("fused_rollout")
                if f64r:
                    targets += [("f64_ring", f64r), ("f64_sampled", f64r.get("sampled_actions")), ("f64_epsilon_greedy", f64r.get("epsilon_greedy_actions"))]
                for key, leg in targets:
                    if leg and key in valu and "us_per_step" in leg:
                        cs = valu[key].get("counters") or {}
                        q = None
                        if cs.get("SQ_WAVES") and "SQ_INSTS_VALU_INT64" in cs:
                            per = cs["SQ_WAVES"] * valu[key]["lanes_per_thread"] * valu[key]["steps_per_launch"]
                            q = (cs["SQ_INSTS_VALU_INT64"] / per, cs.get("SQ_INSTS_VALU_TRANS_F32", 0.0) / per,
                                 (cs.get("SQ_INSTS_VALU_FMA_F64", 0.0) + cs.get("SQ_INSTS_VALU_ADD_F64", 0.0) + cs.get("SQ_INSTS_VALU_MUL_F64", 0.0)) / per)
                        leg["roofline"] = valu_roofline(n, valu[key]["valu_per_env_step"], leg["us_per_step"], simds, src, q)
                        leg["roofline"]["kernel"] = valu[key].get("kernel")
                        busy, cycles = valu_busy_in_pass(cs, simds)
                        if busy is not None:
                            leg["roofline"]["valu_busy_in_pmc_pass"] = busy
                            if cs.get("_duration_ns"):
                                leg["roofline"]["clock_GHz_in_pmc_pass"] = cycles / cs["_duration_ns"]
                                leg["roofline"]["us_per_step_in_pmc_pass"] = cs["_duration_ns"] / 1e3 / valu[key]["steps_per_launch"]
                            leg["roofline"]["valu_busy_note"] = ("SQ_INSTS_VALU x 4 / simds / (GRBM_GUI_ACTIVE / XCDs) of the profiled 64-step launches: the VALU pipe's occupancy in the "
                                                                 "chip's own clocks, no assumed frequency (those short launches run at the ramping clock beside it, 2.0-2.3 GHz). "
                                                                 "`frac` prices the floor at the 2.4 GHz peak clock and 4 clocks per counted instruction against the UNPROFILED "
                                                                 "256-step time; both are model fractions good to a few percent (DESIGN.md §5)")
            except Exception as e:                               # noqa: BLE001 - a secondary figure never costs the headline
                out["fused_rollout"]["roofline_error"] = repr(e)[:300]
        if not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(n, args.cpu_seconds)
                if world > 1:
                    out["cpu_baseline"]["when"] = "on rank 0's host after the other ranks had exited (no GPU timing live)"
            except Exception as e:                               # noqa: BLE001 - a secondary figure never costs the headline
                out["cpu_baseline"] = {"error": repr(e)[:300]}
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
