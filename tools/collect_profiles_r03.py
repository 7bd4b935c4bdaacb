#!/usr/bin/env python3
"""Turns what tools/gpu_profile_r03.sh left under gpurun_out/p3/ into the tracked summaries under profiles/ (tag r03) and
refreshes profiles/traffic.json (bench.py copies the matching entry into roofline.traffic, labelled with its source).

HBM-side traffic per launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB: on gfx950 FETCH_SIZE reports exactly half the bytes of a
wide coalesced streaming read (MI355X_MICROARCH.md §HBM); the two counters come from SEPARATE --pmc passes."""
import contextlib
import io
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import rocpd_summary  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
G = os.path.join(ROOT, "gpurun_out", os.environ.get("GYMNET_PROFILE_DIR", "p3"))
# on the GPU box the databases are too big to travel back (64 MiB cap): summarise THERE into gpurun_out/p2/summary, copy here
P = os.environ.get("GYMNET_PROFILES_OUT") or os.path.join(ROOT, "profiles")
os.makedirs(P, exist_ok=True)
BYTES = {"CartPole-v1": 41, "Pendulum-v1": 37, "MountainCar-v0": 25, "Acrobot-v1": 65}
MOVED = {"CartPole-v1": 41, "Pendulum-v1": 33, "MountainCar-v0": 25, "Acrobot-v1": 57}    # a state row the observation repeats is stored once
LANES = {"CartPole-v1": 4, "Pendulum-v1": 4, "MountainCar-v0": 4, "Acrobot-v1": 4}   # Acrobot at 2^20: step_kernel_pipe, 4 lanes per thread


def capture(fn, *a):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        fn(*a)
    return buf.getvalue()


def avg(db, counter):
    c = sqlite3.connect(db)
    r = c.execute("select avg(value) from counters_collection where kernel_name like '%step_kernel%' and counter_name = ?", (counter,)).fetchone()
    return r[0] if r else None


def bench_line(path):
    if os.path.exists(path):
        for line in open(path, errors="replace"):
            if line.startswith("{"):
                return line
    return None


tj = os.path.join(P, "traffic.json")
src_tj = tj if os.path.exists(tj) else os.path.join(ROOT, "profiles", "traffic.json")
traffic = json.load(open(src_tj)) if os.path.exists(src_tj) else {}
stats_out, pmc_out = [], []
for env, b in BYTES.items():
    d = os.path.join(G, env)
    db = os.path.join(d, "stats", "s_results.db")
    if os.path.exists(db):
        stats_out.append(f"## rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-extras --env {env}\n")
        stats_out.append(capture(rocpd_summary.stats, db))
        line = bench_line(os.path.join(d, "stats.log"))
        if line:
            j = json.loads(line)
            stats_out.append(f"## bench line of that profiled run: ms_per_step {j['ms_per_step']:.6f}  roofline.launch_us {j['roofline']['launch_us']:.3f}  "
                             f"frac {j['roofline']['frac']:.3f}  repeats {j['repeats']}\n\n")
    vals = {}
    for cn in ("FETCH_SIZE", "WRITE_SIZE"):
        pdb = os.path.join(d, cn, "pmc_results.db")
        if os.path.exists(pdb):
            pmc_out.append(f"## {env}: rocprofv3 --pmc {cn} -- python3 bench.py --no-cpu-baseline --no-extras --env {env} --no-graph --steps 100 --warmup 10 --min-seconds 0\n")
            pmc_out.append(capture(rocpd_summary.pmc, pdb))
            vals[cn] = avg(pdb, cn)
    if len(vals) == 2 and None not in vals.values():
        tr = (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024
        traffic.setdefault(env, {})["1048576"] = tr
        pmc_out.append(f"## {env}: HBM-side traffic per launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB = {tr:.0f} B; algorithmic {b} B x 2^20 = {b << 20} B "
                       f"(ratio {tr / (b << 20):.3f}); bytes the kernel really moves {MOVED[env]} B x 2^20 = {MOVED[env] << 20} B (ratio {tr / (MOVED[env] << 20):.3f})\n\n")
    sq = os.path.join(d, "SQ", "pmc_results.db")
    if os.path.exists(sq):
        pmc_out.append(f"## {env}: SQ counters per dispatch (SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count quad-cycles)\n")
        pmc_out.append(capture(rocpd_summary.pmc, sq))
        iv, wv = avg(sq, "SQ_INSTS_VALU"), avg(sq, "SQ_WAVES")
        if iv and wv:
            vec = LANES[env]
            pmc_out.append(f"## {env}: SQ_INSTS_VALU / SQ_WAVES = {iv / wv:.1f} VALU instructions per wave = {iv / wv / vec:.1f} per env-step "
                           f"({vec} env(s) per lane)\n\n")
fdb = os.path.join(G, "fused", "stats", "s_results.db")
if os.path.exists(fdb):
    stats_out.append("## rocprofv3 --kernel-trace --stats -- python3 tools/fused_probe.py   (rollout_kernel: T = 1024 steps per call as 16 launches of 64; avg_us / 64 = us per step)\n")
    stats_out.append(capture(rocpd_summary.stats, fdb))
    line = bench_line(os.path.join(G, "fused", "stats.log"))
    if line:
        stats_out.append("## that run's own figures: " + line + "\n")
fsq = os.path.join(G, "fused", "SQ", "pmc_results.db")
if os.path.exists(fsq):
    pmc_out.append("## fused rollout (rollout_kernel<CartPole,4,true>, 64 steps per launch): SQ counters per dispatch\n")
    pmc_out.append(capture(rocpd_summary.pmc, fsq, "rollout_kernel"))
traffic["_source"] = f"profiles/rocprof_pmc_{tag}.txt: (2*FETCH_SIZE + WRITE_SIZE)*1024, separate rocprofv3 --pmc passes"
json.dump(traffic, open(tj, "w"), indent=1)
open(os.path.join(P, f"rocprof_stats_{tag}.txt"), "w").write("".join(stats_out))
open(os.path.join(P, f"rocprof_pmc_{tag}.txt"), "w").write("".join(pmc_out))
misc = []
for f in ("bench_driver_shape.log", "bench_default.log", "bench_forcedist.log", "bench_gpus2_plain.log", "bench_gpus2_launcher.log", "hbm_copy.log", "rocminfo.log"):
    fp = os.path.join(G, f)
    if os.path.exists(fp):
        keep = [l for l in open(fp, errors="replace") if not l.startswith(("/opt/amdgpu", "RCCL version", "HIP version", "ROCm version", "Hostname", "Librccl"))]
        misc.append(f"## {f}\n" + "".join(keep) + "\n")
open(os.path.join(P, f"bench_runs_{tag}.txt"), "w").write("".join(misc))
print("".join(stats_out))
print("".join(pmc_out))
