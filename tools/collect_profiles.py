#!/usr/bin/env python3
"""Turns the rocprofv3 databases that tools/gpu_profile.sh left under gpurun_out/ into the small, tracked
summaries under profiles/ (round-tagged) and refreshes profiles/traffic.json, which bench.py reads for
roofline.traffic.

HBM traffic per launch = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes): on gfx950 FETCH_SIZE reports exactly
half the bytes of a wide coalesced streaming read (MI355X_MICROARCH.md §HBM), collected in SEPARATE --pmc
passes (FETCH_SIZE and WRITE_SIZE do not fit one pass).
"""
import contextlib
import io
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import rocpd_summary  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
os.makedirs(P, exist_ok=True)


def capture(fn, *a):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        fn(*a)
    return buf.getvalue()


def avg_counter(db, counter, kernel="step_kernel"):
    c = sqlite3.connect(db)
    r = c.execute("select avg(value), count(*) from counters_collection where kernel_name like ? and counter_name = ?",
                  (f"%{kernel}%", counter)).fetchone()
    return (r[0], r[1]) if r and r[0] is not None else (None, 0)


out = []
stats_db = os.path.join(G, "prof_stats", "bench_results.db")
if os.path.exists(stats_db):
    out.append("## command: rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline\n")
    out.append(capture(rocpd_summary.stats, stats_db))
    blog = os.path.join(G, "prof_stats_bench.log")
    if os.path.exists(blog):
        for line in open(blog, errors="replace"):
            if line.startswith("{"):
                out.append("## bench.py line printed by the profiled run\n" + line)
open(os.path.join(P, f"rocprof_stats_{tag}.txt"), "w").write("".join(out))

pm = ["## command per pass: rocprofv3 --pmc <COUNTERS> -- python3 bench.py --no-cpu-baseline --no-graph --steps 200 --warmup 20\n"]
vals = {}
for d in sorted(os.listdir(G)):
    db = os.path.join(G, d, "pmc_results.db")
    if d.startswith("pmc_") and os.path.exists(db):
        pm.append(capture(rocpd_summary.pmc, db))
        for cn in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum", "TCC_EA0_RDREQ_sum", "TCC_EA0_WRREQ_sum"):
            v, n = avg_counter(db, cn)
            if v is not None:
                vals[cn] = v
if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
    traffic = (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0
    pm.append(f"\n## HBM-side traffic per step launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB = {traffic:.0f} bytes "
              f"(algorithmic: 41 B x 1048576 = {41 * 1048576} bytes; ratio {traffic / (41 * 1048576):.3f})\n")
    tj = os.path.join(P, "traffic.json")
    cur = json.load(open(tj)) if os.path.exists(tj) else {}
    cur.setdefault("CartPole-v1", {})["1048576"] = traffic
    cur["_source"] = f"profiles/rocprof_pmc_{tag}.txt: (2*FETCH_SIZE + WRITE_SIZE)*1024, separate rocprofv3 --pmc passes"
    json.dump(cur, open(tj, "w"), indent=1)
if "TCC_HIT_sum" in vals:
    pm.append(f"## L2 hit rate = TCC_HIT/(TCC_HIT+TCC_MISS) = {vals['TCC_HIT_sum'] / (vals['TCC_HIT_sum'] + vals['TCC_MISS_sum']):.3f}\n")
open(os.path.join(P, f"rocprof_pmc_{tag}.txt"), "w").write("".join(pm))

misc = []
for f in ("bench.log", "bench_2p27.log", "bench_Pendulum-v1.log", "bench_MountainCar-v0.log", "bench_Acrobot-v1.log",
          "fused_probe.log", "host_path.log", "hbm_copy.log", "rocminfo.log"):
    fp = os.path.join(G, f)
    if os.path.exists(fp):
        misc.append(f"## {f}\n" + "".join(l for l in open(fp, errors="replace") if not l.startswith("/opt/amdgpu")) + "\n")
open(os.path.join(P, f"bench_runs_{tag}.txt"), "w").write("".join(misc))
probe = []
for f in sorted(os.listdir(G)):
    if f.startswith("probe10_") and f.endswith(".log"):
        probe.append(open(os.path.join(G, f)).read() + "\n")
if probe:
    open(os.path.join(P, f"probe_{tag}.txt"), "w").write(
        "## tools/probe_step: kernel variants timed interleaved in one process (HIP events, us per launch)\n" + "".join(probe))
print(open(os.path.join(P, f"rocprof_stats_{tag}.txt")).read())
print(open(os.path.join(P, f"rocprof_pmc_{tag}.txt")).read())
