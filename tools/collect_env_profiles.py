#!/usr/bin/env python3
"""Summarises what tools/gpu_profile_envs.sh left under gpurun_out/ into profiles/rocprof_envs_<tag>.txt and adds the
per-launch HBM traffic of the three extra envs to profiles/traffic.json (same 2*FETCH_SIZE + WRITE_SIZE rule)."""
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
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
BYTES = {"Pendulum-v1": 37, "MountainCar-v0": 25, "Acrobot-v1": 65}


def capture(fn, *a):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        fn(*a)
    return buf.getvalue()


def avg(db, counter):
    c = sqlite3.connect(db)
    r = c.execute("select avg(value) from counters_collection where kernel_name like '%step_kernel%' and counter_name = ?", (counter,)).fetchone()
    return r[0] if r else None


out = []
tj = os.path.join(P, "traffic.json")
traffic = json.load(open(tj)) if os.path.exists(tj) else {}
for env, b in BYTES.items():
    db = os.path.join(G, f"env_stats_{env}", "s_results.db")
    if os.path.exists(db):
        out.append(f"## rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --env {env} --steps 1024 --warmup 128\n")
        out.append(capture(rocpd_summary.stats, db))
    vals = {}
    for cn in ("FETCH_SIZE", "WRITE_SIZE"):
        pdb = os.path.join(G, f"env_pmc_{env}_{cn}", "pmc_results.db")
        if os.path.exists(pdb):
            out.append(capture(rocpd_summary.pmc, pdb))
            vals[cn] = avg(pdb, cn)
    if len(vals) == 2 and None not in vals.values():
        tr = (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024
        traffic.setdefault(env, {})["1048576"] = tr
        out.append(f"## {env}: HBM-side traffic per launch = {tr:.0f} B; algorithmic {b} B x 2^20 = {b << 20} B; ratio {tr / (b << 20):.3f}\n\n")
sq = os.path.join(G, "pmc_sq", "pmc_results.db")
if os.path.exists(sq):
    out.append("## CartPole step kernel, SQ counters (rocprofv3 --pmc SQ_*, eager launches, per dispatch of 4096 waves)\n")
    out.append(capture(rocpd_summary.pmc, sq))
json.dump(traffic, open(tj, "w"), indent=1)
open(os.path.join(P, f"rocprof_envs_{tag}.txt"), "w").write("".join(out))
print("".join(out))
