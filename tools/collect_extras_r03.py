#!/usr/bin/env python3
"""Summarises what tools/gpu_extras_r03.sh left under gpurun_out/x3/ (rocpd databases) into profiles/extras_r03.txt."""
import glob
import json
import os
import sqlite3

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out", "x3")
N = 1 << 20
# algorithmic bytes per env-step of each variant: 41 (lean) + what the bookkeeping must move
#   episode statistics: running return + length read and written = 16 B per lane
#   per FINISHED lane (4.5 % of lanes per step): list entry 4 B, return + length 8 B, terminal observation 16 B
P = 0.045
ALGO = {"lean": 41, "done_list": 41 + 4 * P, "episode_stats": 41 + 16 + 8 * P, "done_list_stats": 41 + 16 + 12 * P,
        "all": 41 + 16 + 28 * P, "stats_final_obs_dense_": 41 + 16 + 24 * P}


def db(d, sub):
    f = glob.glob(os.path.join(d, sub, "**", "*.db"), recursive=True)
    return f[0] if f else None


def kernel_avg(path):
    c = sqlite3.connect(path)
    return c.execute("select name, count(*), avg(duration) from kernels where name like '%step_kernel%' group by name order by count(*) desc").fetchone()


def counter_avg(path, counter):
    c = sqlite3.connect(path)
    r = c.execute("select avg(value) from counters_collection where kernel_name like '%step_kernel%' and counter_name = ?", (counter,)).fetchone()
    return r[0] if r else None


print("# CartPole-v1, 2^20 lanes, fused auto-reset: the bookkeeping (EXTRAS) variants of the step kernel  (round 3)")
print("# events = HIP events over 1024 back-to-back launches (tools/extras_probe.py); rocprof = rocprofv3 --kernel-trace --stats average over")
print("# 528 launches; traffic = (2 x FETCH_SIZE + WRITE_SIZE) KiB from SEPARATE --pmc passes (gfx950: FETCH_SIZE counts half the bytes of")
print("# a wide streaming read, MI355X_MICROARCH.md); algorithmic = 41 B + 16 B (running return/length, read + written) + per finished")
print("# lane (4.5 % per step) 4 B list entry, 8 B return + length, 16 B terminal observation.")
ev = {}
p = os.path.join(G, "events.log")
if os.path.exists(p):
    for line in open(p):
        if line.startswith("{\"lean\""):
            ev = json.loads(line)
print(f"{'variant':26} {'events_us':>9} {'rocprof_us':>10} {'algo_MB':>8} {'traffic_MB':>10} {'ratio':>6} {'algo_TB/s':>9}  kernel")
for label, key in (("lean", "lean"), ("done_list", "done_list"), ("episode_stats", "episode_stats"), ("done_list+stats", "done_list_stats"),
                   ("all", "all"), ("stats+final_obs(dense)", "stats_final_obs_dense_")):
    d = os.path.join(G, key)
    sdb, fdb, wdb = db(d, "stats"), db(d, "FETCH_SIZE"), db(d, "WRITE_SIZE")
    k = kernel_avg(sdb) if sdb else None
    f = counter_avg(fdb, "FETCH_SIZE") if fdb else None
    w = counter_avg(wdb, "WRITE_SIZE") if wdb else None
    algo = ALGO[key] * N
    traffic = (2 * f + w) * 1024 if (f is not None and w is not None) else None
    us = k[2] / 1e3 if k else None
    print(f"{label:26} {ev.get(label, {}).get('us_per_step', float('nan')):9.3f} {us if us else float('nan'):10.3f} {algo / 1e6:8.2f} "
          f"{traffic / 1e6 if traffic else float('nan'):10.2f} {traffic / algo if traffic else float('nan'):6.3f} "
          f"{algo / (us * 1e-6) / 1e12 if us else float('nan'):9.2f}  {k[0] if k else ev.get(label, {}).get('kernel')}")
for line in open(p) if os.path.exists(p) else []:
    if line.startswith("{") and not line.startswith("{\"lean\""):
        print("# " + line.strip())
