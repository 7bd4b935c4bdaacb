#!/usr/bin/env python3
"""Summarise rocprofv3 (ROCm 7.2, rocpd SQLite output) runs into the small text files kept under profiles/.

    python tools/rocpd_summary.py stats gpurun_out/prof_stats/bench_results.db
    python tools/rocpd_summary.py pmc   gpurun_out/pmc_FETCH_SIZE/pmc_results.db [kernel-substring]
"""
import sqlite3
import sys


def stats(db):
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                     "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"# rocprofv3 --kernel-trace --stats   ({db})")
    print(f"{'calls':>8} {'total_us':>12} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'pct':>6}  kernel")
    for name, n, tot, avg, mn, mx in rows:
        print(f"{n:8d} {tot / 1e3:12.2f} {avg / 1e3:10.3f} {mn / 1e3:10.3f} {mx / 1e3:10.3f} {100 * tot / total:6.2f}  {name}")
    r = c.execute("select vgpr_count, accum_vgpr_count, sgpr_count, grid_x, workgroup_x, lds_size, scratch_size, name from kernels "
                  "where name like '%step_kernel%' limit 1").fetchone()
    if r:
        print(f"# step kernel resources: vgpr={r[0]} agpr={r[1]} sgpr={r[2]} grid_x={r[3]} workgroup_x={r[4]} lds={r[5]} scratch={r[6]}")


def pmc(db, sub="step_kernel"):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, counter_name, count(*), avg(value), min(value), max(value), avg(duration) "
                     "from counters_collection where kernel_name like ? group by kernel_name, counter_name",
                     (f"%{sub}%",)).fetchall()
    print(f"# rocprofv3 --pmc   ({db})   per-dispatch counter values")
    for k, cn, n, avg, mn, mx, dur in rows:
        print(f"{cn:>22} dispatches={n:5d} avg={avg:14.2f} min={mn:14.2f} max={mx:14.2f} avg_kernel_us={dur / 1e3:8.3f}  {k}")


if __name__ == "__main__":
    mode, db = sys.argv[1], sys.argv[2]
    if mode == "stats":
        stats(db)
    else:
        pmc(db, sys.argv[3] if len(sys.argv) > 3 else "step_kernel")
