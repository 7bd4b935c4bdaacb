#!/usr/bin/env python3
"""Summarises the --pmc passes of tools/gpu_trough_r06.sh: per counter and batch size, the average per step-kernel dispatch.  The
dispatches of the batch sizes are told apart by the grid size column when the counters_collection view has one, else by order (the
child runs the sizes one after another, 30 step launches each)."""
import glob
import os
import sqlite3
import sys

out = sys.argv[1]
per = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rows = {}
first = True
for db in sorted(glob.glob(os.path.join(out, "**", "*_results.db"), recursive=True)):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)").fetchall()]
    if first:
        print("# counters_collection columns:", cols)
        first = False
    gcol = next((g for g in ("grid_size_x", "grid_x", "grid_size") if g in cols), None)
    ocol = next((o for o in ("dispatch_id", "id", "start") if o in cols), None)
    tag = os.path.relpath(db, out).split(os.sep)[0]
    q = (f"select {gcol or '0'}, counter_name, value, duration from counters_collection where kernel_name like '%step_kernel%'"
         + (f" order by {ocol}" if ocol else ""))
    seen = {}
    for g, cn, val, dur in c.execute(q).fetchall():
        k = seen.get(cn, 0)
        seen[cn] = k + 1
        key = g if gcol else k // per                      # by grid size, or by position in the run
        e = rows.setdefault((tag, key), {}).setdefault(cn, [0.0, 0, 0.0])
        e[0] += val; e[1] += 1; e[2] += dur
    c.close()
for (tag, g), v in sorted(rows.items(), key=lambda kv: (kv[0][0], kv[0][1] or 0)):
    any_ = next(iter(v.values()))
    print(f"{tag:8s} size-key {g}: " + "  ".join(f"{k}={a / cnt:.1f}" for k, (a, cnt, _) in sorted(v.items()))
          + f"   (dispatches {any_[1]}, avg kernel us under the profiler {any_[2] / any_[1] / 1e3:.2f})")
