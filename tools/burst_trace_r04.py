#!/usr/bin/env python3
"""Round 4 probe: what a 20-launch burst (the driver's `--steps 20` region) looks like on the GPU's own clock.  Reads the rocpd
database of `rocprofv3 --kernel-trace -- python3 bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-traffic
--min-seconds 0.02` and prints, per position in the burst (1st .. 20th launch of a region): the median kernel duration and the
median gap to the previous kernel's end; plus the same for a long (4096-launch) region for comparison.
    python tools/burst_trace_r04.py <results.db> <K>"""
import sqlite3
import sys

db, K = sys.argv[1], int(sys.argv[2])
c = sqlite3.connect(db)
rows = c.execute("select start, end from kernels where name like '%step_kernel%' order by start").fetchall()
# split into bursts: a gap > 30 us starts a new burst
bursts, cur = [], []
for s, e in rows:
    if cur and s - cur[-1][1] > 30_000:
        bursts.append(cur); cur = []
    cur.append((s, e))
if cur:
    bursts.append(cur)
bursts = [b for b in bursts if len(b) == K]
print(f"{len(rows)} step-kernel dispatches, {len(bursts)} bursts of exactly {K}")
import statistics as st
span = [b[-1][1] - b[0][0] for b in bursts]
print(f"burst span (first start -> last end): median {st.median(span) / 1e3:.2f} us = {st.median(span) / 1e3 / K:.3f} us per launch")
show = range(K) if K <= 24 else list(range(4)) + [K // 2] + list(range(K - 3, K))
for i in show:
    d = [b[i][1] - b[i][0] for b in bursts]
    g = [b[i][0] - b[i - 1][1] for b in bursts] if i else [0]
    print(f"  launch {i + 1:4d}: duration median {st.median(d) / 1e3:6.3f} us   gap after previous end {st.median(g) / 1e3:6.3f} us")
