#!/bin/bash
# Round 6 (runs ON THE GPU BOX): which cycle counter measures a rollout launch's length?  One --pmc pass over two rollout variants with
# SQ_BUSY_CYCLES (summed over 32 shader engines), GRBM_GUI_ACTIVE (summed over 8 XCDs) and the dispatch's duration: the implied clock and the
# VALU-pipe occupancy SQ_INSTS_VALU x 4 / 1024 SIMDs / cycles by either base.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/bc; timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU -d /tmp/bc -o pmc -- python3 $R/bench.py --rollout-child f32_ring,f32_sampled > /tmp/bc.log 2>&1
python3 - <<PY
import glob, sqlite3
for db in glob.glob("/tmp/bc/**/*_results.db", recursive=True):
    c = sqlite3.connect(db)
    rows = c.execute("select dispatch_id, counter_name, sum(value), count(*), max(duration) from counters_collection where kernel_name like '%rollout_kernel%' group by dispatch_id, counter_name order by dispatch_id").fetchall()
    by = {}
    for d, cn, v, cnt, dur in rows:
        by.setdefault(d, {"dur": dur})[cn] = (v, cnt)
    for d in sorted(by):
        e = by[d]; dur = e["dur"]
        g = lambda k: e[k][0] if k in e else float("nan")
        print("dispatch %d dur %.1f us  INSTS_VALU %.4g (rows %d)  WAVES %.0f  BUSY_CYCLES %.4g -> /32 = %.0f cycles -> %.3f GHz   GRBM_GUI_ACTIVE %.4g (rows %d) -> %.3f GHz  BUSY_CU_CYCLES %.4g  WAVE_CYCLES %.4g ACTIVE_INST_VALU %.4g  | VALU x4/1024/(BUSY/32) = %.3f ; vs GUI_ACTIVE %.3f"
              % (d, dur / 1e3, g("SQ_INSTS_VALU"), e["SQ_INSTS_VALU"][1], g("SQ_WAVES"), g("SQ_BUSY_CYCLES"), g("SQ_BUSY_CYCLES") / 32, g("SQ_BUSY_CYCLES") / 32 / dur,
                 g("GRBM_GUI_ACTIVE"), e.get("GRBM_GUI_ACTIVE", (0, 0))[1], g("GRBM_GUI_ACTIVE") / max(1, e.get("GRBM_GUI_ACTIVE", (0, 1))[1]) / dur, g("SQ_BUSY_CU_CYCLES"), g("SQ_WAVE_CYCLES"), g("SQ_ACTIVE_INST_VALU"),
                 g("SQ_INSTS_VALU") * 4 / 1024 / (g("SQ_BUSY_CYCLES") / 32), g("SQ_INSTS_VALU") * 4 / 1024 / (g("GRBM_GUI_ACTIVE") / max(1, e.get("GRBM_GUI_ACTIVE", (0, 1))[1]))))
PY
