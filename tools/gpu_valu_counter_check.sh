#!/bin/bash
# Round 6 (runs ON THE GPU BOX): does SQ_INSTS_VALU count what we think?  tools/build/issue_rate_probe under --pmc: every k_fma<...> launch
# executes iters x 16 = 65536 v_fma_f32 per wave (+ ~40 of prologue / epilogue); SQ_INSTS_VALU / SQ_WAVES should say so.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/valu_check
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU -d $O/pmc -o pmc -- $R/tools/build/issue_rate_probe > $O/run.log 2>&1
cd $R
python3 - <<PY
import glob, sqlite3
for db in glob.glob("$O/pmc/**/*_results.db", recursive=True):
    c = sqlite3.connect(db)
    rows = c.execute("select dispatch_id, kernel_name, counter_name, sum(value), max(duration) from counters_collection group by dispatch_id, kernel_name, counter_name order by dispatch_id").fetchall()
    by = {}
    for d, k, cn, v, dur in rows:
        by.setdefault(d, {"k": k, "dur": dur})[cn] = v
    for d in sorted(by):
        e = by[d]
        w = e.get("SQ_WAVES", 0) or 1
        print("%-60s waves %6d  VALU/wave %9.1f  INT64/wave %8.1f  TRANS/wave %8.1f  ACTIVE_INST_VALU/wave %9.1f  THREAD_CYCLES_VALU/wave %11.1f  WAVE_CYCLES/wave %9.1f  BUSY_CYCLES %10.0f  dur us %8.1f"
              % (e["k"][:60], w, e.get("SQ_INSTS_VALU", 0) / w, e.get("SQ_INSTS_VALU_INT64", 0) / w, e.get("SQ_INSTS_VALU_TRANS_F32", 0) / w, e.get("SQ_ACTIVE_INST_VALU", 0) / w,
                 e.get("SQ_THREAD_CYCLES_VALU", 0) / w, e.get("SQ_WAVE_CYCLES", 0) / w, e.get("SQ_BUSY_CYCLES", 0), e["dur"] / 1e3))
PY
rm -rf $O/pmc
