#!/bin/bash
# Round 5 probe (runs ON THE GPU BOX): the ISOLATED duration of the CartPole step kernel (rocprofv3 --kernel-trace --stats: eager launches,
# the traced host cannot keep the stream full, every kernel pays its whole ramp and drain) by workgroup size, beside the unprofiled wall clock.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/iso
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for P in "block=256" "block=128" "block=64"; do
  T=$(echo $P | tr '=,' '__')
  timeout 300 python3 $R/bench.py --no-cpu-baseline --no-extras --no-traffic --policy $P > $O/unprofiled_$T.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/$T -o s -- python3 $R/bench.py --no-cpu-baseline --no-extras --no-traffic --policy $P > $O/stats_$T.log 2>&1
  echo "== $P"
  python3 - <<PY
import json, glob, csv
for l in open("$O/unprofiled_$T.log"):
    if l.startswith("{"):
        j = json.loads(l); print("   unprofiled wall %.3f us  events %.3f us  %s" % (j["ms_per_step"] * 1e3, j["roofline"]["launch_us"], j["roofline"]["kernel"]))
import sys
sys.path.insert(0, "$R/tools")
import rocpd_summary
import io, contextlib
for db in glob.glob("$O/$T/**/*_results.db", recursive=True):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        rocpd_summary.stats(db)
    for line in buf.getvalue().split("\n"):
        if "step_kernel<" in line: print("   rocprofv3 (calls total_us avg_us min_us max_us pct):", line[:150])
PY
  rm -rf $O/$T
done
