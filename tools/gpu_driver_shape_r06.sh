#!/bin/bash
# Round 6 (runs ON THE GPU BOX): the DRIVER's command under rocprofv3 --kernel-trace --stats (the step kernel's average duration that
# bench.py's roofline.launch_us / frac_by_events must agree with), and the same command unprofiled right after it on the same box.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/driver_shape_r06
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats -o s -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/profiled.log 2> $O/profiled.err
timeout 600 python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/unprofiled.log 2> $O/unprofiled.err
cd $R
{
  echo "# Round 6: rocprofv3 --kernel-trace --stats -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline   (the driver's shape; one unselected box)"
  python3 tools/rocpd_summary.py stats $(find $O/stats -name "*_results.db") | head -40
  for f in profiled unprofiled; do
    python3 - <<PY
import json
for l in open("$O/$f.log"):
    if l.startswith("{"):
        j = json.loads(l); r = j["roofline"]
        print("## $f bench line: value %.4g env-steps/s  ms_per_step %.6f  roofline.frac (wall) %.3f  launch_us (HIP events) %.3f  frac_by_events %.3f  repeats %d  kernel %s" % (j["value"], j["ms_per_step"], r["frac"], r["launch_us"], r["frac_by_events"], j["repeats"], r["kernel"]))
PY
  done
} > $O/summary.txt
rm -rf $O/stats
cat $O/summary.txt | cut -c1-200
