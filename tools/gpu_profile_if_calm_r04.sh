#!/bin/bash
# Runs ON THE GPU BOX (round 4).  Profiled figures move with the box of the pool (profiles/README.md): this wrapper first takes ONE
# rocprofv3 --kernel-trace --stats pass of the CartPole bench and runs the full profile script (tools/gpu_profile_r04.sh) on THIS box
# only if the profiler's perturbation is small here (step kernel average below the threshold, default 6.70 us); otherwise it says so
# and leaves.  The decision and the probe's figure are logged beside the profiles (gpurun_out/p4/box_probe.log).
R=${GRAFT_REPO_ROOT:-$(pwd)}
THR=${1:-6.70}
mkdir -p $R/gpurun_out/p4
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/calm && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/calm -o s -- python3 $R/bench.py --no-cpu-baseline --no-extras --no-traffic --env CartPole-v1 > /tmp/calm.log 2>&1
DB=$(find /tmp/calm -name "*_results.db" | head -1)
AVG=$(python3 -c "import sqlite3,sys; print('%.3f' % (sqlite3.connect('$DB').execute(\"select avg(duration) from kernels where name like '%step_kernel%'\").fetchone()[0] / 1e3))")
echo "box probe: step_kernel<CartPole,...> average under rocprofv3 --kernel-trace = $AVG us (threshold $THR)" | tee $R/gpurun_out/p4/box_probe.log
if python3 -c "import sys; sys.exit(0 if float('$AVG') < float('$THR') else 1)"; then
  cd $R && bash tools/gpu_profile_r04.sh
  echo "full profile taken on this box" >> $R/gpurun_out/p4/box_probe.log
  tail -9 $R/gpurun_out/p4/collect.log
else
  echo "perturbed box: full profile NOT taken"
fi
