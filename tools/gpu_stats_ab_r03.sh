#!/bin/bash
# Round 3: the kernel-trace stats pass of every env, repeated (the profiler's perturbation varies from box to box and run to run:
# profiles/rocprof_stats_cartpole_ab_r02.txt).  Prints, per pass, the step kernel's rocprofv3 average next to the HIP-event
# figure of the SAME profiled run.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/p3ab
mkdir -p $O
: > $O/summary.txt
cd /tmp && export TMPDIR=/tmp
for i in 1 2 3; do
  for E in CartPole-v1 Pendulum-v1 MountainCar-v0 Acrobot-v1; do
    timeout 300 rocprofv3 --kernel-trace --stats -d $O/s_${E}_$i -o s -- python3 $R/bench.py --no-cpu-baseline --no-extras --env $E > $O/s_${E}_$i.log 2>&1
    echo "== $E pass $i: rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-extras --env $E" >> $O/summary.txt
    python3 $R/tools/rocpd_summary.py stats $O/s_${E}_$i/s_results.db | sed -n 2,3p | cut -c1-170 >> $O/summary.txt
    grep -o '"ms_per_step": [0-9.e-]*\|"launch_us": [0-9.]*' $O/s_${E}_$i.log | tr '\n' ' ' >> $O/summary.txt
    echo >> $O/summary.txt
    rm -rf $O/s_${E}_$i
  done
done
cat $O/summary.txt
