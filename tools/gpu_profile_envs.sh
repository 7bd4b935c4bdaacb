#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel stats + HBM-traffic PMC passes for the three envs that are not the headline
# (Pendulum / MountainCar / Acrobot at 2^20 lanes), and the SQ-side counters for the CartPole step kernel.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for E in Pendulum-v1 MountainCar-v0 Acrobot-v1; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/env_stats_$E -o s -- python3 $R/bench.py --no-cpu-baseline --env $E --steps 1024 --warmup 128 > $O/env_stats_$E.log 2>&1
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $C -d $O/env_pmc_${E}_$C -o pmc -- python3 $R/bench.py --no-cpu-baseline --env $E --no-graph --steps 100 --warmup 10 > $O/env_pmc_${E}_$C.log 2>&1
  done
done
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES -d $O/pmc_sq -o pmc -- python3 $R/bench.py --no-cpu-baseline --no-graph --steps 200 --warmup 20 > $O/pmc_sq.log 2>&1
