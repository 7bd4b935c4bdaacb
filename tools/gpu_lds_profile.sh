#!/bin/bash
# Runs ON THE GPU BOX: Acrobot at 2^20 lanes, shipped step_kernel_pipe vs the producer / consumer form step_kernel_lds
# (GYMNET_LDS_PIPE=1) — unprofiled bench lines, rocprofv3 kernel-trace stats, and the SQ counters of each (separate passes).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/lds
mkdir -p $O
: > $O/summary.txt
cd $R
bash tools/gpu_lds_ab.sh >> $O/summary.txt 2>&1
cd /tmp && export TMPDIR=/tmp
SQC="SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES"
for F in 0 1; do
  export GYMNET_LDS_PIPE=$F
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/s$F -o s -- python3 $R/bench.py --no-cpu-baseline --no-extras --env Acrobot-v1 > $O/s$F.log 2>&1
  echo "== GYMNET_LDS_PIPE=$F: rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-extras --env Acrobot-v1" >> $O/summary.txt
  python3 $R/tools/rocpd_summary.py stats $O/s$F/s_results.db | sed -n 2,3p | cut -c1-170 >> $O/summary.txt
  timeout 120 rocprofv3 --pmc $SQC -d $O/q$F -o pmc -- python3 $R/bench.py --no-cpu-baseline --no-extras --env Acrobot-v1 --no-graph --steps 100 --warmup 10 --min-seconds 0 > $O/q$F.log 2>&1
  echo "== GYMNET_LDS_PIPE=$F: SQ counters per dispatch (quad-cycles for *_CYCLES / WAIT / ACTIVE)" >> $O/summary.txt
  python3 $R/tools/rocpd_summary.py pmc $O/q$F/pmc_results.db step_kernel | cut -c1-118 >> $O/summary.txt
  rm -rf $O/s$F $O/q$F
done
cat $O/summary.txt
