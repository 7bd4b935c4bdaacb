#!/bin/bash
# Runs ON THE GPU BOX (round 5).  EVERY box that runs this is kept (no selection: VERDICT r4 asked for the median box with the spread,
# not the calmest one): per configuration at 2^20 lanes —
#   * rocprofv3 --kernel-trace --stats of the bench command: the step kernel's AVERAGE DURATION, and from the same timestamped trace
#     the BEGIN-TO-BEGIN SPACING of consecutive launches inside the 4096-launch regions (what the unprofiled wall clock sees);
#   * a separate burst trace with 1024-launch regions (tools/burst_trace_r04.py shape), same two figures;
#   * the same burst with the launches replayed from a captured hipGraph (--policy graph=1): under rocprofv3 every EAGER launch costs
#     the host 8-9 us, more than the short kernels run, so the eager traces show a starved stream; a graph replay is one host call
#     per 1024 launches and the trace shows the GPU's own begin-to-begin spacing;
#   * separate --pmc passes: FETCH_SIZE, WRITE_SIZE (HBM-side traffic; never share a pass) and the SQ counters (VALU per env-step);
#   * the unprofiled bench line.
# --pmc passes carry no trace flags (gpurun refuses the mix); the program after `--` is python3 itself.
# Output: gpurun_out/p5/<tag>/...; tools/collect_profiles_r05.py summarises ON THE BOX into gpurun_out/p5/<tag>/summary/.
#   bash tools/gpu_profile_r05.sh <tag> [configs...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-box}
shift
CFGS=${@:-"CartPole-v1 CartPole-v1-f64 Pendulum-v1 MountainCar-v0 Acrobot-v1"}
O=$R/gpurun_out/p5/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SQC="SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES"
for E in $CFGS; do
  mkdir -p $O/$E
  A="--env $E"
  if [ "$E" = "CartPole-v1-f64" ]; then A="--env CartPole-v1 --dtype f64"; fi
  timeout 300 python3 $R/bench.py --no-cpu-baseline --no-extras --no-traffic $A > $O/$E/unprofiled.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/$E/stats -o s -- python3 $R/bench.py --no-cpu-baseline --no-extras --no-traffic $A > $O/$E/stats.log 2>&1
  timeout 300 rocprofv3 --kernel-trace -d $O/$E/burst -o b -- python3 $R/bench.py --no-cpu-baseline --no-extras --no-traffic $A --steps 1024 --warmup 64 --min-seconds 0.05 > $O/$E/burst.log 2>&1
  timeout 300 rocprofv3 --kernel-trace -d $O/$E/gburst -o g -- python3 $R/bench.py --no-cpu-baseline --no-extras --no-traffic $A --policy graph=1 --ring 256 --steps 1024 --warmup 256 --min-seconds 0.05 > $O/$E/gburst.log 2>&1
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $C -d $O/$E/$C -o pmc -- python3 $R/bench.py --no-cpu-baseline --no-extras --no-traffic $A --no-graph --steps 100 --warmup 10 --min-seconds 0 > $O/$E/$C.log 2>&1
  done
  timeout 300 rocprofv3 --pmc $SQC -d $O/$E/SQ -o pmc -- python3 $R/bench.py --no-cpu-baseline --no-extras --no-traffic $A --no-graph --steps 100 --warmup 10 --min-seconds 0 > $O/$E/SQ.log 2>&1
done
cd $R
rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock" | head -12 > $O/rocminfo.log 2>&1
GYMNET_PROFILE_DIR=p5/$TAG GYMNET_PROFILES_OUT=$O/summary python3 tools/collect_profiles_r05.py $TAG > $O/collect.log 2>&1
for E in $CFGS; do rm -rf $O/$E/stats $O/$E/burst $O/$E/gburst $O/$E/FETCH_SIZE $O/$E/WRITE_SIZE $O/$E/SQ; done
du -sh $O >> $O/collect.log
tail -12 $O/collect.log
