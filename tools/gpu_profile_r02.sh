#!/bin/bash
# Runs ON THE GPU BOX (round 2): per env at 2^20 lanes — rocprofv3 kernel-trace stats of the bench, separate PMC passes for
# HBM-side traffic (FETCH_SIZE / WRITE_SIZE never share a pass: MI355X_MICROARCH.md "rocprofv3 PMC slots"), and the SQ
# counters (VALU instructions per wave, wave cycles, waits).  --pmc passes carry no trace flags (gpurun refuses the mix).
# Output: gpurun_out/p2/<env>/{stats,FETCH_SIZE,WRITE_SIZE,SQ}; tools/collect_profiles_r02.py turns it into profiles/*_r02.*
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/p2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for E in CartPole-v1 Pendulum-v1 MountainCar-v0 Acrobot-v1; do
  mkdir -p $O/$E
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/$E/stats -o s -- python3 $R/bench.py --no-cpu-baseline --no-extras --env $E > $O/$E/stats.log 2>&1
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $C -d $O/$E/$C -o pmc -- python3 $R/bench.py --no-cpu-baseline --no-extras --env $E --no-graph --steps 100 --warmup 10 --min-seconds 0 > $O/$E/$C.log 2>&1
  done
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES -d $O/$E/SQ -o pmc -- python3 $R/bench.py --no-cpu-baseline --no-extras --env $E --no-graph --steps 100 --warmup 10 --min-seconds 0 > $O/$E/SQ.log 2>&1
done
# Acrobot's opt-in two-lanes-per-thread packed-FP32 form: kernel time and VALU count, for the record
export GYMNET_VEC=2
mkdir -p $O/Acrobot-v1-packed
timeout 300 rocprofv3 --kernel-trace --stats -d $O/Acrobot-v1-packed/stats -o s -- python3 $R/bench.py --no-cpu-baseline --no-extras --env Acrobot-v1 > $O/Acrobot-v1-packed/stats.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES -d $O/Acrobot-v1-packed/SQ -o pmc -- python3 $R/bench.py --no-cpu-baseline --no-extras --env Acrobot-v1 --no-graph --steps 100 --warmup 10 --min-seconds 0 > $O/Acrobot-v1-packed/SQ.log 2>&1
unset GYMNET_VEC
# the driver-shaped line and the full default line, unprofiled, for the record
cd $R
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_shape.log 2>&1
timeout 300 python3 bench.py > $O/bench_default.log 2>&1
timeout 300 python3 bench.py --force-dist --no-cpu-baseline --no-extras --steps 512 --warmup 64 > $O/bench_forcedist.log 2>&1
timeout 600 python3 bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_gpus2_plain.log 2>&1
timeout 300 python3 tools/hbm_copy_probe.py > $O/hbm_copy.log 2>&1
rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock" | head -12 > $O/rocminfo.log 2>&1
find $O -name "*.db" | sed "s|$O/||" > $O/files.txt
# the rocpd databases are far beyond what may travel back (64 MiB): summarise here, drop them
GYMNET_PROFILES_OUT=$O/summary python3 tools/collect_profiles_r02.py r02 > $O/collect.log 2>&1
for E in CartPole-v1 Pendulum-v1 MountainCar-v0 Acrobot-v1 Acrobot-v1-packed; do rm -rf $O/$E/stats $O/$E/FETCH_SIZE $O/$E/WRITE_SIZE $O/$E/SQ; done
du -sh $O >> $O/collect.log
