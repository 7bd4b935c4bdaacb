#!/bin/bash
# Runs ON THE GPU BOX (round 4): per configuration at 2^20 lanes — rocprofv3 kernel-trace stats of the bench, separate PMC passes
# for HBM-side traffic (FETCH_SIZE / WRITE_SIZE never share a pass: MI355X_MICROARCH.md "rocprofv3 PMC slots"), and the SQ
# counters (VALU instructions per wave, wave cycles, waits).  New in round 4: CartPole with the float64-derived done flag (row a3)
# and the GYMNET_FLAG_F64 kernel (73 B per env-step).  --pmc passes carry no trace flags (gpurun refuses the mix); the program
# after `--` is python3 itself.  Output: gpurun_out/p4/...; tools/collect_profiles_r04.py summarises ON THE BOX.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/p4
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SQC="SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES"
for E in CartPole-v1 CartPole-v1-f64 Pendulum-v1 MountainCar-v0 Acrobot-v1; do
  mkdir -p $O/$E
  A="--env $E"
  if [ "$E" = "CartPole-v1-f64" ]; then A="--env CartPole-v1 --dtype f64"; fi
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/$E/stats -o s -- python3 $R/bench.py --no-cpu-baseline --no-extras $A > $O/$E/stats.log 2>&1
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $C -d $O/$E/$C -o pmc -- python3 $R/bench.py --no-cpu-baseline --no-extras $A --no-graph --steps 100 --warmup 10 --min-seconds 0 > $O/$E/$C.log 2>&1
  done
  timeout 300 rocprofv3 --pmc $SQC -d $O/$E/SQ -o pmc -- python3 $R/bench.py --no-cpu-baseline --no-extras $A --no-graph --steps 100 --warmup 10 --min-seconds 0 > $O/$E/SQ.log 2>&1
done
# the driver-shaped line and the full default line, unprofiled, for the record; N = 2 on one GPU is a PLUMBING check (gloo, ranks share the GPU)
cd $R
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_shape.log 2>&1
timeout 300 python3 bench.py > $O/bench_default.log 2>&1
timeout 600 python3 bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_gpus2_plain.log 2>&1
timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_gpus2_launcher.log 2>&1
rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock" | head -12 > $O/rocminfo.log 2>&1
GYMNET_PROFILES_OUT=$O/summary python3 tools/collect_profiles_r04.py r04 > $O/collect.log 2>&1
for E in CartPole-v1 CartPole-v1-f64 Pendulum-v1 MountainCar-v0 Acrobot-v1; do rm -rf $O/$E/stats $O/$E/FETCH_SIZE $O/$E/WRITE_SIZE $O/$E/SQ; done
du -sh $O >> $O/collect.log
