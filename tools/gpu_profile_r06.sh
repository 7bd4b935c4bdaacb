#!/bin/bash
# Runs ON THE GPU BOX (round 6).  EVERY box that runs this is kept (no selection).  Per box:
#   (a) the five one-step configurations at 2^20 lanes exactly as round 5 measured them (tools/gpu_profile_r05.sh minus the eager burst trace):
#       unprofiled line, rocprofv3 --kernel-trace --stats, hipGraph-replay trace, FETCH_SIZE / WRITE_SIZE / SQ --pmc passes (each its own pass);
#   (b) NEW — the fused rollouts: the default bench line (its fused legs carry their roofline objects, and the 2^27 point),
#       a --kernel-trace --stats pass and an SQ --pmc pass over `bench.py --rollout-child all` (six variants: float32 / float64 x ring /
#       sampled / epsilon-greedy, one warm-up + 3 launches of 64 steps each, told apart by order);
#   (c) NEW — the HBM-resident point: CartPole at 2^27 lanes under --kernel-trace --stats and the two traffic passes.
# --pmc passes carry no trace flags (gpurun refuses the mix); the program after `--` is python3 itself.
#   bash tools/gpu_profile_r06.sh <tag> [configs...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-box}
shift
CFGS=${@:-"CartPole-v1 CartPole-v1-f64 Pendulum-v1 MountainCar-v0 Acrobot-v1"}
O=$R/gpurun_out/p6/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SQC="SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES"
B="python3 $R/bench.py --no-cpu-baseline --no-extras --no-traffic"
for E in $CFGS; do
  mkdir -p $O/$E
  A="--env $E"
  if [ "$E" = "CartPole-v1-f64" ]; then A="--env CartPole-v1 --dtype f64"; fi
  timeout 300 $B $A > $O/$E/unprofiled.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/$E/stats -o s -- $B $A > $O/$E/stats.log 2>&1
  timeout 300 rocprofv3 --kernel-trace -d $O/$E/gburst -o g -- $B $A --policy graph=1 --ring 256 --steps 1024 --warmup 256 --min-seconds 0.05 > $O/$E/gburst.log 2>&1
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $C -d $O/$E/$C -o pmc -- $B $A --no-graph --steps 100 --warmup 10 --min-seconds 0 > $O/$E/$C.log 2>&1
  done
  timeout 300 rocprofv3 --pmc $SQC -d $O/$E/SQ -o pmc -- $B $A --no-graph --steps 100 --warmup 10 --min-seconds 0 > $O/$E/SQ.log 2>&1
done
# (b) fused rollouts
mkdir -p $O/rollout
timeout 600 python3 $R/bench.py > $O/bench_default.log 2> $O/bench_default.err
timeout 300 rocprofv3 --kernel-trace --stats -d $O/rollout/stats -o s -- python3 $R/bench.py --rollout-child all > $O/rollout/stats.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/rollout/SQ -o pmc -- python3 $R/bench.py --rollout-child all > $O/rollout/SQ.log 2>&1
# (c) 2^27 lanes: nothing stays in the 256 MiB Infinity Cache
mkdir -p $O/big
BIG="$B --num-envs 134217728 --ring 2 --min-seconds 0"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/big/stats -o s -- $BIG --steps 20 --warmup 4 > $O/big/stats.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C -d $O/big/$C -o pmc -- $BIG --no-graph --steps 10 --warmup 2 > $O/big/$C.log 2>&1
done
cd $R
rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock" | head -12 > $O/rocminfo.log 2>&1
GYMNET_PROFILE_DIR=p6/$TAG GYMNET_PROFILES_OUT=$O/summary python3 tools/collect_profiles_r05.py $TAG > $O/collect.log 2>&1
python3 tools/collect_rollouts_r06.py $TAG >> $O/collect.log 2>&1
for E in $CFGS; do rm -rf $O/$E/stats $O/$E/gburst $O/$E/FETCH_SIZE $O/$E/WRITE_SIZE $O/$E/SQ; done
rm -rf $O/rollout/stats $O/rollout/SQ $O/big/stats $O/big/FETCH_SIZE $O/big/WRITE_SIZE
du -sh $O >> $O/collect.log
tail -40 $O/collect.log
