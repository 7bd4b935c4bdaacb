#!/bin/bash
# Round-2 second pass: Acrobot diet + SLP off.  A/B: libgymnet_amd.so (no SLP) vs libgymnet_amd_slp.so (SLP on).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r2
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests -m gpu -q > $O/pytest_gpu3.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu3.log
for E in CartPole-v1 Pendulum-v1 MountainCar-v0 Acrobot-v1; do
  timeout 300 python3 bench.py --no-cpu-baseline --no-extras --env $E --steps 2048 --warmup 256 > $O/b_noslp_$E.log 2>&1
  GYMNET_LIB_PATH=$R/gym.net_amd/lib/libgymnet_amd_slp.so timeout 300 python3 bench.py --no-cpu-baseline --no-extras --env $E --steps 2048 --warmup 256 > $O/b_slp_$E.log 2>&1
done
timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_k20_b.log 2>&1
timeout 300 python3 bench.py --force-dist --no-cpu-baseline --no-extras --steps 512 --warmup 64 > $O/bench_forcedist_b.log 2>&1
