#!/bin/bash
# Round-2 first GPU pass: the whole -m gpu suite, then bench lines in the driver's shapes.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r2
mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 300 python3 bench.py --steps 20 --warmup 5 > $O/bench_k20.log 2>&1
GYMNET_BENCH_HSA_INTERRUPT=1 timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_k20_irq.log 2>&1
timeout 300 python3 bench.py --no-cpu-baseline > $O/bench_default.log 2>&1
timeout 300 python3 bench.py --force-dist --no-cpu-baseline --steps 512 --warmup 64 > $O/bench_forcedist.log 2>&1
timeout 300 python3 bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_gpus2_plain.log 2>&1
for E in Pendulum-v1 MountainCar-v0 Acrobot-v1; do
  timeout 300 python3 bench.py --no-cpu-baseline --no-extras --env $E --steps 1024 --warmup 128 > $O/bench_$E.log 2>&1
done
rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock" | head -12 > $O/rocminfo.log 2>&1
