#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats of the default bench, PMC passes for HBM
# traffic (separate passes, as MI355X_MICROARCH.md prescribes), a 2^27-lane point that cannot be cache
# resident, and the copy-bandwidth probe.  Everything lands in gpurun_out/; summaries are copied to profiles/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_stats -o bench -- python3 $R/bench.py --no-cpu-baseline > $O/prof_stats_bench.log 2>&1
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  tag=$(echo $C | tr ' ' '_')
  timeout 600 rocprofv3 --pmc $C -d $O/pmc_$tag -o pmc -- python3 $R/bench.py --no-cpu-baseline --no-graph --steps 200 --warmup 20 > $O/pmc_$tag.log 2>&1
done
timeout 300 python3 $R/bench.py --no-cpu-baseline --num-envs 134217728 --ring 4 --steps 64 --warmup 8 > $O/bench_2p27.log 2>&1
for E in Pendulum-v1 MountainCar-v0 Acrobot-v1; do
  timeout 300 python3 $R/bench.py --no-cpu-baseline --env $E --steps 1024 --warmup 128 > $O/bench_$E.log 2>&1
done
timeout 300 python3 $R/tools/fused_probe.py > $O/fused_probe.log 2>&1
timeout 300 python3 $R/tools/host_path_probe.py > $O/host_path.log 2>&1
timeout 300 python3 $R/tools/hbm_copy_probe.py > $O/hbm_copy.log 2>&1
find $O -name "*.csv" | head -50 > $O/files.txt
