#!/bin/bash
# Round 6 (runs ON THE GPU BOX): float64 sizes table (looped multi-pair kernel), float32 trough counters, the action probe again.
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
timeout 900 python3 tools/f64_sizes_probe.py > gpurun_out/r6_f64_sizes.txt 2>&1
timeout 600 python3 tools/rollout_actions_probe.py > gpurun_out/r6_actions_probe2.txt 2>&1
timeout 1500 bash tools/gpu_trough_r06.sh > gpurun_out/r6_trough.txt 2>&1
cat gpurun_out/r6_f64_sizes.txt gpurun_out/r6_actions_probe2.txt
tail -60 gpurun_out/r6_trough.txt
