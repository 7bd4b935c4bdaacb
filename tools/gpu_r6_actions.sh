#!/bin/bash
# Round 6 (runs ON THE GPU BOX): action stream v2 — the sampler / rollout parity tests, then the timing probe.
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_gpu_abi_surface.py tests/test_gpu_fused_rollout_ex.py tests/test_gpu_group.py -q -m gpu -x -k "sampling or epsilon or fused_rollout or masked or sampled" 2>&1 | tail -15 > gpurun_out/r6_actions_tests.txt
timeout 600 python tools/rollout_actions_probe.py > gpurun_out/r6_actions_probe.txt 2>&1
cat gpurun_out/r6_actions_tests.txt gpurun_out/r6_actions_probe.txt
