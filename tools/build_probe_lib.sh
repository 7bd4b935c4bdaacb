#!/bin/bash
# Builds a PROBE variant of the library (same sources, extra -D flags) into tools/build/libgymnet_amd_<name>.so for A/B runs on one box:
#   GYMNET_LIB_PATH=tools/build/libgymnet_amd_<name>.so python3 tools/...      (tools/build/ is git-ignored and travels to the GPU box)
#   bash tools/build_probe_lib.sh masked -DGYMNET_PROBE_RESET_MASKED
cd "$(dirname "$0")/.."
NAME=$1; shift
mkdir -p tools/build/obj_$NAME
for S in env_cartpole env_cartpole64 env_acrobot env_pendulum env_mountaincar kernels capi group; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC "$@" -c gym.net_amd/csrc/$S.hip -o tools/build/obj_$NAME/$S.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared tools/build/obj_$NAME/*.o -ldl -o tools/build/libgymnet_amd_$NAME.so && rm -rf tools/build/obj_$NAME
ls -la tools/build/libgymnet_amd_$NAME.so
