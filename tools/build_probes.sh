#!/bin/bash
# Builds the standalone HIP probes of tools/ (NOT part of the product) into tools/build/ (git-ignored; travels to the GPU box).
#   bash tools/build_probes.sh [skeleton_floor deferred_reset_probe ...]
cd "$(dirname "$0")/.."
mkdir -p tools/build
for P in ${@:-skeleton_floor deferred_reset_probe}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize tools/$P.hip -o tools/build/$P &
done
wait
ls -la tools/build
