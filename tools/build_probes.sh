#!/bin/bash
# Builds the standalone HIP probes of tools/ (NOT part of the product) into tools/build/ (git-ignored; travels to the GPU box).
#   bash tools/build_probes.sh [skeleton_floor deferred_reset_probe ...]
cd "$(dirname "$0")/.."
mkdir -p tools/build
for P in ${@:-skeleton_floor deferred_reset_probe store_flavour_probe}; do
  [ "$P" = sc1_store_probe ] && continue
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize tools/$P.hip -o tools/build/$P &
done
wait
# the shipped CartPole kernel with its non-temporal 16-byte stores as shipped / written through (profiles/store_flavour_r05.txt)
if [ $# -eq 0 ] || [[ " $* " == *" sc1_store_probe "* ]]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize tools/sc1_store_probe.hip -o tools/build/sc1_store_probe_nt &
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -DGYMNET_PROBE_STORE_SC1 tools/sc1_store_probe.hip -o tools/build/sc1_store_probe_sc1 &
  wait
fi
ls -la tools/build
