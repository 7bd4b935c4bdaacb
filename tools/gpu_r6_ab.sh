#!/bin/bash
# Round 6 A/B on ONE box: the library as built (full-mask reset draw) against the probe variant tools/build/libgymnet_amd_$1.so — the action probe
# (fused rollouts) and the one-step headline of every env, alternating A B A B.
R=${GRAFT_REPO_ROOT:-$(pwd)}
V=${1:-masked}
cd $R
for rep in 1 2; do
  for L in "" "$R/tools/build/libgymnet_amd_$V.so"; do
    echo "=== library: ${L:-default}"
    GYMNET_LIB_PATH=$L python3 tools/rollout_actions_probe.py 2>&1 | grep "float\|stand-alone"
    for A in "--env CartPole-v1" "--env CartPole-v1 --dtype f64" "--env MountainCar-v0"; do
      GYMNET_LIB_PATH=$L python3 bench.py --no-cpu-baseline --no-extras --no-traffic $A 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('   $A', j['roofline']['kernel'], 'events us %.3f' % j['roofline']['launch_us'], 'wall us/step %.3f' % (j['ms_per_step'] * 1e3))"
    done
  done
done
