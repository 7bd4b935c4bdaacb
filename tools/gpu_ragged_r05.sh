#!/bin/bash
# Round 5 probe (runs ON THE GPU BOX): what a ragged batch end costs — every env at 2^20 lanes against 2^20 + 1 and 10^6 (default policy).
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for E in "CartPole-v1" "CartPole-v1 --dtype f64" "Pendulum-v1" "MountainCar-v0" "Acrobot-v1"; do
  for N in 1048576 1048577 1000000; do
    python3 bench.py --no-cpu-baseline --no-extras --no-traffic --env $E --num-envs $N 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('%-24s n = %8d' % ('$E', $N), '%-44s' % j['roofline']['kernel'], 'wall us/step %8.3f' % (j['ms_per_step']*1e3), 'per 2^20 lanes %7.3f' % (j['ms_per_step']*1e3 * 1048576 / $N))"
  done
done
