#!/bin/bash
# Round 5 probe (runs ON THE GPU BOX): float64 CartPole at batch sizes that are not whole multi-pair groups: one-shot vs 4 pairs per thread.
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for N in "$@"; do
  for P in "sequential_lanes=1" "sequential_lanes=4"; do
    python3 bench.py --no-cpu-baseline --no-extras --no-traffic --dtype f64 --num-envs $N --policy $P 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('n = %8d' % $N, '%-44s' % j['roofline']['kernel'], 'wall us/step %8.3f' % (j['ms_per_step']*1e3), 'per 2^20 lanes %7.3f' % (j['ms_per_step']*1e3 * 1048576 / $N))"
  done
done
