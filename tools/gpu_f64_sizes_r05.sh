#!/bin/bash
# Round 5 probe (runs ON THE GPU BOX): float64 CartPole, one-shot kernel against the multi-pair kernel with the deferred reset, by batch size.
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for N in 524288 786432 1048576 1310720 1572864 2097152 4194304 8388608; do
  for P in "sequential_lanes=1" "sequential_lanes=2" "sequential_lanes=4"; do
    python3 bench.py --no-cpu-baseline --no-extras --no-traffic --dtype f64 --num-envs $N --policy $P 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('n = %8d' % $N, '%-44s' % j['roofline']['kernel'], 'wall us/step %8.3f' % (j['ms_per_step']*1e3), 'per 2^20 lanes %7.3f' % (j['ms_per_step']*1e3 * 1048576 / $N))"
  done
done
