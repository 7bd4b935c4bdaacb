#!/bin/bash
# Round 5 probe (runs ON THE GPU BOX): the non-temporal mask of the CartPole step kernel re-measured by batch size (0 none, 12 action + reward / done,
# 15 every stream), after tools/store_flavour_probe.hip showed that the best store flavour of a pure copy depends on the size.
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for N in "$@"; do
  for P in "nt=0" "nt=12" "nt=15"; do
    python3 bench.py --no-cpu-baseline --no-extras --no-traffic --num-envs $N --policy vec=4,$P 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('n = %9d  %-6s' % ($N, '$P'), '%-44s' % j['roofline']['kernel'], 'wall us/step %8.3f' % (j['ms_per_step']*1e3), 'per 2^20 lanes %7.3f' % (j['ms_per_step']*1e3 * 1048576 / $N))"
  done
done
