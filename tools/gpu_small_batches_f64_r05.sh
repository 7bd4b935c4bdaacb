#!/bin/bash
# Round 5 probe (runs ON THE GPU BOX): float64 CartPole below 2^20 lanes, one double per thread against two (default policy otherwise).
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for N in 65536 131072 262144 524288; do
  for P in "vec=1" "vec=2,reset_form=0" "vec=2,reset_form=1"; do
    python3 bench.py --no-cpu-baseline --no-extras --no-traffic --dtype f64 --num-envs $N --policy $P 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('CartPole f64  n = %8d  %-20s' % ($N, '$P'), '%-44s' % j['roofline']['kernel'], 'wall us/step %8.3f' % (j['ms_per_step']*1e3))"
  done
done
