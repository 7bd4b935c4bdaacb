#!/bin/bash
# Round 5 probe (runs ON THE GPU BOX): launch forms of the float64 CartPole kernel through bench.py --policy, two runs each.
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for P in "$@"; do
  echo "== policy [$P]"
  for i in 1 2; do python3 bench.py --no-cpu-baseline --no-extras --no-traffic --dtype f64 ${P:+--policy $P} 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('   ', j['roofline']['kernel'], j['config']['launch_policy'], 'wall us/step %.3f' % (j['ms_per_step']*1e3), 'events %.3f' % j['roofline']['launch_us'])"; done
done
