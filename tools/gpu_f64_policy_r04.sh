#!/bin/bash
# Runs ON THE GPU BOX (round 4): the float64 CartPole kernel at 2^20 lanes, one-shot vs the multi-item forms (lane pairs per
# thread), through bench.py --policy.  Two interleaved rounds.
run() { python bench.py --no-cpu-baseline --no-traffic --dtype f64 --policy "$1" 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print('%-28s %-36s wall %.3f us  events %.3f us  frac %.3f' % ('$1', r['kernel'], j['ms_per_step']*1e3, r['launch_us'], r['frac']))"; }
for round in 1 2; do
  for P in "sequential_lanes=1" "sequential_lanes=2" "sequential_lanes=3" "sequential_lanes=4" "sequential_lanes=2,nt=12" "sequential_lanes=1"; do run "$P"; done
done
