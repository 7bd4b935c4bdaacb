#!/bin/bash
# Round 5 probe (runs ON THE GPU BOX): scalar lanes vs 16-byte lanes (workgroups of 64 / 256) below 2^20 lanes — the launch policy's
# "<= 24 MiB per vector step: scalar lanes" threshold dates from round 1, before the 64-thread workgroups and the wave-compacted reset.
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for E in "CartPole-v1" "MountainCar-v0" "Pendulum-v1"; do
  for N in 65536 131072 262144 524288 786432 1000000; do
    for P in "vec=1" "vec=4,block=64" "vec=4,block=256"; do
      python3 bench.py --no-cpu-baseline --no-extras --no-traffic --env $E --num-envs $N --policy $P 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('%-16s n = %8d  %-16s' % ('$E', $N, '$P'), '%-44s' % j['roofline']['kernel'], 'wall us/step %8.3f' % (j['ms_per_step']*1e3))"
    done
  done
done
