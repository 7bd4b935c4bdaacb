#!/bin/bash
# Runs ON THE GPU BOX: the four envs at 2^20 lanes through bench.py (unprofiled), two runs each; prints wall us/step and events us/launch.
for E in CartPole-v1 Pendulum-v1 MountainCar-v0 Acrobot-v1; do
  for i in 1 2; do
    python bench.py --no-cpu-baseline --no-extras --env $E "$@" 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('$E', r['kernel'], 'wall_us', round(j['ms_per_step']*1e3,3), 'events_us', round(r['launch_us'],3), 'frac', round(r['frac'],3))"
  done
done
