#!/bin/bash
# Runs ON THE GPU BOX: non-temporal stream masks (0 none, 12 action + reward/done, 15 all) at 2^20 lanes with the round-3 kernels.
for E in CartPole-v1 Pendulum-v1 MountainCar-v0; do
  for NT in 15 12 0 15; do
    GYMNET_NT=$NT python bench.py --no-cpu-baseline --no-extras --env $E 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('$E nt=$NT', r['kernel'], 'wall_us', round(j['ms_per_step']*1e3,3), 'events_us', round(r['launch_us'],3))"
  done
done
for B in 256 128 64; do
  GYMNET_BLOCK=$B python bench.py --no-cpu-baseline --no-extras --env CartPole-v1 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('CartPole block=$B', 'wall_us', round(j['ms_per_step']*1e3,3), 'events_us', round(r['launch_us'],3))"
  GYMNET_BLOCK=$B python bench.py --no-cpu-baseline --no-extras --env MountainCar-v0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('MountainCar block=$B', 'wall_us', round(j['ms_per_step']*1e3,3), 'events_us', round(r['launch_us'],3))"
done
