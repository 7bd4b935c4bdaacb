for cfg in "256 15" "64 15" "64 12" "128 12" "64 15" "256 15" "64 12"; do set -- $cfg
  GYMNET_BLOCK=$1 GYMNET_NT=$2 python bench.py --no-cpu-baseline --no-extras --env MountainCar-v0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('MountainCar block=$1 nt=$2', 'wall_us', round(j['ms_per_step']*1e3,3), 'events_us', round(r['launch_us'],3))"
done
for cfg in "256 15" "64 15" "256 15" "64 15"; do set -- $cfg
  GYMNET_BLOCK=$1 GYMNET_NT=$2 python bench.py --no-cpu-baseline --no-extras --env Pendulum-v1 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('Pendulum block=$1 nt=$2', 'wall_us', round(j['ms_per_step']*1e3,3), 'events_us', round(r['launch_us'],3))"
done
