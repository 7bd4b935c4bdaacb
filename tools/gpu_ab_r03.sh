mkdir -p gpurun_out/r3
for E in Acrobot-v1 Pendulum-v1; do
  for L in "" "gym.net_amd/lib/libgymnet_amd_dedup.so"; do
    for i in 1 2; do
      GYMNET_LIB_PATH=${L:+$PWD/$L} python bench.py --no-cpu-baseline --no-extras --env $E 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$E', '${L:-base}', j['ms_per_step']*1e3, j['roofline']['launch_us'])"
    done
  done
done
for i in 1 2; do GYMNET_RESET_FORM=1 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('CartPole resetform1', j['ms_per_step']*1e3, j['roofline']['launch_us'])"; python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('CartPole base', j['ms_per_step']*1e3, j['roofline']['launch_us'])"; done
python tools/forms_probe.py 2>/dev/null
