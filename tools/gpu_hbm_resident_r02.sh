#!/bin/bash
# HBM-resident (2^27 lanes: working set >> the 256 MiB Infinity Cache) figure for every env, through bench.py.
mkdir -p gpurun_out/r2
out=gpurun_out/r2/hbm_resident_all_envs.txt
: > $out
for env in CartPole-v1 Pendulum-v1 MountainCar-v0 Acrobot-v1; do
  echo "## $env 2^27 lanes" >> $out
  timeout 600 python3 bench.py --env $env --num-envs 134217728 --ring 4 --steps 64 --warmup 8 \
      --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' >> $out
done
python3 - <<'PY'
import json
for line in open("gpurun_out/r2/hbm_resident_all_envs.txt"):
    if line.startswith("{"):
        j = json.loads(line)
        r = j["roofline"]
        print(j["config"]["workload"][:60], "ms/step %.4f" % j["ms_per_step"], "value %.3e" % j["value"], "achieved %.0f GB/s frac %.3f" % (r["achieved"], r["frac"]))
PY
