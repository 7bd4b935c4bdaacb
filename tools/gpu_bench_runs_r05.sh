#!/bin/bash
# Runs ON THE GPU BOX (round 5): the bench lines a reader needs, one box, unprofiled — driver shape, default, every env, float64.
cd ${GRAFT_REPO_ROOT:-$(pwd)}
run() { echo "## python bench.py $*"; python3 bench.py "$@" 2>/dev/null | grep '^{' | tail -1; echo; }
run --gpus 1 --steps 20 --warmup 5
run
run --dtype f64 --no-cpu-baseline
run --env Pendulum-v1 --no-cpu-baseline --no-extras
run --env MountainCar-v0 --no-cpu-baseline --no-extras
run --env Acrobot-v1 --no-cpu-baseline --no-extras
