#!/bin/bash
# Runs ON THE GPU BOX (round 4): Acrobot at 2^20 lanes under every launch form, through `bench.py --policy` (the ABI call — no
# environment variables): one-shot kernel (sequential_lanes=1), the multi-lane kernel with 2..5 lanes per thread (4 = the default
# at this size), the packed two-lane form (vec=2), the producer / consumer form (lds_pipe=1), and the lane-PAIR multi-lane form
# (vec=2 with sequential_lanes=k: step_kernel_pipe2).  Two rounds, interleaved, so that
# box drift shows.  Prints wall us/step, HIP-event us/launch and the moved-bytes fraction.
run() { python bench.py --no-cpu-baseline --no-extras --env Acrobot-v1 --policy "$1" 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print('%-44s %-40s wall %.3f us  events %.3f us  frac %.3f' % ('$1', r['kernel'], j['ms_per_step']*1e3, r['launch_us'], r['frac']))"; }
for round in 1 2; do
  for P in "sequential_lanes=4" "sequential_lanes=1" "sequential_lanes=2" "sequential_lanes=3" "sequential_lanes=5" "sequential_lanes=1,vec=2" "sequential_lanes=4,lds_pipe=1" "vec=2,sequential_lanes=2" "vec=2,sequential_lanes=4" "sequential_lanes=4"; do run "$P"; done
done
