#!/bin/bash
# Runs ON THE GPU BOX: Acrobot at 2^20 lanes, step_kernel_pipe (shipped) vs the producer / consumer forms (GYMNET_LDS_PIPE=1: 8
# computing waves + 1 storing wave per workgroup, 512-lane tiles; the 4 + 1 / 256-lane shape of profiles/acrobot_lds_r03.txt was removed), bench.py wall / events.
for i in 1 2; do
for F in 0 1; do GYMNET_LDS_PIPE=$F python bench.py --no-cpu-baseline --no-extras --env Acrobot-v1 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('GYMNET_LDS_PIPE=$F', r['kernel'], 'wall_us', round(j['ms_per_step']*1e3,3), 'events_us', round(r['launch_us'],3), 'frac', round(r['frac'],3))"; done
done
