#!/bin/bash
# Runs ON THE GPU BOX (round 3): the bookkeeping (EXTRAS) variants of the CartPole step kernel at 2^20 lanes — HIP-event
# timing of every variant, then per variant a rocprofv3 kernel-trace pass (average kernel duration) and two separate PMC
# passes (FETCH_SIZE, WRITE_SIZE: never in one pass, no trace flags with --pmc).  tools/collect_extras_r03.py summarises.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/x3
mkdir -p $O
cd $R
python3 tools/extras_probe.py > $O/events.log 2>&1
cd /tmp && export TMPDIR=/tmp
for V in lean done_list episode_stats "done_list+stats" all "stats+final_obs(dense)"; do
  D=$O/$(echo "$V" | tr '+()' '___')
  mkdir -p $D
  timeout 300 rocprofv3 --kernel-trace --stats -d $D/stats -o s -- python3 $R/tools/extras_probe.py "$V" 400 > $D/stats.log 2>&1
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $C -d $D/$C -o pmc -- python3 $R/tools/extras_probe.py "$V" 60 > $D/$C.log 2>&1
  done
done
cd $R
python3 tools/collect_extras_r03.py > $O/extras_r03.txt 2> $O/collect.err
cat $O/extras_r03.txt
find $O -name "*.db" -delete
