#!/bin/bash
# A/B of the kernel-trace stats pass: does the bench's HSA_ENABLE_INTERRUPT=0 (polling) interact with rocprofv3?
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/p2ab
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for i in 1 2; do
  for IRQ in 0 1; do
    export GYMNET_BENCH_HSA_INTERRUPT=$IRQ
    timeout 300 rocprofv3 --kernel-trace --stats -d $O/s_${IRQ}_$i -o s -- python3 $R/bench.py --no-cpu-baseline --no-extras > $O/s_${IRQ}_$i.log 2>&1
    echo "== irq=$IRQ run $i" >> $O/summary.txt
    python3 $R/tools/rocpd_summary.py stats $O/s_${IRQ}_$i/s_results.db | head -4 | cut -c1-160 >> $O/summary.txt
    grep -o '"launch_us": [0-9.]*' $O/s_${IRQ}_$i.log >> $O/summary.txt
    rm -rf $O/s_${IRQ}_$i
  done
done
cat $O/summary.txt
