#!/bin/bash
# Round 6 probe (runs ON THE GPU BOX): why CartPole float32 costs 7.1-7.3 us per 2^20 lanes between 2^21 and 2^22 lanes
# (6.4-6.6 at 2^20, 6.8 at 2^24).  Unprofiled sizes table with the three non-temporal masks, then --pmc passes (no trace flags with
# --pmc; the program after `--` is python3 itself) of the default policy at 2^20 / 2^21 / 2^22 / 2^24 lanes: L2 hit / miss, memory-side
# request counts and stalls, average memory-side latencies (LEVEL / REQ), FETCH_SIZE and WRITE_SIZE in passes of their own.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/trough_r06
rm -rf $O; mkdir -p $O
cd $R
timeout 600 python3 tools/f64_sizes_probe.py f32 1048576,1572864,2097152,3145728,4194304,8388608,16777216 > $O/sizes_f32.txt 2>&1
cd /tmp && export TMPDIR=/tmp
i=0
for C in "TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum" "FETCH_SIZE" "WRITE_SIZE TCC_NORMAL_WRITEBACK_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
         "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" \
         "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C -d $O/pass$i -o pmc -- python3 $R/tools/trough_pmc_child.py > $O/pass$i.log 2>&1
done
# the same first pass with every stream non-temporal (nt = 15) at the trough sizes: does the cacheable state help or hurt there
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum -d $O/nt15 -o pmc -- python3 $R/tools/trough_pmc_child.py 2097152,4194304 nt=15 > $O/nt15.log 2>&1
cd $R
python3 tools/trough_collect.py $O > $O/counters.txt 2>&1
find $O -name "*.db" -delete; find $O -type d -empty -delete
cat $O/sizes_f32.txt; cat $O/counters.txt; tail -3 $O/pass1.log
