#!/bin/bash
# Eight ranks on the ONE GPU of the box: exercises the 8-way node barrier, the 8-peer HIP-IPC mesh, the RCCL
# communicator at world size 8 and the watchdog — not a performance figure (eight processes share one GPU).
mkdir -p gpurun_out/r2
out=gpurun_out/r2/ranks8_on_one_gpu.txt
: > $out
echo "## python3 bench.py --gpus 8 --steps 20 --warmup 5 (self-spawned ranks)" >> $out
( time timeout 900 python3 bench.py --gpus 8 --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/r2/ranks8_a.err | grep '^{' >> $out ) 2>> $out
echo "rc=$?" >> $out
echo "## torch.distributed.run --nproc-per-node 8 bench.py --gpus 8 --steps 20 --warmup 5" >> $out
( time timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29611 \
    bench.py --gpus 8 --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/r2/ranks8_b.err | grep '^{' >> $out ) 2>> $out
echo "rc=$?" >> $out
for f in gpurun_out/r2/ranks8_a.err gpurun_out/r2/ranks8_b.err; do tail -n 5 $f | cut -c1-300; done
cut -c1-1500 $out
