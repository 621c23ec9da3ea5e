#!/bin/bash
# Round 3, GPU session R: the N-rank code path on the final kernels, all ranks on GPU 0 (gloo): tile shards + reduce
out=gpurun_out; mkdir -p $out
export TMPDIR=/tmp
for n in 2 8; do
  ( time timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2952$n bench.py --gpus $n --steps 1 --warmup 1 --backend gloo --share-gpu --no-cpu-baseline --no-extras --no-live-counters > $out/r3r_ranks$n.log 2> $out/r3r_ranks$n.err ) 2>&1 | tail -3
  tail -1 $out/r3r_ranks$n.log | cut -c1-600; tail -2 $out/r3r_ranks$n.err | cut -c1-200
done
