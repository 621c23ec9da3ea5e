#!/bin/bash
# the N-rank bench path (2 / 4 / 8 ranks sharing ONE GPU over gloo: spawn, tile shards, film reduce) on the round's final kernels — bench.py's own spawn and the driver's
# launcher command line (the same torch.distributed.run), alternating: ranks that share a GPU run their kernels against each other, so the lines vary from run to run
tag=${1:-r6x}; out=$(pwd)/gpurun_out; mkdir -p $out
A="--share-gpu --backend gloo --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-live-counters"
for rep in 1 2 3; do
  timeout 300 python bench.py --gpus 2 $A > $out/${tag}_spawn2_$rep.log 2> $out/${tag}_spawn2_$rep.err; tail -1 $out/${tag}_spawn2_$rep.log | cut -c1-330
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29520 + rep)) bench.py --gpus 2 $A > $out/${tag}_launcher2_$rep.log 2> $out/${tag}_launcher2_$rep.err; tail -1 $out/${tag}_launcher2_$rep.log | cut -c1-330
done
for n in 4 8; do timeout 400 python bench.py --gpus $n $A > $out/${tag}_spawn$n.log 2> $out/${tag}_spawn$n.err; tail -1 $out/${tag}_spawn$n.log | cut -c1-330; done
