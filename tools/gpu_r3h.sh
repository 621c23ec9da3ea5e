#!/bin/bash
# Round 3, GPU session H: the default bench line with its live PMC passes; the N-rank code path with all ranks on GPU 0 (gloo)
out=gpurun_out; mkdir -p $out
export TMPDIR=/tmp
( time timeout 900 python bench.py > $out/r3h_bench.log 2> $out/r3h_bench.err ) 2>&1 | tail -3
python - $out/r3h_bench.log <<'P'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(j["value"], j["ms_per_step"]); print(json.dumps(j["roofline"])[:1400]); print(json.dumps(j["cpu_baseline"])[:300]); print(list(j["extras"].keys()))
P
for n in 2 8; do
  ( time timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2951$n bench.py --gpus $n --steps 1 --warmup 1 --backend gloo --share-gpu > $out/r3h_ranks$n.log 2> $out/r3h_ranks$n.err ) 2>&1 | tail -3
  tail -1 $out/r3h_ranks$n.log | cut -c1-700; tail -3 $out/r3h_ranks$n.err | cut -c1-300
done
