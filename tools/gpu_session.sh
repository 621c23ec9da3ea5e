#!/bin/bash
# One GPU session of round 6 (overwritten per session; results under gpurun_out/<tag>_*). Usage: bash tools/gpu_session.sh <tag>
# This one (r6i): the placed launch with ONE host round trip (every rank's C2 shard again), its parity tests, plan 1's refill batch A/B (8 / 16 / 32)
tag=${1:-r6i}; out=$(pwd)/gpurun_out; mkdir -p $out
(timeout 900 python -m pytest tests -m gpu -x -q -k "placed or shard or multi_gpu or options" 2>&1 | grep -v "^$" | tail -6) > $out/${tag}_pytest_placed.txt; grep -h "passed\|failed" $out/${tag}_pytest_placed.txt
timeout 900 python tools/shard_table.py --configs c2 --out $out/${tag}_shards_c2.txt --json $out/${tag}_shards_c2.json > $out/${tag}_shards_c2.log 2>&1; tail -4 $out/${tag}_shards_c2.txt
B="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-live-counters"
line() { name=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift; env "${envs[@]}" timeout 400 $B "$@" > $out/${tag}_${name}.log 2> $out/${tag}_${name}.err; python - $out/${tag}_${name}.log $name <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
    print("%-28s %8.1f Msamples/s  %8.1f ms/frame  kernels %s" % (sys.argv[2], d["value"], d["ms_per_step"], r["kernel_ms"] if r else None))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for rep in 1 2; do
line c3_plan1_batch16_$rep -- --scene matball --spp 256 --plan 1
line c3_plan1_batch8_$rep MIWAVE_LIB_DIR=$(pwd)/build_exp/batch8 -- --scene matball --spp 256 --plan 1
line c3_plan1_batch32_$rep MIWAVE_LIB_DIR=$(pwd)/build_exp/batch32 -- --scene matball --spp 256 --plan 1
done
line c4_plan1_batch16 -- --scene interior --spp 32 --plan 1
line c4_plan1_batch8 MIWAVE_LIB_DIR=$(pwd)/build_exp/batch8 -- --scene interior --spp 32 --plan 1
line c4_plan1_batch32 MIWAVE_LIB_DIR=$(pwd)/build_exp/batch32 -- --scene interior --spp 32 --plan 1
