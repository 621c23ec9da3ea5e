#!/bin/bash
# Round 3, GPU session D: GPU test tier with priorities always on, placed per-SIMD queues for small shards, LBVH leaf 2; A/B lines.
out=gpurun_out; mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=6 > $out/r3d_pytest.log 2>&1; echo "pytest rc $?" >> $out/r3d_pytest.log
tail -4 $out/r3d_pytest.log; grep -E "^(FAILED|ERROR)" $out/r3d_pytest.log | head -20
line() {   # line <label> <env...> -- <bench args>
  label=$1; shift; envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs MIW_DEBUG=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-live-counters "$@" > $out/r3d_$label.log 2> $out/r3d_$label.err
  python - "$out/r3d_$label.log" "$label" <<'P'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-22s %8.1f Msamples/s step %8.2f ms bvh %7.1f ms (%s) kernels %s" % (sys.argv[2], j["value"], j["ms_per_step"], j["config"]["bvh"]["build_ms"], j["config"]["bvh"]["builder"][:6], j["roofline"]["kernel_ms"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
P
}
line base --
line base_noprio MIW_TAIL_PRIO=0 --
line shard8 -- --shard tiles --shard-of 8
line shard8_noplace MIW_PLACE=0 -- --shard tiles --shard-of 8
line shard8_noprio MIW_TAIL_PRIO=0 MIW_PLACE=0 -- --shard tiles --shard-of 8
line shard4 -- --shard tiles --shard-of 4
line shard2 -- --shard tiles --shard-of 2
line c3 -- --scene matball --spp 64
line c3_noprio MIW_TAIL_PRIO=0 -- --scene matball --spp 64
line c3_lbvh -- --scene matball --spp 64 --bvh-quality 0
line c3_shard8 -- --scene matball --spp 256 --shard tiles --shard-of 8
line c3_shard8_noplace MIW_PLACE=0 -- --scene matball --spp 256 --shard tiles --shard-of 8
line c3_1024 -- --scene matball --spp 1024 --steps 1
line c4 -- --scene interior --spp 16
line c4_noprio MIW_TAIL_PRIO=0 -- --scene interior --spp 16
line c4_lbvh -- --scene interior --spp 16 --bvh-quality 0
line c4_64 -- --scene interior --spp 64 --steps 1
line c5 -- --variant scalar_spectral --scene glassblock
