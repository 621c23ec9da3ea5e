#!/bin/bash
# Round 3, GPU session O: the shade vote of the phase machine re-tuned for the leaner four-wave kernels
out=gpurun_out; mkdir -p $out
export TMPDIR=/tmp
line() {
  label=$1; shift; envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-live-counters "$@" > $out/r3o_$label.log 2> $out/r3o_$label.err
  python - "$out/r3o_$label.log" "$label" <<'P'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-26s %8.1f Msamples/s step %8.2f ms kernels %s" % (sys.argv[2], j["value"], j["ms_per_step"], j["roofline"]["kernel_ms"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
P
}
line c3_default -- --scene matball --spp 128
for v in 1:1 4:5 3:5 1:2 2:5; do line c3_vote_$v MIW_SHADE_VOTE=$v -- --scene matball --spp 128; done
line c4_default -- --scene interior --spp 16
for v in 1:1 2:3 3:5 2:5 1:3; do line c4_vote_$v MIW_SHADE_VOTE=$v -- --scene interior --spp 16; done
line c3_default_2 -- --scene matball --spp 128
line c4_default_2 -- --scene interior --spp 16
