#!/bin/bash
# Round 3, GPU session G: waves per SIMD of the phase machine on the interior (the 4-wave variant spills ~80 VGPRs to scratch: 70 GB of writes per frame)
out=gpurun_out; mkdir -p $out
line() {
  label=$1; shift; envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-live-counters "$@" > $out/r3g_$label.log 2> $out/r3g_$label.err
  python - "$out/r3g_$label.log" "$label" <<'P'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-22s %8.1f Msamples/s step %8.2f ms kernels %s" % (sys.argv[2], j["value"], j["ms_per_step"], j["roofline"]["kernel_ms"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
P
}
line c4_w4 -- --scene interior --spp 16
line c4_w3 MIW_PHASED_WAVES=3 -- --scene interior --spp 16
line c4_w3_vote32 MIW_PHASED_WAVES=3 MIW_SHADE_VOTE=2:3 -- --scene interior --spp 16
line c4_w4_vote11 MIW_SHADE_VOTE=1:1 -- --scene interior --spp 16
line c4_w4_vote31 MIW_SHADE_VOTE=1:3 -- --scene interior --spp 16
line c4_lbvh_w3 MIW_PHASED_WAVES=3 -- --scene interior --spp 16 --bvh-quality 0
line c3_w3 -- --scene matball --spp 64
line c3_w4 MIW_PHASED_WAVES=4 -- --scene matball --spp 64
line c3_vote21 MIW_SHADE_VOTE=1:2 -- --scene matball --spp 64
line c3_vote43 MIW_SHADE_VOTE=3:4 -- --scene matball --spp 64
