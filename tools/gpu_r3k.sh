#!/bin/bash
# Round 3, GPU session K: a fifth wavefront per SIMD — k_path_phased at 96 VGPRs (five workgroups' stacks need 28-entry columns:
# build_exp/w5, against the same columns at four waves: build_exp/s28) and the Cornell packet kernel at 96 VGPRs (build_exp/c2w5)
out=gpurun_out; mkdir -p $out
export TMPDIR=/tmp
line() {
  label=$1; shift; envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-live-counters "$@" > $out/r3k_$label.log 2> $out/r3k_$label.err
  python - "$out/r3k_$label.log" "$label" <<'P'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-26s %8.1f Msamples/s step %8.2f ms kernels %s" % (sys.argv[2], j["value"], j["ms_per_step"], j["roofline"]["kernel_ms"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
P
}
W5="MIWAVE_LIB_DIR=$PWD/build_exp/w5"; S28="MIWAVE_LIB_DIR=$PWD/build_exp/s28"; C2W5="MIWAVE_LIB_DIR=$PWD/build_exp/c2w5"
line c4_default -- --scene interior --spp 16
line c4_s28 $S28 -- --scene interior --spp 16
line c4_w5 $W5 MIW_PHASED_WAVES=5 -- --scene interior --spp 16
line c4_w5_vote $W5 MIW_PHASED_WAVES=5 MIW_SHADE_VOTE=1:1 -- --scene interior --spp 16
line c4lbvh_default -- --scene interior --spp 16 --bvh-quality 0
line c4lbvh_w5 $W5 MIW_PHASED_WAVES=5 -- --scene interior --spp 16 --bvh-quality 0
line c3_default -- --scene matball --spp 128
line c3_w4 MIW_PHASED_WAVES=4 -- --scene matball --spp 128
line c3_w5 $W5 MIW_PHASED_WAVES=5 -- --scene matball --spp 128
line c2_default --
line c2_w5 $C2W5 MIW_WG_PER_CU=5 --
line c2_w5_wg4 $C2W5 --
line c2_default_2 --
line c4_default_2 -- --scene interior --spp 16
line c4_w5_2 $W5 MIW_PHASED_WAVES=5 -- --scene interior --spp 16
