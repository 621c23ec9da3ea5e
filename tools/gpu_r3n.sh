#!/bin/bash
# Round 3, GPU session N: QueueWork without the placed-queue machinery in the tree kernels (interior kernel 64 -> 20 spilled VGPRs)
# mint of the walk taken from the path state, the best hit's primitive id looked up on ties): interior kernel 76 -> 64 spilled VGPRs
out=gpurun_out; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_bvh4.py tests/test_gpu_parity.py tests/test_gpu_configured.py -m gpu -x -q > $out/r3n_pytest.log 2>&1; echo "pytest rc $?"; tail -3 $out/r3n_pytest.log
line() {
  label=$1; shift; envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-live-counters "$@" > $out/r3n_$label.log 2> $out/r3n_$label.err
  python - "$out/r3n_$label.log" "$label" <<'P'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-26s %8.1f Msamples/s step %8.2f ms kernels %s" % (sys.argv[2], j["value"], j["ms_per_step"], j["roofline"]["kernel_ms"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
P
}
H="MIWAVE_LIB_DIR=$PWD/build_exp/head"
for rep in 1 2; do
  line c4_new_$rep -- --scene interior --spp 16
  line c4_head_$rep $H -- --scene interior --spp 16
  line c3_new_$rep -- --scene matball --spp 128
  line c3_head_$rep $H -- --scene matball --spp 128
done
line c4lbvh_new -- --scene interior --spp 16 --bvh-quality 0
line c3lbvh_new -- --scene matball --spp 128 --bvh-quality 0
line c4_w3_new MIW_PHASED_WAVES=3 -- --scene interior --spp 16
line c3_w4_new MIW_PHASED_WAVES=4 -- --scene matball --spp 128
line c2_new --
line c3_w3_new MIW_PHASED_WAVES=3 -- --scene matball --spp 128
line tess3_w3 MIW_PHASED_WAVES=3 -- --scene matball --tess 3 --spp 128
line tess3_w4 MIW_PHASED_WAVES=4 -- --scene matball --tess 3 --spp 128
line tess1_w3 MIW_PHASED_WAVES=3 -- --scene matball --tess 1 --spp 128
line tess1_w4 MIW_PHASED_WAVES=4 -- --scene matball --tess 1 --spp 128
line c3_shard8 -- --scene matball --spp 256 --shard tiles --shard-of 8
line c3_shard8_head $H -- --scene matball --spp 256 --shard tiles --shard-of 8
line shard8_c2 -- --shard tiles --shard-of 8
