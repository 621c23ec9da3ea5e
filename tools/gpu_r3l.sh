#!/bin/bash
# Round 3, GPU session L: loop-invariant uniform values (camera origin, crop window as floats, emitter count) handed to the kernels
# as arguments instead of being computed into vector registers (C2 kernel 12 -> 2 spilled VGPRs, interior kernel 121 -> 75),
# against the kernels of the commit before (build_exp/head)
out=gpurun_out; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_direct.py tests/test_bvh4.py -m gpu -x -q > $out/r3l_pytest.log 2>&1; echo "pytest rc $?"; tail -3 $out/r3l_pytest.log
line() {
  label=$1; shift; envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-live-counters "$@" > $out/r3l_$label.log 2> $out/r3l_$label.err
  python - "$out/r3l_$label.log" "$label" <<'P'
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
  line c2_new_$rep --
  line c2_head_$rep $H --
  line c4_new_$rep -- --scene interior --spp 16
  line c4_head_$rep $H -- --scene interior --spp 16
done
line c3_new -- --scene matball --spp 128
line c3_head $H -- --scene matball --spp 128
line c4lbvh_new -- --scene interior --spp 16 --bvh-quality 0
line c4lbvh_head $H -- --scene interior --spp 16 --bvh-quality 0
line c5_new -- --variant scalar_spectral --scene glassblock
line shard8_new -- --shard tiles --shard-of 8
line shard8_head $H -- --shard tiles --shard-of 8
line direct_c2_new -- --integrator direct
line direct_c4_new -- --integrator direct --scene interior --spp 64
