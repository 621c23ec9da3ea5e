#!/bin/bash
# Round 3, GPU session J: the phase machine with its node / triangle steps as shared functions (miw/bvh4.h) against the
# kernels of the commit before (build_exp/head), and the GPU tests that touch what changed
out=gpurun_out; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_bvh4.py tests/test_direct.py tests/test_gpu_fullsize.py -m gpu -x -q > $out/r3j_pytest.log 2>&1; echo "pytest rc $?"; tail -3 $out/r3j_pytest.log
line() {
  label=$1; shift; envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-live-counters "$@" > $out/r3j_$label.log 2> $out/r3j_$label.err
  python - "$out/r3j_$label.log" "$label" <<'P'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-26s %8.1f Msamples/s step %8.2f ms kernels %s" % (sys.argv[2], j["value"], j["ms_per_step"], j["roofline"]["kernel_ms"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
P
}
for rep in 1 2; do
  for lib in default head; do
    L=""; [ $lib != default ] && L="MIWAVE_LIB_DIR=$PWD/build_exp/$lib"
    line c3_${lib}_$rep $L -- --scene matball --spp 128
    line c4_${lib}_$rep $L -- --scene interior --spp 16
    line c4lbvh_${lib}_$rep $L -- --scene interior --spp 16 --bvh-quality 0
  done
done
line direct_c2 -- --integrator direct
