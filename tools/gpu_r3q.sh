#!/bin/bash
# Round 3, GPU session Q: the triangle step's updates as selects instead of nested branches (build_exp/prev = the commit before)
out=gpurun_out; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_bvh4.py tests/test_gpu_parity.py -m gpu -x -q > $out/r3q_pytest.log 2>&1; echo "pytest rc $?"; tail -2 $out/r3q_pytest.log
line() {
  label=$1; shift; envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-live-counters "$@" > $out/r3q_$label.log 2> $out/r3q_$label.err
  python - "$out/r3q_$label.log" "$label" <<'P'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-26s %8.1f Msamples/s step %8.2f ms kernels %s" % (sys.argv[2], j["value"], j["ms_per_step"], j["roofline"]["kernel_ms"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
P
}
H="MIWAVE_LIB_DIR=$PWD/build_exp/prev"
for rep in 1 2; do
  line c4_new_$rep -- --scene interior --spp 16
  line c4_prev_$rep $H -- --scene interior --spp 16
  line c3_new_$rep -- --scene matball --spp 128
  line c3_prev_$rep $H -- --scene matball --spp 128
done
line c4lbvh_new -- --scene interior --spp 16 --bvh-quality 0
line c4lbvh_prev $H -- --scene interior --spp 16 --bvh-quality 0
