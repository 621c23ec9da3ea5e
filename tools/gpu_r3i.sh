#!/bin/bash
# Round 3, GPU session I: the direct integrator's kernels (two query sites; 2 waves per SIMD without spills against 3 with, against
# HEAD's single-site loop) and two instruction-scheduler strategies over the whole library (build_exp/<name> from tools/build_ab.sh)
out=gpurun_out; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_direct.py tests/test_gpu_parity.py -m gpu -x -q > $out/r3i_pytest.log 2>&1; echo "pytest rc $?"; tail -3 $out/r3i_pytest.log
line() {
  label=$1; shift; envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-live-counters "$@" > $out/r3i_$label.log 2> $out/r3i_$label.err
  python - "$out/r3i_$label.log" "$label" <<'P'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-26s %8.1f Msamples/s step %8.2f ms kernels %s" % (sys.argv[2], j["value"], j["ms_per_step"], j["roofline"]["kernel_ms"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
P
}
for lib in default direct3 head; do
  L=""; [ $lib != default ] && L="MIWAVE_LIB_DIR=$PWD/build_exp/$lib"
  line direct_c2_$lib $L -- --integrator direct
  line direct_c3_$lib $L -- --integrator direct --scene matball --spp 256
  line direct_c4_$lib $L -- --integrator direct --scene interior --spp 64
done
for lib in default ilp memclause; do
  L=""; [ $lib != default ] && L="MIWAVE_LIB_DIR=$PWD/build_exp/$lib"
  line c2_$lib $L --
  line c3_$lib $L -- --scene matball --spp 128
  line c4_$lib $L -- --scene interior --spp 16
done
line c2_default_again --
