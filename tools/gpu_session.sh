#!/bin/bash
# One GPU session of round 4 (overwritten per session; results under gpurun_out/<tag>_*). Usage: bash tools/gpu_session.sh <tag>
# This one: MIW_PIN_TREE_PTRS=1 (the phase machine's node / triangle pointers kept in registers instead of re-read from the kernel
# arguments on every trip) against the in-tree library on C3 and C4, three runs each, + the tree parity tests on the variant.
tag=${1:-s}; out=gpurun_out; mkdir -p $out
B="--no-cpu-baseline --no-extras --no-live-counters"
run() {  # run <label> <lib dir or -> <env...> -- <bench args...>
  local label=$1 lib=$2; shift 2
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  local libenv=(); [ "$lib" != "-" ] && libenv=(MIWAVE_LIB_DIR=$PWD/build_exp/$lib)
  env "${libenv[@]}" "${envs[@]}" timeout 400 python bench.py $B "$@" > $out/${tag}_$label.log 2> $out/${tag}_$label.err
  python - "$out/${tag}_$label.log" "$label" <<'P'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-30s %9.1f Msamples/s  step %8.2f ms  %s" % (sys.argv[2], j["value"], j["ms_per_step"], (j.get("roofline") or {}).get("kernel_ms")), flush=True)
except Exception as e:
    print(sys.argv[2], "FAILED", e, flush=True)
P
}
for rep in 1 2 3; do
  for v in - pin; do
    run c3_${v}_$rep $v -- --scene matball --spp 256 --steps 2 --warmup 1
    run c4_${v}_$rep $v -- --scene interior --spp 64 --steps 2 --warmup 1
  done
done
MIWAVE_LIB_DIR=$PWD/build_exp/pin timeout 600 python -m pytest tests/test_gpu_configured.py -x -q -k "c3_window or c4_windows" 2>&1 | tail -2
