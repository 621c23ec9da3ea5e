#!/bin/bash
# One GPU session of round 4 (overwritten per session; results under gpurun_out/<tag>_*). Usage: bash tools/gpu_session.sh <tag>
# This one: MIW_LDS_TABLES (the scene's small tables read from LDS by the packet kernels and the phase machine; in-tree library)
# against the same sources built with -DMIW_LDS_TABLES=0 (build_exp/notab), then the whole GPU tier on the in-tree library.
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
for rep in 1 2; do
  for v in notab -; do
    run c2_${v}_$rep $v -- --steps 3 --warmup 1
    run c3_${v}_$rep $v -- --scene matball --spp 256 --steps 2 --warmup 1
    run c4_${v}_$rep $v -- --scene interior --spp 64 --steps 2 --warmup 1
  done
done
run c5_tree - -- --variant scalar_spectral --scene glassblock --steps 2 --warmup 1
run direct_tree - -- --integrator direct --steps 2 --warmup 1
run direct_notab notab -- --integrator direct --steps 2 --warmup 1
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > $out/${tag}_pytest_gpu.txt; tail -4 $out/${tag}_pytest_gpu.txt
