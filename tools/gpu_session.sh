#!/bin/bash
# One GPU session of round 4 (overwritten per session; results under gpurun_out/<tag>_*). Usage: bash tools/gpu_session.sh <tag>
# This one: the environment warp's top levels in LDS (MIW_ENV_TOP=0 switches the staging off at run time) on C4, the packet kernels
# with their triangle records in LDS (C2, C5, direct: against r4f's numbers), the shade vote re-swept on the cheaper shade body,
# then the whole GPU tier.
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
C3="--scene matball --spp 256 --steps 2 --warmup 1"; C4="--scene interior --spp 64 --steps 2 --warmup 1"
for rep in 1 2; do
  run c4_envtop_$rep - -- $C4
  run c4_noenvtop_$rep - MIW_ENV_TOP=0 -- $C4
done
run c2_1 - -- --steps 3 --warmup 1
run c2_2 - -- --steps 3 --warmup 1
run c5 - -- --variant scalar_spectral --scene glassblock --steps 2 --warmup 1
run direct - -- --integrator direct --steps 2 --warmup 1
run c3 - -- $C3
for v in 1:1 3:4 1:2; do run c3_vote_$v - MIW_SHADE_VOTE=$v -- $C3; done
for v in 2:3 1:3 2:5; do run c4_vote_$v - MIW_SHADE_VOTE=$v -- $C4; done
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > $out/${tag}_pytest_gpu.txt; tail -4 $out/${tag}_pytest_gpu.txt
