#!/bin/bash
# A/B runs of experiment builds (tools/build_variants.py) on the GPU box: for every "name[:ENV=VAL,...]" argument
# one bench line of the material-ball scene (C3 geometry) and of the interior scene (C4 class), short spp
# (throughput is spp-independent). Usage: bash tools/exp_ab.sh <tag> variant1 variant2:MIW_PHASED=0 ...
# A pseudo-variable BENCH="--plan 1" (spaces as +: BENCH=--plan+1) appends arguments to the bench command line.
tag=$1; shift
out=gpurun_out; mkdir -p $out
for spec in "$@"; do
  name=${spec%%:*}; envs=""; [[ "$spec" == *:* ]] && envs=$(echo "${spec#*:}" | tr ',' ' ')
  label=$(echo "$spec" | tr ':=,+' '____')
  bargs=""; for kv in $envs; do [[ "$kv" == BENCH=* ]] && bargs=$(echo "${kv#BENCH=}" | tr '+' ' '); done
  envs=$(for kv in $envs; do [[ "$kv" == BENCH=* ]] || echo -n "$kv "; done)
  for sc in "matball 64" "interior 16"; do
    set -- $sc
    env MIWAVE_LIB_DIR=$PWD/build_exp/$name $envs MIW_DEBUG=1 timeout 300 python bench.py --scene $1 --spp $2 --steps 2 --warmup 1 --no-cpu-baseline $bargs \
        > $out/${tag}_${label}_$1.log 2> $out/${tag}_${label}_$1.err
    python - "$out/${tag}_${label}_$1.log" "$label" "$1" <<'P'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-28s %-9s %8.1f Msamples/s  step %8.1f ms  kernels %s" % (sys.argv[2], sys.argv[3], j["value"], j["ms_per_step"], j["roofline"]["kernel_ms"]))
except Exception as e:
    print(sys.argv[2], sys.argv[3], "FAILED", e)
P
    grep "walk stats" $out/${tag}_${label}_$1.err
  done
done
