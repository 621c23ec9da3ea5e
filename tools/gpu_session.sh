#!/bin/bash
# One GPU session of round 4 (overwritten per session; results under gpurun_out/<tag>_*). Usage: bash tools/gpu_session.sh <tag>
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
    b = j["config"]["bvh"]
    print("%-30s %9.1f Msamples/s  step %8.2f ms  %s  [%s %.1f ms, depth %s]" % (sys.argv[2], j["value"], j["ms_per_step"], (j.get("roofline") or {}).get("kernel_ms"), b["builder"], b["build_ms"], b["depth"]), flush=True)
except Exception as e:
    print(sys.argv[2], "FAILED", e, flush=True)
P
}
echo "== film replay variants (C2, 3 steps)"
run c2_cols42   - -- --steps 3 --warmup 1
run c2_cols44   - MIW_FILM_COLUMNS=44 -- --steps 3 --warmup 1
run c2_cols82   - MIW_FILM_COLUMNS=82 -- --steps 3 --warmup 1
run c2_groups   - MIW_FILM_COLUMNS=0 -- --steps 3 --warmup 1
echo "== device build timing"
run c4_dev   - MIW_DEBUG=1 -- --scene interior --spp 32 --steps 2 --warmup 1
run c3_dev   - MIW_DEBUG=1 -- --scene matball --spp 256 --steps 2 --warmup 1
grep -h "device builder\|bvh4" $out/${tag}_c3_dev.err $out/${tag}_c4_dev.err | head
echo "== the whole GPU tier"
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
du -sh $out | tail -1
