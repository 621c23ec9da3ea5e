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
echo "== parity: device builder (PLOC) through the fuzz tier, the C3 window, the full-size crops; placed queues"
timeout 900 python -m pytest tests/test_gpu_configured.py -x -q -k "fuzz or c3_window" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "placed_queues" 2>&1 | tail -3
C3="--scene matball --spp 256 --steps 2 --warmup 1"
C4="--scene interior --spp 32 --steps 2 --warmup 1"
echo "== builders"
run c3_sah       - -- $C3
run c3_r03       r03 -- $C3
run c3_ploc      - MIW_DEBUG=1 -- $C3 --bvh-quality 0
run c3_ploc_r8   - MIW_PLOC_RADIUS=8 -- $C3 --bvh-quality 0
run c3_ploc_l4   - MIW_LBVH_LEAF=4 -- $C3 --bvh-quality 0
run c3_lbvh      - MIW_DEVICE_BUILDER=lbvh -- $C3 --bvh-quality 0
run c4_sah       - -- $C4
run c4_r03       r03 -- $C4
run c4_ploc      - MIW_DEBUG=1 -- $C4 --bvh-quality 0
run c4_ploc_r8   - MIW_PLOC_RADIUS=8 -- $C4 --bvh-quality 0
run c4_ploc_r32  - MIW_PLOC_RADIUS=32 -- $C4 --bvh-quality 0
run c4_ploc_l4   - MIW_LBVH_LEAF=4 -- $C4 --bvh-quality 0
run c4_ploc_l1   - MIW_LBVH_LEAF=1 -- $C4 --bvh-quality 0
run c4_lbvh      - MIW_DEVICE_BUILDER=lbvh -- $C4 --bvh-quality 0
grep -h "device builder\|bvh4" $out/${tag}_c3_ploc.err $out/${tag}_c4_ploc.err | head
echo "== 1/8 shards: spread (phase machine default) / contiguous pieces / no placement"
S3="--scene matball --spp 256 --steps 2 --warmup 1 --shard tiles --shard-of 8"
S4="--scene interior --spp 128 --steps 2 --warmup 1 --shard tiles --shard-of 8"
run sh_c3         - -- $S3
run sh_c3_contig  - MIW_PLACE_SPREAD=0 -- $S3
run sh_c3_plain   - MIW_PLACE=0 -- $S3
run sh_c3_noprio  - MIW_TAIL_PRIO=0 -- $S3
run sh_c4         - -- $S4
run sh_c4_contig  - MIW_PLACE_SPREAD=0 -- $S4
run sh_c4_plain   - MIW_PLACE=0 -- $S4
run sh_c4_noprio  - MIW_TAIL_PRIO=0 -- $S4
run sh_c4_m4      - MIW_PLACE_MEASURE=4 -- $S4
run sh_c2         - -- --steps 3 --warmup 1 --shard tiles --shard-of 8
run sh_c2_spread  - MIW_PLACE_SPREAD=1 -- --steps 3 --warmup 1 --shard tiles --shard-of 8
run c3_full256    - -- --scene matball --spp 256 --steps 1 --warmup 1
run c4_full128    - -- --scene interior --spp 128 --steps 1 --warmup 1
run c2_full       - -- --steps 3 --warmup 1
du -sh $out | tail -1
