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
    print("%-30s %9.1f Msamples/s  step %8.2f ms  %s" % (sys.argv[2], j["value"], j["ms_per_step"], (j.get("roofline") or {}).get("kernel_ms")), flush=True)
except Exception as e:
    print(sys.argv[2], "FAILED", e, flush=True)
P
}
echo "== device build timing (per-wave bins, 4-per-thread partition, BIG 2048)"
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configured.py -x -q -k "device_builder or fuzz or c3_window" 2>&1 | tail -3
MIW_DEBUG=1 timeout 300 python - <<'P' 2>&1 | grep -v "amdgpu.ids\|bvh4"
import time
from mitsuba2_amd import api, scenes
api.host_lib()
for name, mk in (("matball", lambda: scenes.cornell_box(1920, 1080, 16, diffuse_only=False, device=-1)), ("interior", lambda: scenes.interior_scene(1920, 1080, 16, device=-1))):
    scene, _ = mk()
    dev = api.Device(0)
    for rep in range(3):
        t0 = time.perf_counter(); dev.upload(scene.desc()); t = (time.perf_counter() - t0) * 1e3
        c = dev.counters()
        print("%s build %d: mi_scene_upload + mi_bvh_build %.1f ms wall, ms_bvh_build %.1f, builder %d, nodes %d depth %d" % (name, rep, t, c.ms_bvh_build, c.bvh_builder, c.bvh_nodes, c.bvh_depth), flush=True)
    dev.close()
P
C3="--scene matball --spp 256 --steps 2 --warmup 1"
C4="--scene interior --spp 32 --steps 2 --warmup 1"
echo "== shade vote num:den (shade once n_shade * num >= lead * den; defaults 2:3, with an environment map 2:4)"
for v in 2:3 1:2 2:5; do run c3_vote_$v - MIW_SHADE_VOTE=$v -- $C3; done
for v in 2:4 2:5 1:3; do run c4_vote_$v - MIW_SHADE_VOTE=$v -- $C4; done
