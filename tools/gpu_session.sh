#!/bin/bash
# One GPU session of round 6 (overwritten per session; results under gpurun_out/<tag>_*). Usage: bash tools/gpu_session.sh <tag>
# This one (r6f): plan 1's stream kernel over the 8-wide tree, the direct integrator on the phase machine — parity tier, then A/B against the forms they replace
tag=${1:-r6f}; out=$(pwd)/gpurun_out; mkdir -p $out
(timeout 1500 python -m pytest tests -m gpu -x -q -k "not full_frame" 2>&1 | grep -v "^$" | tail -12) > $out/${tag}_pytest_gpu.txt; tail -6 $out/${tag}_pytest_gpu.txt
B="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-live-counters"
line() { name=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift; env "${envs[@]}" timeout 400 $B "$@" > $out/${tag}_${name}.log 2> $out/${tag}_${name}.err; python - $out/${tag}_${name}.log $name <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
    print("%-28s %8.1f Msamples/s  %8.1f ms/frame  kernels %s" % (sys.argv[2], d["value"], d["ms_per_step"], r["kernel_ms"] if r else None))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
line c3_plan1_wide8 -- --scene matball --spp 256 --plan 1
line c3_plan1_bvh2 MIW_BVH8=0 -- --scene matball --spp 256 --plan 1
line c4_plan1_wide8 -- --scene interior --spp 32 --plan 1
line c4_plan1_bvh2 MIW_BVH8=0 -- --scene interior --spp 32 --plan 1
line c3_direct_phased -- --scene matball --spp 256 --integrator direct
line c3_direct_lockstep MIW_PHASED=0 -- --scene matball --spp 256 --integrator direct
line c4_direct_phased -- --scene interior --spp 64 --integrator direct
line c4_direct_lockstep MIW_PHASED=0 -- --scene interior --spp 64 --integrator direct
