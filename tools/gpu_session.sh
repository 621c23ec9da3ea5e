#!/bin/bash
# One GPU session of round 6 (overwritten per session; results under gpurun_out/<tag>_*). Usage: bash tools/gpu_session.sh <tag>
# This one (r6t): with chunk jobs the launch's tail is one chunk long — are the least-progress-first wave priorities (QueueWork::tick, made for the drain of a launch of
# whole-pixel jobs) still worth their instructions on full frames? MIW_TAIL_PRIO=0 against the default, C2 / C3 / C4 / C5 / direct, interleaved
tag=${1:-r6t}; out=$(pwd)/gpurun_out; mkdir -p $out
B="python bench.py --no-cpu-baseline --no-extras --no-live-counters"
line() { name=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift; env "${envs[@]}" timeout 200 $B "$@" > $out/${tag}_${name}.log 2> $out/${tag}_${name}.err; python - $out/${tag}_${name}.log $name <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
    print("%-28s %8.1f Msamples/s  %8.2f ms/frame  parity %s kernels %s" % (sys.argv[2], d["value"], d["ms_per_step"], (d.get("parity") or {}).get("match"), {k: round(v / d["steps"], 2) for k, v in r["kernel_ms"].items()}))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
C3="--scene matball --steps 1 --warmup 1 --spp 1024"; C4="--scene interior --steps 1 --warmup 1 --spp 512"
for rep in 1 2; do
line c2_prio_$rep -- --steps 3 --warmup 1
line c2_noprio_$rep MIW_TAIL_PRIO=0 -- --steps 3 --warmup 1
done
line c3_prio -- $C3
line c3_noprio MIW_TAIL_PRIO=0 -- $C3
line c4_prio -- $C4
line c4_noprio MIW_TAIL_PRIO=0 -- $C4
line c5_prio -- --variant scalar_spectral --scene glassblock --steps 1 --warmup 1
line c5_noprio MIW_TAIL_PRIO=0 -- --variant scalar_spectral --scene glassblock --steps 1 --warmup 1
line direct_prio -- --integrator direct --steps 2 --warmup 1
line direct_noprio MIW_TAIL_PRIO=0 -- --integrator direct --steps 2 --warmup 1
