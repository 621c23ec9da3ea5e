#!/bin/bash
tag=${1:-r5m}; out=gpurun_out; mkdir -p $out
MIW_DEBUG=1 MIW_DEBUG_ALLOC=1 timeout 600 python bench.py --no-cpu-baseline --no-live-counters > $out/${tag}_bench.log 2> $out/${tag}_bench.err
grep "build set-up: triangle\|device builder" $out/${tag}_bench.err
python - <<'P'
import json
j=json.loads(open("gpurun_out/r5m_bench.log").read().strip().splitlines()[-1])
print(j["value"], {k:(v["bvh"]["build_ms"], v["bvh"]["first_allocation_after_the_previous_context_ms"]) for k,v in j["extras"].items() if isinstance(v,dict)})
P
