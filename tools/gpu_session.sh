#!/bin/bash
# One GPU session of round 6 (overwritten per session; results under gpurun_out/<tag>_*). Usage: bash tools/gpu_session.sh <tag>
# This one: the round's profile session on the final kernels (tools/profile_round.sh), the smoke entry and the whole GPU tier first; then the opt-in replay beside the
# path kernel once more (A/B lines), and the device fuzz by hand (tools/fuzz_gpu.py: the scalar restatement live on the box's CPUs)
tag=${1:-r06}; out=$(pwd)/gpurun_out; mkdir -p $out
(timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3) > $out/${tag}_smoke.txt; tail -1 $out/${tag}_smoke.txt
(timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^$" | tail -12) > $out/${tag}_pytest_gpu.txt; grep -h "passed\|failed" $out/${tag}_pytest_gpu.txt
LEAN=1 bash tools/profile_round.sh $tag > $out/${tag}_profile_round.log 2>&1; tail -12 $out/${tag}_profile_round.log
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-live-counters"
line() { name=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift; env "${envs[@]}" timeout 400 $B "$@" > $out/${tag}_${name}.log 2> $out/${tag}_${name}.err; python - $out/${tag}_${name}.log $name <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
    print("%-28s %8.1f Msamples/s  %8.2f ms/frame  overlapped %s parity %s kernels %s" % (sys.argv[2], d["value"], d["ms_per_step"], r.get("film_overlapped"), (d.get("parity") or {}).get("match"), {k: round(v / d["steps"], 2) for k, v in r["kernel_ms"].items()}))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for rep in 1 2 3; do
line ab_c2_plain_$rep --
line ab_c2_beside_$rep MIW_FILM_OVERLAP=1 --
done
(timeout 1500 python tools/fuzz_gpu.py --seeds 600 --first 12000 2>&1 | tail -2) > $out/${tag}_fuzz_a.txt; tail -1 $out/${tag}_fuzz_a.txt
(MIW_FILM_LANES=1 timeout 1500 python tools/fuzz_gpu.py --seeds 400 --first 12600 2>&1 | tail -2) > $out/${tag}_fuzz_b.txt; tail -1 $out/${tag}_fuzz_b.txt
(MIW_FILM_LANES=1 MIW_FILM_OVERLAP=1 timeout 1500 python tools/fuzz_gpu.py --seeds 300 --first 13000 2>&1 | tail -2) > $out/${tag}_fuzz_c.txt; tail -1 $out/${tag}_fuzz_c.txt
(MIW_POOLED=1 timeout 1500 python tools/fuzz_gpu.py --seeds 200 --first 13300 2>&1 | tail -2) > $out/${tag}_fuzz_d.txt; tail -1 $out/${tag}_fuzz_d.txt
(timeout 1500 python tools/fuzz_gpu.py --seeds 200 --first 13500 --variant scalar_spectral 2>&1 | tail -2) > $out/${tag}_fuzz_e.txt; tail -1 $out/${tag}_fuzz_e.txt
