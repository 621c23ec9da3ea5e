#!/bin/bash
# One GPU session of round 5 (overwritten per session; results under gpurun_out/<tag>_*). Usage: bash tools/gpu_session.sh <tag>
# This one (r05): the round's profile session (tools/profile_round.sh: kernel stats, PMC passes, bench lines, every rank's shard),
# smoke(), then the whole GPU tier.
tag=${1:-r05}; out=$(pwd)/gpurun_out; mkdir -p $out
rm -rf $out/${tag}_*_pmc[1-4] $out/${tag}_*_trace          # (a second session under the same tag must not add its counters to the first's)
LEAN=1 bash tools/profile_round.sh $tag > $out/${tag}_profile_round.log 2>&1; tail -75 $out/${tag}_profile_round.log
python -c "import __graft_entry__ as g; g.smoke()" > $out/${tag}_smoke.txt 2>&1; tail -3 $out/${tag}_smoke.txt
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^$" | tail -12) > $out/${tag}_pytest_gpu.txt; tail -6 $out/${tag}_pytest_gpu.txt
