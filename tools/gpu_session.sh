#!/bin/bash
# One GPU session of round 6 (overwritten per session; results under gpurun_out/<tag>_*). Usage: bash tools/gpu_session.sh <tag>
# This one: the round's profile session on the final kernels (tools/profile_round.sh), the smoke entry and the whole GPU tier first
tag=${1:-r06}; out=$(pwd)/gpurun_out; mkdir -p $out
(timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3) > $out/${tag}_smoke.txt; tail -1 $out/${tag}_smoke.txt
(timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^$" | tail -12) > $out/${tag}_pytest_gpu.txt; grep -h "passed\|failed" $out/${tag}_pytest_gpu.txt
LEAN=1 bash tools/profile_round.sh $tag > $out/${tag}_profile_round.log 2>&1; tail -30 $out/${tag}_profile_round.log
