#!/bin/bash
# One GPU session of round 6 (overwritten per session; results under gpurun_out/<tag>_*). Usage: bash tools/gpu_session.sh <tag>
# This one (r06): the NaN-poison debug build on the parity tier, the section clock of the phase machine's shade body, the round's profile session
# (tools/profile_round.sh: kernel stats, PMC passes incl. plan 1, bench lines, every rank's shard), smoke(), then the whole GPU tier.
tag=${1:-r06}; out=$(pwd)/gpurun_out; mkdir -p $out
rm -rf $out/${tag}_*_pmc[1-4] $out/${tag}_*_trace
(MIWAVE_LIB_DIR=$(pwd)/build_exp/poison timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_film_output.py tests/test_direct.py tests/test_moment.py -m gpu -x -q 2>&1 | grep -v "^$" | tail -8) > $out/${tag}_poison_pytest.txt; tail -3 $out/${tag}_poison_pytest.txt
MIWAVE_LIB_DIR=$(pwd)/build_exp/sections MIW_DEBUG=1 timeout 300 python tools/ab_render.py --scenes matball:64,interior:16 --set "" --reps 1 > $out/${tag}_sections.txt 2>&1; grep "section\|Msamples" $out/${tag}_sections.txt | tail -30
LEAN=1 bash tools/profile_round.sh $tag > $out/${tag}_profile_round.log 2>&1; tail -60 $out/${tag}_profile_round.log
python -c "import __graft_entry__ as g; g.smoke()" > $out/${tag}_smoke.txt 2>&1; tail -3 $out/${tag}_smoke.txt
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^$" | tail -12) > $out/${tag}_pytest_gpu.txt; tail -6 $out/${tag}_pytest_gpu.txt
