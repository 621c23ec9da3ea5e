#!/bin/bash
# One GPU session of round 6 (overwritten per session; results under gpurun_out/<tag>_*). Usage: bash tools/gpu_session.sh <tag>
# This one (r6a): first contact of k_path_pooled — smoke, the parity tier's tree tests, A/B against k_path_phased, phase statistics.
tag=${1:-r6a}; out=$(pwd)/gpurun_out; mkdir -p $out
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/${tag}_smoke.txt 2>&1; tail -3 $out/${tag}_smoke.txt
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_xml.py -m gpu -x -q 2>&1 | grep -v "^$" | tail -12) > $out/${tag}_pytest_parity.txt; tail -6 $out/${tag}_pytest_parity.txt
timeout 600 python tools/ab_render.py --scenes matball:128,interior:32 --set "" --set MIW_POOLED=0 --reps 2 > $out/${tag}_ab.txt 2> $out/${tag}_ab.err; cat $out/${tag}_ab.txt; tail -3 $out/${tag}_ab.err
MIWAVE_LIB_DIR=$(pwd)/build_exp/stats MIW_DEBUG=1 timeout 300 python tools/ab_render.py --scenes matball:64,interior:16 --set "" --set MIW_POOLED=0 --reps 1 > $out/${tag}_stats.txt 2>&1; grep "pooled\|phase \|Msamples" $out/${tag}_stats.txt | tail -40
