#!/bin/bash
# One GPU session of round 5 (overwritten per session; results under gpurun_out/<tag>_*). Usage: bash tools/gpu_session.sh <tag>
# This one (r5e): the speculating 8-wide walk as the default with its stack sized by the tree's depth (MIW_STACK8_FULL=1: the full
# 16 entries), the shade vote re-swept on it, then the whole GPU tier.
tag=${1:-r5e}; out=gpurun_out; mkdir -p $out
S="--scenes matball:256,interior:64"
MIW_DEBUG=1 timeout 600 python tools/ab_render.py $S --set "" --set MIW_BVH8=0 --set MIW_STACK8_FULL=1 --reps 3 > $out/${tag}_head.txt 2> $out/${tag}_head.err; cat $out/${tag}_head.txt; grep "LDS per" $out/${tag}_head.err | sort | uniq | head -6
timeout 600 python tools/ab_render.py --scenes matball:256 --set "" --set MIW_SHADE_VOTE=1:1 --set MIW_SHADE_VOTE=3:4 --set MIW_SHADE_VOTE=1:2 --set MIW_SHADE_VOTE=2:5 --reps 2 > $out/${tag}_vote3.txt 2> $out/${tag}_vote3.err; cat $out/${tag}_vote3.txt
timeout 600 python tools/ab_render.py --scenes interior:64 --set "" --set MIW_SHADE_VOTE=2:3 --set MIW_SHADE_VOTE=2:5 --set MIW_SHADE_VOTE=1:3 --set MIW_SHADE_VOTE=1:4 --reps 2 > $out/${tag}_vote4.txt 2> $out/${tag}_vote4.err; cat $out/${tag}_vote4.txt
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^$" | tail -12) > $out/${tag}_pytest_gpu.txt; tail -6 $out/${tag}_pytest_gpu.txt
