#!/bin/bash
# One GPU session of round 5 (overwritten per session; results under gpurun_out/<tag>_*). Usage: bash tools/gpu_session.sh <tag>
# This one (r5b): the one-process multi-context frame (tests/test_multi_gpu.py), then where the 8-wide walk's time goes:
# -DMIW_PHASE_STATS=1 builds (wave cycles by body) of the 8-wide walk, its speculating variant and the 4-wide walk, and timed A/B runs.
tag=${1:-r5b}; out=gpurun_out; mkdir -p $out
(timeout 600 python -m pytest tests/test_multi_gpu.py tests/test_bvh8.py -m gpu -x -q 2>&1 | tail -8) > $out/${tag}_pytest_multi.txt; tail -3 $out/${tag}_pytest_multi.txt
S="--scenes matball:256,interior:64"
for v in stats8 spec8stats; do
  MIWAVE_LIB_DIR=$PWD/build_exp/$v MIW_DEBUG=1 timeout 600 python tools/ab_render.py $S --set "" --set MIW_BVH8=0 --reps 1 > $out/${tag}_$v.txt 2> $out/${tag}_$v.err
  cat $out/${tag}_$v.txt; grep "^\[ab\]\|phase" $out/${tag}_$v.err | grep -v "rep 0" | awk '/\[ab\]/ {hdr=$0; next} {print hdr " :: " $0}' | grep -v "^$" | head -40
done
MIWAVE_LIB_DIR=$PWD/build_exp/spec8 timeout 600 python tools/ab_render.py $S --set "" --set MIW_BVH8=0 --reps 3 > $out/${tag}_spec8.txt 2> $out/${tag}_spec8.err; cat $out/${tag}_spec8.txt
timeout 600 python tools/ab_render.py $S --set "" --set MIW_BVH8=0 --reps 3 > $out/${tag}_head.txt 2> $out/${tag}_head.err; cat $out/${tag}_head.txt
