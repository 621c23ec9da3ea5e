#!/bin/bash
# One GPU session of round 5 (overwritten per session; results under gpurun_out/<tag>_*). Usage: bash tools/gpu_session.sh <tag>
# This one (r5c): the triangle step's six loads issued together (against a build without: nofetch2), the speculating 8-wide walk
# (spec8), the sliced SAH decision for huge candidates (MIW_SAH_HUGE=0: round 4's), the builder tests.
tag=${1:-r5c}; out=gpurun_out; mkdir -p $out
S="--scenes matball:256,interior:64"
MIW_DEBUG=1 timeout 600 python tools/ab_render.py $S --set "" --set MIW_BVH8=0 --reps 3 > $out/${tag}_head.txt 2> $out/${tag}_head.err; cat $out/${tag}_head.txt; grep "LDS per\|device builder\|bvh8" $out/${tag}_head.err | sort | uniq | head
for v in spec8 nofetch2; do
  MIWAVE_LIB_DIR=$PWD/build_exp/$v timeout 600 python tools/ab_render.py $S --set "" --set MIW_BVH8=0 --reps 3 > $out/${tag}_$v.txt 2> $out/${tag}_$v.err; echo "== $v"; cat $out/${tag}_$v.txt
done
echo "== MIW_SAH_HUGE=0"; MIW_SAH_HUGE=0 MIW_DEBUG=1 timeout 300 python tools/ab_render.py --scenes interior:16 --reps 1 > $out/${tag}_sah0.txt 2> $out/${tag}_sah0.err; head -1 $out/${tag}_sah0.txt; grep "device builder" $out/${tag}_sah0.err | head -2
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_bvh8.py tests/test_multi_gpu.py -m gpu -x -q -k "builder or bvh8 or contexts" -s 2>&1 | grep -v "^$" | tail -12) > $out/${tag}_pytest.txt; tail -6 $out/${tag}_pytest.txt
