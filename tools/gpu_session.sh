#!/bin/bash
# One GPU session of round 5 (overwritten per session; results under gpurun_out/<tag>_*). Usage: bash tools/gpu_session.sh <tag>
# This one (r5d): the environment warp read two levels per lookup (against a build without: plainenv), the speculating 8-wide
# walk (spec8), then the environment-map parity tests and the configured C4 frames (600 block interiors of the whole-frame oracle run).
tag=${1:-r5d}; out=gpurun_out; mkdir -p $out
S="--scenes matball:256,interior:64"
MIW_DEBUG=1 timeout 600 python tools/ab_render.py $S --set "" --set MIW_BVH8=0 --reps 3 > $out/${tag}_head.txt 2> $out/${tag}_head.err; cat $out/${tag}_head.txt; grep "LDS per" $out/${tag}_head.err | sort | uniq | head -3
for v in spec8 plainenv; do
  MIWAVE_LIB_DIR=$PWD/build_exp/$v timeout 600 python tools/ab_render.py $S --set "" --set MIW_BVH8=0 --reps 3 > $out/${tag}_$v.txt 2> $out/${tag}_$v.err; echo "== $v"; cat $out/${tag}_$v.txt; tail -2 $out/${tag}_$v.err | cut -c1-300
done
(timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configured.py -m gpu -x -q -k "environment or envmap or c4" 2>&1 | grep -v "^$" | tail -8) > $out/${tag}_pytest.txt; tail -5 $out/${tag}_pytest.txt
