#!/bin/bash
# One GPU session of round 6 (overwritten per session; results under gpurun_out/<tag>_*). Usage: bash tools/gpu_session.sh <tag>
# This one (r6c): k_path_pooled opt-in (MIW_POOLED=1): shapes 12 x 1 and 8 x 2 (two pixels per lane) against k_path_phased; parity of both on the fuzz tier
tag=${1:-r6c}; out=$(pwd)/gpurun_out; mkdir -p $out
(MIW_POOLED=1 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fuzz or tree or both or phased or matball or bvh8" 2>&1 | grep -v "^$" | tail -6) > $out/${tag}_pytest_12x1.txt; tail -3 $out/${tag}_pytest_12x1.txt
(MIW_POOLED=1 MIW_POOL_SHAPE=8x2 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fuzz or tree or both or phased or matball or bvh8" 2>&1 | grep -v "^$" | tail -6) > $out/${tag}_pytest_8x2.txt; tail -3 $out/${tag}_pytest_8x2.txt
timeout 900 python tools/ab_render.py --scenes matball:128 --set "" --set MIW_POOLED=1 --set MIW_POOLED=1,MIW_POOL_SHAPE=8x2 --set MIW_POOLED=1,MIW_POOL_SHAPE=8x2,MIW_POOL_VOTE=56:16:40:20:8 --set MIW_POOLED=1,MIW_POOL_SHAPE=8x2,MIW_POOL_VOTE=48:16:64:32:16 --set MIW_POOLED=1,MIW_POOL_SHAPE=8x2,MIW_POOL_VOTE=48:16:40:20:1 --set MIW_POOLED=1,MIW_POOL_SHAPE=8x2,MIW_POOL_VOTE=32:16:40:20:8 --reps 2 > $out/${tag}_ab.txt 2> $out/${tag}_ab.err; cat $out/${tag}_ab.txt; tail -3 $out/${tag}_ab.err
MIWAVE_LIB_DIR=$(pwd)/build_exp/stats MIW_DEBUG=1 timeout 300 python tools/ab_render.py --scenes matball:64 --set MIW_POOLED=1 --set MIW_POOLED=1,MIW_POOL_SHAPE=8x2 --reps 1 > $out/${tag}_stats.txt 2>&1; grep "pooled\|Msamples" $out/${tag}_stats.txt | tail -24
