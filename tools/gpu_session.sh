#!/bin/bash
# One GPU session of round 5 (overwritten per session; results under gpurun_out/<tag>_*). Usage: bash tools/gpu_session.sh <tag>
# This one (r5a): the 8-wide tree — its GPU parity tests, A/B against the 4-wide walk on C3 / C4 inside one process, the configured
# frames' digests, then every rank's shard of an 8-GPU frame (tools/shard_table.py).
tag=${1:-r5a}; out=gpurun_out; mkdir -p $out
(timeout 600 python -m pytest tests/test_bvh8.py tests/test_bvh4.py -m gpu -x -q 2>&1 | tail -8) > $out/${tag}_pytest_bvh.txt; tail -3 $out/${tag}_pytest_bvh.txt
MIW_DEBUG=1 timeout 600 python tools/ab_render.py --scenes matball:256,interior:64 --set "" --set MIW_BVH8=0 --reps 3 > $out/${tag}_ab.txt 2> $out/${tag}_ab.err; cat $out/${tag}_ab.txt; grep "bvh8\|device builder" $out/${tag}_ab.err | head
(timeout 900 python -m pytest tests/test_gpu_configured.py -m gpu -x -q 2>&1 | tail -8) > $out/${tag}_pytest_configured.txt; tail -3 $out/${tag}_pytest_configured.txt
timeout 900 python tools/shard_table.py --out $out/${tag}_shards.txt --json $out/${tag}_shards.json > $out/${tag}_shards.log 2>&1; tail -60 $out/${tag}_shards.txt
