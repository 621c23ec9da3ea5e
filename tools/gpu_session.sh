#!/bin/bash
# One GPU session of round 6 (overwritten per session; results under gpurun_out/<tag>_*). Usage: bash tools/gpu_session.sh <tag>
# This one (r6j): the placed launch dealt in alternating rounds on the host (no cost copy, no heap): its parity tests, every rank's C2 / C3 shard
tag=${1:-r6j}; out=$(pwd)/gpurun_out; mkdir -p $out
(timeout 900 python -m pytest tests -m gpu -x -q -k "placed or shard or multi_gpu or options" 2>&1 | grep -v "^$" | tail -6) > $out/${tag}_pytest_placed.txt; grep -h "passed\|failed" $out/${tag}_pytest_placed.txt
timeout 900 python tools/shard_table.py --configs c2,c3 --out $out/${tag}_shards.txt --json $out/${tag}_shards.json > $out/${tag}_shards.log 2>&1; grep -h "max-rank\|^floor" $out/${tag}_shards.txt
