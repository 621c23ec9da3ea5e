#!/bin/bash
tag=${1:-r6z}; out=$(pwd)/gpurun_out; mkdir -p $out
sh() { name=$1; shift; (env "$@" 2>&1 | tail -30) > $out/${tag}_$name.log; grep -h "^## \|max-rank" $out/${tag}_$name.log | cut -c1-200; }
sh r2_default timeout 300 python tools/shard_table.py --configs c2 --ranks 2
sh r4_default timeout 300 python tools/shard_table.py --configs c2 --ranks 4
sh r4_jobs MIW_JOB_CHUNK_FORCE=1 timeout 300 python tools/shard_table.py --configs c2 --ranks 4
sh r3_default timeout 300 python tools/shard_table.py --configs c2 --ranks 3
sh r5_jobs MIW_JOB_CHUNK_FORCE=1 timeout 300 python tools/shard_table.py --configs c2 --ranks 5
sh r5_default timeout 300 python tools/shard_table.py --configs c2 --ranks 5
sh r6_jobs MIW_JOB_CHUNK_FORCE=1 timeout 300 python tools/shard_table.py --configs c2 --ranks 6
sh r6_default timeout 300 python tools/shard_table.py --configs c2 --ranks 6
sh c3r4_default timeout 400 python tools/shard_table.py --configs c3 --ranks 4
sh c3r4_jobs MIW_JOB_CHUNK_FORCE=1 timeout 400 python tools/shard_table.py --configs c3 --ranks 4
