#!/bin/bash
# extended device fuzz on the round's final kernels (every run under its own timeout)
tag=${1:-r6w}; out=$(pwd)/gpurun_out; mkdir -p $out
f() { name=$1; shift; (env "$@" 2>&1 | tail -2) > $out/${tag}_fuzz_$name.txt; echo "$name: $(tail -1 $out/${tag}_fuzz_$name.txt)"; }
f a timeout 500 python tools/fuzz_gpu.py --seeds 700 --first 16100
f b MIW_JOB_CHUNK_FORCE=1 MIW_JOB_CHUNK=1 timeout 500 python tools/fuzz_gpu.py --seeds 400 --first 16800
f c MIW_JOB_CHUNK_FORCE=1 MIW_JOB_CHUNK=8 timeout 500 python tools/fuzz_gpu.py --seeds 400 --first 17200
f d MIW_JOB_CHUNK_FORCE=1 MIW_JOB_CHUNK=2 timeout 500 python tools/fuzz_gpu.py --seeds 300 --first 17600 --variant scalar_spectral
f e MIW_JOB_CHUNK_FORCE=1 MIW_JOB_CHUNK=4 MIW_BVH8=0 timeout 500 python tools/fuzz_gpu.py --seeds 300 --first 17900
