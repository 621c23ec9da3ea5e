#!/bin/bash
# r5r: instruction-cache counters of the phase machine (57.7 KB of code against a 64 KB instruction cache shared by two CUs)
tag=${1:-r5r}; out=$(pwd)/gpurun_out; mkdir -p $out
export TMPDIR=/tmp MIW_BENCH_NO_LIVE=1
repo=$(pwd)
for sc in "c3 --scene matball --spp 64" "c4 --scene interior --spp 16" "c2 --spp 128"; do
  set -- $sc; name=$1; shift
  ( cd /tmp; timeout 150 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $out/${tag}_${name}_icache -- python $repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras $* > $out/${tag}_${name}_icache.log 2>&1; echo "rc $?" )
done
for n in c3 c4 c2; do python tools/rocprof_summary.py pmc $out/${tag}_${n}_icache; done 2>&1 | grep "k_path" | cut -c1-300
