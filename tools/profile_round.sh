#!/bin/bash
# One GPU-box session that produces everything profiles/ and DESIGN.md §5 quote:
#   rocprofv3 kernel trace + stats of the default bench (C2), four PMC passes (each counter set in its own run,
#   never combined with a trace), the bench lines of C2 (with the CPU baseline), C3, C5, C4-class and the
#   shard-size tables (tile shards and pass shards). Usage (from the repo root, on the GPU box):  bash tools/profile_round.sh r01
# Outputs under gpurun_out/<tag>_*; summarise with tools/rocprof_summary.py and copy into profiles/.
tag=${1:-r01}
repo=$(pwd)
out=$repo/gpurun_out
mkdir -p $out
export TMPDIR=/tmp
B="python $repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_trace -- $B > $out/${tag}_trace.log 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $out/${tag}_pmc1 -- $B > $out/${tag}_pmc1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $out/${tag}_pmc2 -- $B > $out/${tag}_pmc2.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/${tag}_pmc3 -- $B > $out/${tag}_pmc3.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/${tag}_pmc4 -- $B > $out/${tag}_pmc4.log 2>&1
cd $repo
timeout 400 python bench.py > $out/${tag}_bench_c2.log 2>&1
timeout 300 python bench.py --scene matball --spp 1024 --steps 1 --warmup 1 --no-cpu-baseline > $out/${tag}_bench_c3.log 2>&1
timeout 300 python bench.py --variant scalar_spectral --steps 1 --warmup 1 --no-cpu-baseline > $out/${tag}_bench_c5.log 2>&1
timeout 400 python bench.py --scene interior --spp 32 --steps 1 --warmup 1 --no-cpu-baseline > $out/${tag}_bench_c4.log 2>&1
timeout 400 python bench.py --scene interior --spp 32 --steps 1 --warmup 1 --no-cpu-baseline --bvh-quality 0 > $out/${tag}_bench_c4_lbvh.log 2>&1
for so in 1 2 4 8; do timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --shard tiles --shard-of $so > $out/${tag}_shard_$so.log 2>&1; done
for so in 2 4 8; do timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --shard passes --shard-of $so > $out/${tag}_pass_shard_$so.log 2>&1; done
# keep the merged artefacts small: traces of the PMC passes are only needed for the per-kernel durations
find $out -name "*.db" -size +20M -delete 2>/dev/null
du -sh $out | tail -1
for f in $out/${tag}_bench_*.log $out/${tag}_shard_*.log; do echo "== $f"; tail -1 $f | cut -c1-400; done
