#!/bin/bash
# One GPU-box session that produces everything profiles/ and DESIGN.md §5 quote:
#   rocprofv3 kernel trace + stats and PMC passes (each counter set in its own run, never combined with a trace domain
#   other than --kernel-trace; SQ / TCC / GRBM counters only — the TA / TCP / TD groups hang rocprofv3 on this pool) of
#   C2 (the default bench), C3 (material balls) and the C4-class interior; the bench lines of C2 (with the CPU baseline),
#   C3, C4 (SAH and device LBVH), C5, the triangle-count series, the wavefront plan, and the tile-shard table.
# Usage (from the repo root, on the GPU box):  bash tools/profile_round.sh r03
# Outputs under gpurun_out/<tag>_*; tools/make_round_profiles.py <tag> turns them into profiles/<tag>_* and profiles/traffic.json.
tag=${1:-r03}
repo=$(pwd)
out=$repo/gpurun_out
mkdir -p $out
export TMPDIR=/tmp
export MIW_BENCH_NO_LIVE=1     # bench.py's own live PMC passes only in the default line below (these runs ARE the PMC passes)
pmc() {   # pmc <name> <bench args...>
  name=$1; shift
  B="python $repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras $*"
  ( cd /tmp
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_${name}_trace -- $B > $out/${tag}_${name}_trace.log 2>&1
    timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $out/${tag}_${name}_pmc1 -- $B > $out/${tag}_${name}_pmc1.log 2>&1
    timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $out/${tag}_${name}_pmc2 -- $B > $out/${tag}_${name}_pmc2.log 2>&1
    timeout 300 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --kernel-trace --output-format csv -d $out/${tag}_${name}_pmc3 -- $B > $out/${tag}_${name}_pmc3.log 2>&1
    timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum --kernel-trace --output-format csv -d $out/${tag}_${name}_pmc4 -- $B > $out/${tag}_${name}_pmc4.log 2>&1 )
}
pmc c2
pmc c3 --scene matball --spp 64
pmc c4 --scene interior --spp 16
cd $repo
env -u MIW_BENCH_NO_LIVE timeout 900 python bench.py > $out/${tag}_bench_c2.log 2>&1
timeout 300 python bench.py --scene matball --spp 1024 --steps 1 --warmup 1 --no-cpu-baseline > $out/${tag}_bench_c3.log 2>&1
timeout 300 python bench.py --scene matball --spp 256 --steps 1 --warmup 1 --no-cpu-baseline --plan 1 > $out/${tag}_bench_c3_plan1.log 2>&1
MIW_PHASED=0 timeout 300 python bench.py --scene matball --spp 256 --steps 1 --warmup 1 --no-cpu-baseline > $out/${tag}_bench_c3_lockstep.log 2>&1
timeout 300 python bench.py --variant scalar_spectral --scene glassblock --steps 1 --warmup 1 --no-cpu-baseline > $out/${tag}_bench_c5.log 2>&1
timeout 400 python bench.py --scene interior --spp 32 --steps 1 --warmup 1 --no-cpu-baseline > $out/${tag}_bench_c4.log 2>&1
timeout 400 python bench.py --scene interior --spp 32 --steps 1 --warmup 1 --no-cpu-baseline --bvh-quality 0 > $out/${tag}_bench_c4_lbvh.log 2>&1
# the triangle-count series between the packet kernels (<= 64 triangles) and config 3 (icosphere levels 0..4 of the two balls)
for t in 0 1 2 3 4; do timeout 200 python bench.py --scene matball --tess $t --spp 128 --steps 1 --warmup 1 --no-cpu-baseline > $out/${tag}_tess_$t.log 2>&1; done
for so in 1 2 4 8; do timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --shard tiles --shard-of $so > $out/${tag}_shard_$so.log 2>&1; done
# the 1/8 shard without the per-SIMD placement, without the priorities as well, and with a shorter measuring launch
MIW_PLACE=0 timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --shard tiles --shard-of 8 > $out/${tag}_shard_8_noplace.log 2>&1
MIW_PLACE=0 MIW_TAIL_PRIO=0 timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --shard tiles --shard-of 8 > $out/${tag}_shard_8_plain.log 2>&1
MIW_PLACE_MEASURE=16 timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --shard tiles --shard-of 8 > $out/${tag}_shard_8_measure16.log 2>&1
MIW_TAIL_PRIO=0 timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $out/${tag}_bench_c2_noprio.log 2>&1
MIW_FILM_LEGACY=1 timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $out/${tag}_bench_c2_legacy_log.log 2>&1
for so in 2 8; do timeout 300 python bench.py --scene matball --spp 256 --steps 1 --warmup 1 --no-cpu-baseline --shard tiles --shard-of $so > $out/${tag}_c3_shard_$so.log 2>&1; done
MIW_TAIL_PRIO=0 timeout 300 python bench.py --scene matball --spp 256 --steps 1 --warmup 1 --no-cpu-baseline --shard tiles --shard-of 8 > $out/${tag}_c3_shard_8_plain.log 2>&1
timeout 300 python bench.py --scene interior --spp 64 --steps 1 --warmup 1 --no-cpu-baseline --shard tiles --shard-of 8 > $out/${tag}_c4_shard_8.log 2>&1
MIW_PLACE=0 MIW_TAIL_PRIO=0 timeout 300 python bench.py --scene interior --spp 64 --steps 1 --warmup 1 --no-cpu-baseline --shard tiles --shard-of 8 > $out/${tag}_c4_shard_8_plain.log 2>&1
MIW_BVH4_HOST=1 timeout 400 python bench.py --scene interior --spp 32 --steps 1 --warmup 1 --no-cpu-baseline --bvh-quality 0 > $out/${tag}_bench_c4_lbvh_hostcollapse.log 2>&1
MIW_LBVH_LEAF=1 timeout 400 python bench.py --scene interior --spp 32 --steps 1 --warmup 1 --no-cpu-baseline --bvh-quality 0 > $out/${tag}_bench_c4_lbvh_leaf1.log 2>&1
timeout 300 python bench.py --scene matball --spp 256 --steps 1 --warmup 1 --no-cpu-baseline --bvh-quality 0 > $out/${tag}_bench_c3_lbvh.log 2>&1
# three wavefronts per SIMD instead of four (the default since the register diet)
MIW_PHASED_WAVES=3 timeout 300 python bench.py --scene matball --spp 256 --steps 1 --warmup 1 --no-cpu-baseline > $out/${tag}_bench_c3_w3.log 2>&1
MIW_PHASED_WAVES=3 timeout 400 python bench.py --scene interior --spp 32 --steps 1 --warmup 1 --no-cpu-baseline > $out/${tag}_bench_c4_w3.log 2>&1
timeout 300 python bench.py --integrator direct --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $out/${tag}_bench_direct_c2.log 2>&1
# keep the merged artefacts small: traces of the PMC passes are only needed for the per-kernel durations
find $out -name "*.db" -size +20M -delete 2>/dev/null
du -sh $out | tail -1
for f in $out/${tag}_bench_*.log $out/${tag}_tess_*.log $out/${tag}_shard_*.log $out/${tag}_c3_shard_*.log $out/${tag}_c4_shard_*.log; do echo "== $f"; tail -1 $f | cut -c1-330; done
