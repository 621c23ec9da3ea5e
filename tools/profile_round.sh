#!/bin/bash
# One GPU-box session that produces everything profiles/ and DESIGN.md §5 quote:
#   rocprofv3 kernel trace + stats and PMC passes (each counter set in its own run, never combined with a trace domain
#   other than --kernel-trace; SQ / TCC / GRBM counters only — the TA / TCP / TD groups hang rocprofv3 on this pool) of
#   C2 (the default bench), C3 (material balls) and the C4-class interior; the bench lines of C2 (the default line: CPU baseline,
#   live counters, extras at the configured spp), C3, C4 (device-built SAH tree, host-built, radix tree), C5, the triangle-count
#   series, the wavefront plan and every rank's tile shard of an 8-GPU frame of C2 / C3 / C4 (tools/shard_table.py).
# Usage (from the repo root, on the GPU box):  [LEAN=1] bash tools/profile_round.sh r04   (LEAN skips the A/B lines of knobs that are not defaults)
# Outputs under gpurun_out/<tag>_*; tools/make_round_profiles.py <tag> turns them into profiles/<tag>_* and profiles/traffic.json.
tag=${1:-r04}
repo=$(pwd)
out=$repo/gpurun_out
mkdir -p $out
export TMPDIR=/tmp
export MIW_BENCH_NO_LIVE=1     # bench.py's own live PMC passes only in the default line below (these runs ARE the PMC passes)
pmc() {   # [ENV=VAL] pmc <name> <bench args...>  (the environment of the call reaches bench.py)
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
pmc c3plan1 --scene matball --spp 64 --plan 1                                # round 6: the wavefront plan (SoA queues in HBM: the architecture north_star names) — REAL queue bytes against 8 TB/s
[ -n "$LEAN" ] || MIW_BVH8=0 pmc c4bvh4 --scene interior --spp 16          # the 4-wide twin of the same build: traffic and wait share against the 8-wide walk's
cd $repo
line() {  # line <name> [ENV=VAL ...] -- <bench args...>: one bench line into $out/${tag}_<name>.log
  local name=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 400 python bench.py --no-cpu-baseline --no-extras "$@" > $out/${tag}_${name}.log 2> $out/${tag}_${name}.err
}
env -u MIW_BENCH_NO_LIVE timeout 900 python bench.py > $out/${tag}_bench_c2.log 2> $out/${tag}_bench_c2.err      # THE default line: live counters, CPU leg, extras at the configured spp
C3="--scene matball --steps 1 --warmup 1"; C4="--scene interior --steps 1 --warmup 1"
line bench_c3 -- $C3 --spp 1024                                   # configs[2] as configured
line bench_c3_bvh4 MIW_BVH8=0 -- $C3 --spp 256                    # the 4-wide walk (round 4's tree) on the same build
line bench_c3_plan1 -- $C3 --spp 256 --plan 1                     # wavefront plan (stream walk kernel)
line bench_c3_hostsah -- $C3 --spp 256 --bvh-quality 1            # the same trees built / collapsed by the host
line bench_c3_lbvh -- $C3 --spp 256 --bvh-quality 64              # MI_BVH_RADIX_TREE: the radix tree of rounds 2 - 3 (4-wide walk)
line bench_c3_pooled MIW_POOLED=1 -- $C3 --spp 256                # round 6: k_path_pooled (opt-in), 12 x 1 and 8 x 2
line bench_c3_pooled82 MIW_POOLED=1 MIW_POOL_SHAPE=8x2 -- $C3 --spp 256
line bench_c5 -- --variant scalar_spectral --scene glassblock --steps 1 --warmup 1
line bench_c4 MIW_DEBUG=1 -- $C4 --spp 32                         # configs[3] class, device-built SAH tree (the default); .err: the builder's timing
line bench_c4_bvh4 MIW_BVH8=0 -- $C4 --spp 32
line bench_c4_pooled MIW_POOLED=1 -- $C4 --spp 32
line bench_c4_hostsah -- $C4 --spp 32 --bvh-quality 1
line bench_c4_lbvh -- $C4 --spp 32 --bvh-quality 64
line bench_c4_sah_r4 MIW_SAH_HUGE=0 MIW_DEBUG=1 -- $C4 --spp 16   # the builder with one workgroup per candidate (round 4's launch shape): build ms
line bench_direct_c2 -- --integrator direct --steps 2 --warmup 1
[ -n "$LEAN" ] || { for t in 0 1 2 3 4; do line tess_$t -- --scene matball --tess $t --spp 128 --steps 1 --warmup 1; done; }
# every rank's tile shard of an 8-GPU frame, at the configured spp, + the floor (tools/shard_table.py)
timeout 900 python tools/shard_table.py --out $out/${tag}_shards.txt --json $out/${tag}_shards.json > $out/${tag}_shards.log 2>&1
# keep the merged artefacts small: traces of the PMC passes are only needed for the per-kernel durations
find $out -name "*.db" -size +20M -delete 2>/dev/null
du -sh $out | tail -1
for f in $out/${tag}_bench_*.log; do echo "== $f"; tail -1 $f | cut -c1-330; done
tail -50 $out/${tag}_shards.txt
