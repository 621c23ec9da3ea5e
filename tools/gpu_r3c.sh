#!/bin/bash
# Round 3, GPU session C: GPU test tier on the fat-leaf LBVH, the unconditional-load film replay, packed-fma box tests, streaming
# log stores and least-progress-first wave priorities; A/B lines; SQ counters of the C2 kernels.
out=gpurun_out; mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 > $out/r3c_pytest.log 2>&1; echo "pytest rc $?" >> $out/r3c_pytest.log
tail -4 $out/r3c_pytest.log; grep -E "^(FAILED|ERROR)" $out/r3c_pytest.log | head -20
line() {   # line <label> <env...> -- <bench args>
  label=$1; shift; envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs MIW_DEBUG=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-live-counters "$@" > $out/r3c_$label.log 2> $out/r3c_$label.err
  python - "$out/r3c_$label.log" "$label" <<'P'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-22s %8.1f Msamples/s step %8.2f ms bvh %7.1f ms (%s) kernels %s" % (sys.argv[2], j["value"], j["ms_per_step"], j["config"]["bvh"]["build_ms"], j["config"]["bvh"]["builder"][:6], j["roofline"]["kernel_ms"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
P
}
line base --
line chunk32 MIWAVE_LIB_DIR=$PWD/build_exp/chunk32 --
line film_noxcd MIW_FILM_XCD=0 --
line wg3 MIW_WG_PER_CU=3 --
line shard8 -- --shard tiles --shard-of 8
line shard8_noprio MIW_TAIL_PRIO=0 -- --shard tiles --shard-of 8
line shard8_wg4 MIW_WG_PER_CU=4 -- --shard tiles --shard-of 8
line shard4 -- --shard tiles --shard-of 4
line shard4_prio MIW_TAIL_PRIO=1 -- --shard tiles --shard-of 4
line shard2_prio MIW_TAIL_PRIO=1 -- --shard tiles --shard-of 2
line full_prio MIW_TAIL_PRIO=1 --
line c3 -- --scene matball --spp 64
line c3_lbvh -- --scene matball --spp 64 --bvh-quality 0
line c3_lbvh1 MIW_LBVH_LEAF=1 -- --scene matball --spp 64 --bvh-quality 0
line c3_lbvh8 MIW_LBVH_LEAF=8 -- --scene matball --spp 64 --bvh-quality 0
line c3_shard8 -- --scene matball --spp 256 --shard tiles --shard-of 8
line c3_shard8_noprio MIW_TAIL_PRIO=0 -- --scene matball --spp 256 --shard tiles --shard-of 8
line c4 -- --scene interior --spp 16
line c4_lbvh -- --scene interior --spp 16 --bvh-quality 0
line c4_lbvh2 MIW_LBVH_LEAF=2 -- --scene interior --spp 16 --bvh-quality 0
line c4_lbvh8 MIW_LBVH_LEAF=8 -- --scene interior --spp 16 --bvh-quality 0
grep -h "bvh4" $out/r3c_c3_lbvh.err $out/r3c_c4_lbvh.err | head
# SQ counters of the C2 kernels (each counter set in its own pass, kernel trace only)
B="python $PWD/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --no-live-counters"
( cd /tmp
  timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OLDPWD/$out/r3c_c2_pmc1 -- $B > $OLDPWD/$out/r3c_c2_pmc1.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OLDPWD/$out/r3c_c2_pmc2 -- $B > $OLDPWD/$out/r3c_c2_pmc2.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OLDPWD/$out/r3c_c2_pmc5 -- $B > $OLDPWD/$out/r3c_c2_pmc5.log 2>&1 )
find $out -name "*.db" -size +20M -delete 2>/dev/null
python tools/rocprof_summary.py pmc $out/r3c_c2_pmc1 $out/r3c_c2_pmc2 $out/r3c_c2_pmc5 2>&1 | grep -E "==|k_path|k_film_g" | cut -c1-420
