#!/bin/bash
# Texture-address / L1 (TA, TCP, TD) and SQ counters of one bench configuration: is the kernel bound by VALU issue, by
# waiting on memory, or by the gather rate of the L1 pipe? Usage: bash tools/pmc_ta.sh <tag> <lib dir or ""> <bench args...>
tag=$1; lib=$2; shift 2
repo=$(pwd); out=$repo/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
[ -n "$lib" ] && export MIWAVE_LIB_DIR=$repo/$lib
B="python $repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline $*"
cd /tmp
i=0
for set in "GRBM_GUI_ACTIVE GRBM_TA_BUSY TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_LATENCY_sum" \
           "TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_READ_sum TCP_TOTAL_ACCESSES_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/${tag}_ta$i -- $B > $out/${tag}_ta$i.log 2>&1
  tail -1 $out/${tag}_ta$i.log | cut -c1-160
done
cd $repo
find $out -name "*.db" -size +20M -delete 2>/dev/null
python tools/rocprof_summary.py pmc $out/${tag}_ta1 $out/${tag}_ta2 $out/${tag}_ta3 $out/${tag}_ta4 $out/${tag}_ta5 2>/dev/null | grep -E "k_trace_stream|k_path_phased|k_shade|k_trace<|k_path_resident|^==" | cut -c1-700
