#!/bin/bash
# SQ-only PMC passes of one bench configuration (the TA / TCP / TD counter groups hang rocprofv3 on this pool — do not add them).
# Usage: bash tools/pmc_sq.sh <tag> <lib dir or ""> <bench args...>
tag=$1; lib=$2; shift 2
repo=$(pwd); out=$repo/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
[ -n "$lib" ] && export MIWAVE_LIB_DIR=$repo/$lib
B="python $repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline $*"
cd /tmp
i=0
for set in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH" \
           "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/${tag}_sq$i -- $B > $out/${tag}_sq$i.log 2>&1
done
cd $repo
find $out -name "*.db" -size +20M -delete 2>/dev/null
python tools/rocprof_summary.py pmc $out/${tag}_sq1 $out/${tag}_sq2 $out/${tag}_sq3 $out/${tag}_sq4 2>/dev/null | grep -E "k_trace_stream|k_path_phased|k_shade|k_trace<|k_path_resident|^==" | cut -c1-700
