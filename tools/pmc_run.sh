#!/bin/bash
# rocprofv3 kernel trace + PMC passes of one bench configuration (each counter set in its own run, never combined with a
# trace domain other than --kernel-trace). Usage: bash tools/pmc_run.sh <tag> <lib dir or ""> <bench args...>
# Outputs gpurun_out/<tag>_{trace,pmc1..4}/ ; summarise with tools/rocprof_summary.py.
tag=$1; lib=$2; shift 2
repo=$(pwd); out=$repo/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
[ -n "$lib" ] && export MIWAVE_LIB_DIR=$repo/$lib
B="python $repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline $*"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_trace -- $B > $out/${tag}_trace.log 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $out/${tag}_pmc1 -- $B > $out/${tag}_pmc1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $out/${tag}_pmc2 -- $B > $out/${tag}_pmc2.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --kernel-trace --output-format csv -d $out/${tag}_pmc3 -- $B > $out/${tag}_pmc3.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum TCP_TCC_READ_REQ_sum --kernel-trace --output-format csv -d $out/${tag}_pmc4 -- $B > $out/${tag}_pmc4.log 2>&1
timeout 300 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM --kernel-trace --output-format csv -d $out/${tag}_pmc5 -- $B > $out/${tag}_pmc5.log 2>&1
cd $repo
find $out -name "*.db" -size +20M -delete 2>/dev/null
for p in trace pmc1 pmc2 pmc3 pmc4 pmc5; do tail -2 $out/${tag}_$p.log | cut -c1-300; done
