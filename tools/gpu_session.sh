#!/bin/bash
# One GPU session of round 6 (overwritten per session; results under gpurun_out/<tag>_*). Usage: bash tools/gpu_session.sh <tag>
# This one (r6q): chunk jobs (resident_kernel.h: QueueWork::fetch_job) with the parked job in LDS, in the packet kernels AND the phase machine: the hand-over tests, then A/B lines —
# C2 (switch off / on), C3 at 1 024 spp and C4 at 512 spp through three settings: the build without chunk jobs in the phase machine (build_exp/pj0: -DMIW_PHASED_JOBS=0, the kernels
# of the last profile session), this build with the switch off (what the compiled-in code costs), this build as it is; C5. Every step under its own timeout.
tag=${1:-r6q}; out=$(pwd)/gpurun_out; mkdir -p $out
(timeout 420 python -m pytest tests/test_job_chunks.py -m gpu -x -q 2>&1 | tail -15) > $out/${tag}_pytest_jobs.txt; tail -3 $out/${tag}_pytest_jobs.txt
B="python bench.py --no-cpu-baseline --no-extras --no-live-counters"
line() { name=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift; env "${envs[@]}" timeout 200 $B "$@" > $out/${tag}_${name}.log 2> $out/${tag}_${name}.err; python - $out/${tag}_${name}.log $name <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
    print("%-28s %8.1f Msamples/s  %8.2f ms/frame  parity %s kernels %s" % (sys.argv[2], d["value"], d["ms_per_step"], (d.get("parity") or {}).get("match"), {k: round(v / d["steps"], 2) for k, v in r["kernel_ms"].items()}))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
PJ0=MIWAVE_LIB_DIR=$(pwd)/build_exp/pj0
line c2_off MIW_JOB_CHUNK=0 -- --steps 3 --warmup 1
line c2_jobs -- --steps 3 --warmup 1
C3="--scene matball --steps 1 --warmup 1 --spp 1024"; C4="--scene interior --steps 1 --warmup 1 --spp 512"
for rep in 1 2; do
line c3_pj0_$rep $PJ0 -- $C3
line c3_off_$rep MIW_JOB_CHUNK=0 -- $C3
line c3_jobs_$rep -- $C3
done
for rep in 1 2; do
line c4_pj0_$rep $PJ0 -- $C4
line c4_off_$rep MIW_JOB_CHUNK=0 -- $C4
line c4_jobs_$rep -- $C4
done
line c3_jobs128 MIW_JOB_CHUNK=128 -- $C3
line c3_jobs256 MIW_JOB_CHUNK=256 -- $C3
line c5_off MIW_JOB_CHUNK=0 -- --variant scalar_spectral --scene glassblock --steps 1 --warmup 1
line c5_jobs -- --variant scalar_spectral --scene glassblock --steps 1 --warmup 1
(MIW_JOB_CHUNK_FORCE=1 MIW_JOB_CHUNK=4 timeout 300 python tools/fuzz_gpu.py --seeds 150 --first 14300 2>&1 | tail -2) > $out/${tag}_fuzz_b.txt; tail -1 $out/${tag}_fuzz_b.txt
(timeout 300 python tools/fuzz_gpu.py --seeds 150 --first 14000 2>&1 | tail -2) > $out/${tag}_fuzz_a.txt; tail -1 $out/${tag}_fuzz_a.txt
