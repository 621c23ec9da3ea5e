#!/bin/bash
# Round 3, GPU session A: the whole GPU test tier on the new film path, the default bench line, A/B lines of the film replay
# (group shape, XCD swizzle, legacy log), of the candidate-pair packet kernel and of the leaf size, the 1/8 shard, and the
# traffic counters of C2. Usage (on the GPU box, repo root): bash tools/gpu_r3a.sh
out=gpurun_out; mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=20 > $out/r3a_pytest.log 2>&1; echo "pytest rc $?" >> $out/r3a_pytest.log
tail -5 $out/r3a_pytest.log
timeout 500 python bench.py > $out/r3a_bench_c2.log 2> $out/r3a_bench_c2.err; tail -1 $out/r3a_bench_c2.log | cut -c1-1500
line() {   # line <label> <env...> -- <bench args>
  label=$1; shift; envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-live-counters "$@" > $out/r3a_$label.log 2> $out/r3a_$label.err
  python - "$out/r3a_$label.log" "$label" <<'P'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-26s %9.1f Msamples/s  step %8.2f ms  kernels %s" % (sys.argv[2], j["value"], j["ms_per_step"], j["roofline"]["kernel_ms"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
P
}
line base --
line film_g2 MIW_FILM_GROUP=2 --
line film_g4 MIW_FILM_GROUP=4 --
line film_noxcd MIW_FILM_XCD=0 --
line film_legacy MIW_FILM_LEGACY=1 --
line pair MIWAVE_LIB_DIR=$PWD/build_exp/pair --
line leaf1 MIW_MAX_LEAF=1 --
line leaf3 MIW_MAX_LEAF=3 --
line leaf4 MIW_MAX_LEAF=4 --
line shard8 -- --shard tiles --shard-of 8
line shard8_pair MIWAVE_LIB_DIR=$PWD/build_exp/pair -- --shard tiles --shard-of 8
line shard8_wg4 MIW_WG_PER_CU=4 -- --shard tiles --shard-of 8
line shard8_wg2 MIW_WG_PER_CU=2 -- --shard tiles --shard-of 8
# traffic of the C2 kernels (each counter set in its own pass, kernel trace only)
B="python $PWD/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --no-live-counters"
( cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$out/r3a_c2_trace -- $B > $OLDPWD/$out/r3a_c2_trace.log 2>&1
  timeout 300 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --kernel-trace --output-format csv -d $OLDPWD/$out/r3a_c2_pmc3 -- $B > $OLDPWD/$out/r3a_c2_pmc3.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum --kernel-trace --output-format csv -d $OLDPWD/$out/r3a_c2_pmc4 -- $B > $OLDPWD/$out/r3a_c2_pmc4.log 2>&1 )
find $out -name "*.db" -size +20M -delete 2>/dev/null
python tools/rocprof_summary.py stats $out/r3a_c2_trace 2>&1 | head -12
python tools/rocprof_summary.py pmc $out/r3a_c2_pmc3 $out/r3a_c2_pmc4 2>&1 | cut -c1-400 | head -30
