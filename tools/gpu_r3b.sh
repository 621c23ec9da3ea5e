#!/bin/bash
# Round 3, GPU session B: the GPU test tier on the register / DPP film replay, the two-level class search, the device-side 4-wide
# collapse and the per-XCD pixel queues; A/B lines of each; C3 / C4 with both tree builders; traffic counters of C2.
out=gpurun_out; mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=12 > $out/r3b_pytest.log 2>&1; echo "pytest rc $?" >> $out/r3b_pytest.log
tail -4 $out/r3b_pytest.log; grep -E "^(FAILED|ERROR)" $out/r3b_pytest.log | head -20
timeout 600 python bench.py > $out/r3b_bench_c2.log 2> $out/r3b_bench_c2.err; tail -1 $out/r3b_bench_c2.log | cut -c1-400; python - $out/r3b_bench_c2.log <<'P'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("kernels", j["roofline"]["kernel_ms"]); print("cpu", json.dumps(j["cpu_baseline"])[:600]); print("extras", json.dumps(j["extras"])[:1500])
except Exception as e:
    print("bench parse failed", e)
P
line() {   # line <label> <env...> -- <bench args>
  label=$1; shift; envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs MIW_DEBUG=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-live-counters "$@" > $out/r3b_$label.log 2> $out/r3b_$label.err
  python - "$out/r3b_$label.log" "$label" <<'P'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-22s %8.1f Msamples/s step %8.2f ms bvh %7.1f ms (%s) kernels %s" % (sys.argv[2], j["value"], j["ms_per_step"], j["config"]["bvh"]["build_ms"], j["config"]["bvh"]["builder"][:6], j["roofline"]["kernel_ms"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
P
}
line base --
line film_lds MIW_FILM_DPP=0 --
line film_g4 MIW_FILM_GROUP=4 --
line film_g2 MIW_FILM_GROUP=2 --
line nt MIWAVE_LIB_DIR=$PWD/build_exp/nt --
line xcdq MIW_XCD_QUEUES=1 --
line shard8 -- --shard tiles --shard-of 8
line c3 -- --scene matball --spp 64
line c3_q1 MIW_XCD_QUEUES=0 -- --scene matball --spp 64
line c3_lbvh -- --scene matball --spp 64 --bvh-quality 0
line c3_lbvh_host MIW_BVH4_HOST=1 -- --scene matball --spp 64 --bvh-quality 0
line c4 -- --scene interior --spp 16
line c4_q1 MIW_XCD_QUEUES=0 -- --scene interior --spp 16
line c4_lbvh -- --scene interior --spp 16 --bvh-quality 0
line c4_lbvh_host MIW_BVH4_HOST=1 -- --scene interior --spp 16 --bvh-quality 0
grep -h "bvh4" $out/r3b_c3_lbvh.err $out/r3b_c4_lbvh.err $out/r3b_c4_lbvh_host.err | head
# traffic of the C2 kernels (each counter set in its own pass, kernel trace only), default build and the streaming-store build
B="python $PWD/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --no-live-counters"
( cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$out/r3b_c2_trace -- $B > $OLDPWD/$out/r3b_c2_trace.log 2>&1
  timeout 300 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --kernel-trace --output-format csv -d $OLDPWD/$out/r3b_c2_pmc3 -- $B > $OLDPWD/$out/r3b_c2_pmc3.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum --kernel-trace --output-format csv -d $OLDPWD/$out/r3b_c2_pmc4 -- $B > $OLDPWD/$out/r3b_c2_pmc4.log 2>&1
  MIWAVE_LIB_DIR=$OLDPWD/build_exp/nt timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum --kernel-trace --output-format csv -d $OLDPWD/$out/r3b_c2nt_pmc4 -- $B > $OLDPWD/$out/r3b_c2nt_pmc4.log 2>&1 )
find $out -name "*.db" -size +20M -delete 2>/dev/null
python tools/rocprof_summary.py stats $out/r3b_c2_trace 2>&1 | head -6
python tools/rocprof_summary.py pmc $out/r3b_c2_pmc3 $out/r3b_c2_pmc4 $out/r3b_c2nt_pmc4 2>&1 | grep -E "==|k_path|k_film" | cut -c1-330
