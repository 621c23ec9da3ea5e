#!/bin/bash
tag=r6n; out=$(pwd)/gpurun_out; mkdir -p $out
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-live-counters"
line() { name=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift; env "${envs[@]}" timeout 400 $B "$@" > $out/${tag}_${name}.log 2> $out/${tag}_${name}.err; python - $out/${tag}_${name}.log $name <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
    print("%-28s %8.1f Msamples/s  %8.2f ms/frame  parity %s kernels %s" % (sys.argv[2], d["value"], d["ms_per_step"], (d.get("parity") or {}).get("match"), {k: round(v / d["steps"], 2) for k, v in r["kernel_ms"].items()}))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for rep in 1 2; do
line c2_w4_$rep --
line c2_w5_$rep MIWAVE_LIB_DIR=$(pwd)/build_exp/waves5 MIW_WG_PER_CU=5 --
done
line c2_w5_wg4 MIWAVE_LIB_DIR=$(pwd)/build_exp/waves5 --
line c2_w5_wg6 MIWAVE_LIB_DIR=$(pwd)/build_exp/waves5 MIW_WG_PER_CU=6 --
