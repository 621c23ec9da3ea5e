#!/bin/bash
# One GPU session of round 6 (overwritten per session; results under gpurun_out/<tag>_*). Usage: bash tools/gpu_session.sh <tag>
# This one (r6d): the whole GPU tier on the options refactor, FETCH_SIZE calibration, the fixed phase statistics, the default bench line with parity fields
tag=${1:-r6d}; out=$(pwd)/gpurun_out; mkdir -p $out
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^$" | tail -15) > $out/${tag}_pytest_gpu.txt; tail -6 $out/${tag}_pytest_gpu.txt
timeout 600 bash tools/fetch_calib.sh > $out/${tag}_fetch_calib.txt 2>&1; cat $out/${tag}_fetch_calib.txt
MIWAVE_LIB_DIR=$(pwd)/build_exp/stats MIW_DEBUG=1 timeout 300 python tools/ab_render.py --scenes matball:64,interior:16 --set "" --reps 1 > $out/${tag}_stats.txt 2>&1; grep "phase \|Msamples" $out/${tag}_stats.txt | tail -14
timeout 900 python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6d_bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "parity", d.get("parity"))
for k, v in (d.get("extras") or {}).items():
    print(k, v if not isinstance(v, dict) else (round(v["value"], 1), v.get("ms_path_kernel"), v.get("parity")))
PY
