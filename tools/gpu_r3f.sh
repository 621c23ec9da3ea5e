#!/bin/bash
# Round 3, GPU session F: the GPU test tier and __graft_entry__.smoke() on the final tree, then the round's profile session
out=gpurun_out; mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=10 > $out/r03_pytest_gpu.log 2>&1; echo "pytest rc $?" >> $out/r03_pytest_gpu.log
tail -14 $out/r03_pytest_gpu.log; grep -E "^(FAILED|ERROR)" $out/r03_pytest_gpu.log | head -20
timeout 300 python __graft_entry__.py --smoke > $out/r03_smoke.log 2>&1; tail -2 $out/r03_smoke.log
bash tools/profile_round.sh r03 2>&1 | tail -120
