#!/bin/bash
# One GPU session of round 6 (overwritten per session; results under gpurun_out/<tag>_*). Usage: bash tools/gpu_session.sh <tag>
# This one: the round's profile session on the final kernels (chunk jobs in every full-frame path kernel): the smoke entry, the whole GPU tier, tools/profile_round.sh
# (kernel stats + PMC passes + bench lines + shards), the device fuzz by hand (defaults; 2-sample chunks forced). Every step under its own timeout.
tag=${1:-r06}; out=$(pwd)/gpurun_out; mkdir -p $out
(timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3) > $out/${tag}_smoke.txt; tail -1 $out/${tag}_smoke.txt
(timeout 420 python -m pytest tests/test_job_chunks.py -m gpu -x -q 2>&1 | tail -8) > $out/${tag}_pytest_jobs.txt; tail -2 $out/${tag}_pytest_jobs.txt
(timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^$" | tail -12) > $out/${tag}_pytest_gpu.txt; grep -h "passed\|failed" $out/${tag}_pytest_gpu.txt
LEAN=1 timeout 1500 bash tools/profile_round.sh $tag > $out/${tag}_profile_round.log 2>&1; tail -12 $out/${tag}_profile_round.log
(timeout 400 python tools/fuzz_gpu.py --seeds 400 --first 15000 2>&1 | tail -2) > $out/${tag}_fuzz_a.txt; tail -1 $out/${tag}_fuzz_a.txt
(MIW_JOB_CHUNK_FORCE=1 MIW_JOB_CHUNK=2 timeout 400 python tools/fuzz_gpu.py --seeds 300 --first 15400 2>&1 | tail -2) > $out/${tag}_fuzz_b.txt; tail -1 $out/${tag}_fuzz_b.txt
(MIW_FILM_LANES=1 timeout 400 python tools/fuzz_gpu.py --seeds 200 --first 15700 2>&1 | tail -2) > $out/${tag}_fuzz_c.txt; tail -1 $out/${tag}_fuzz_c.txt
(timeout 400 python tools/fuzz_gpu.py --seeds 150 --first 15900 --variant scalar_spectral 2>&1 | tail -2) > $out/${tag}_fuzz_e.txt; tail -1 $out/${tag}_fuzz_e.txt
# the threshold of the chunk jobs at 1.2 pixels per resident lane: a rank's shard of a 4-GPU frame (1.6 pixels per lane) now takes them by default
(timeout 300 python tools/shard_table.py --configs c2 --ranks 4 2>&1 | grep "^## \|max-rank" | cut -c1-200) > $out/${tag}_shards_4ranks.txt; cat $out/${tag}_shards_4ranks.txt
(timeout 400 python tools/shard_table.py --configs c3 --ranks 4 2>&1 | grep "^## \|max-rank" | cut -c1-200) >> $out/${tag}_shards_4ranks.txt; tail -2 $out/${tag}_shards_4ranks.txt
# two ranks SHARING the GPU (testing mode): is the run-to-run pathology the chunk jobs'? with the switch off, twice each
A="--gpus 2 --share-gpu --backend gloo --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-live-counters"
for rep in 1 2; do
  MIW_JOB_CHUNK=0 timeout 300 python bench.py $A > $out/${tag}_shared2_off_$rep.log 2> $out/${tag}_shared2_off_$rep.err; tail -1 $out/${tag}_shared2_off_$rep.log | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/shared2 off $rep /"
  timeout 300 python bench.py $A > $out/${tag}_shared2_on_$rep.log 2> $out/${tag}_shared2_on_$rep.err; tail -1 $out/${tag}_shared2_on_$rep.log | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/shared2 on $rep /"
done
