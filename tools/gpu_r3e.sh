#!/bin/bash
# Round 3, GPU session E: what the placed per-SIMD queues do on the 1/8 shard (debug print), and the event times of its two launches
out=gpurun_out; mkdir -p $out
export TMPDIR=/tmp
for spec in "place:" "noplace:MIW_PLACE=0"; do
  label=${spec%%:*}; envs=${spec#*:}
  env $envs MIW_DEBUG=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-live-counters --shard tiles --shard-of 8 > $out/r3e_$label.log 2> $out/r3e_$label.err
  tail -1 $out/r3e_$label.log | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$label', j['ms_per_step'], j['roofline']['kernel_ms'], j['roofline']['launches'])"
  grep "placed queues" $out/r3e_$label.err | tail -2
done
( cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$out/r3e_trace -- python $OLDPWD/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --no-live-counters --shard tiles --shard-of 8 > $OLDPWD/$out/r3e_trace.log 2>&1 )
python tools/rocprof_summary.py stats $out/r3e_trace | head -8
