cd $GRAFT_REPO_ROOT
export MIW_BENCH_NO_LIVE=1
for v in "MIW_FL_DBG=0" "MIW_FL_DBG=4" "MIW_FL_DBG=8" "MIW_FL_DBG=16" "MIW_FL_DBG=4" "MIW_FL_DBG=8" "MIW_FL_DBG=16"; do env $v timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['value'],1), round(d['ms_per_step'],2), d['roofline']['kernel_ms'])"; done
