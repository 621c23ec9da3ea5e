set -x
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "film_replay or float32_film or ragged or cornell_diffuse or golden_films or tiny_blocks or samples_per_pass" 2>&1 | tail -5
export MIW_BENCH_NO_LIVE=1
for v in "MIW_X=1" "MIW_FILM_LANES=0" "MIW_FILM_LANES=2" "MIW_X=1" "MIW_FILM_LANES=0"; do env $v timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['value'],1), round(d['ms_per_step'],2), d['roofline']['kernel_ms'])"; done
